"""
Row-slab sharding of one match over the GPUs of a node: one process per GPU under
``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

Every output pixel is independent (reference loop ``model.pyx:476-492``),
so the path shards by contiguous output rows with no collective in the data path:

  rank g owns output rows [r0, r1) and needs input rows [r0, r1 + 2*padding) of every frame
  (its slab plus ``padding = Nw + max_shift`` halo rows on either side, ``model.pyx:286``).

Two input situations are covered:

  * the host owns the frames (the reference's API): ``input_rows`` tells each rank which rows
    to upload; nothing is exchanged.
  * the frames are already row-sharded on the devices (an upstream GPU producer):
    ``exchange_halo`` tops every rank's rows up with the neighbours' boundary rows by
    point-to-point send/recv (one xGMI link each way; 9 rows x 8192 x 10 frames x 8 B = 5.9 MB
    for BASELINE config C4).

``gather_rows`` collects the output slabs on one rank (or everywhere).
"""
import numpy as np

__all__ = ["slab_bounds", "input_rows", "match_rows", "exchange_halo", "gather_rows"]


def slab_bounds(n_rows, world, rank):
    """Output rows [r0, r1) of ``rank``: contiguous, sizes differ by at most one (larger slabs first)."""
    base, extra = divmod(int(n_rows), int(world))
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def input_rows(n_out_rows, padding, world, rank):
    """Frame rows [a, b) rank needs for its output slab (slab + halo)."""
    r0, r1 = slab_bounds(n_out_rows, world, rank)
    return r0, r1 + 2 * padding


def match_rows(model_cls, sam, ref, window_size, max_shift, world, rank, **match_kw):
    """Match this rank's output-row slab of the stacks ``sam``/``ref`` ([K, H, W] host arrays).

    Returns ``(result, (r0, r1))`` where ``result`` is the usual dictionary for output rows
    [r0, r1) of the whole image.  The slab is cut on the host, so only slab + halo rows travel
    to this rank's GPU.
    """
    sam = np.asarray(sam)
    ref = np.asarray(ref)
    padding = window_size + max_shift + getattr(model_cls, "safe_crop", 0)
    n_out = sam.shape[1] - 2 * padding
    r0, r1 = slab_bounds(n_out, world, rank)
    a, b = r0, r1 + 2 * padding
    m = model_cls(np.ascontiguousarray(sam[:, a:b]), np.ascontiguousarray(ref[:, a:b]),
                  window_size=window_size, max_shift=max_shift)
    debug = match_kw.pop("debug", None)
    if debug is not None:
        m.debug = debug
    match_kw.setdefault("quiet", True)
    return m.match(**match_kw), (r0, r1)


def exchange_halo(rows, halo, group=None):
    """Frames already sharded by rows: ``rows`` is this rank's ``[K, n_local, W]`` torch tensor of
    *owned* input rows.  Returns ``[K, halo_top + n_local + halo_bottom, W]`` with up to ``halo`` rows
    from the previous / next rank attached (edge ranks get no rows on their outer side).

    Neighbour send/recv pairs batched into one group call (``ncclSend``/``ncclRecv`` under RCCL).
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    K, n, W = rows.shape
    assert n >= halo, "a rank must own at least `halo` rows"
    ops, top, bot = [], None, None
    if rank > 0:
        top = torch.empty((K, halo, W), dtype=rows.dtype, device=rows.device)
        ops.append(dist.P2POp(dist.isend, rows[:, :halo].contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, top, rank - 1, group))
    if rank < world - 1:
        bot = torch.empty((K, halo, W), dtype=rows.dtype, device=rows.device)
        ops.append(dist.P2POp(dist.isend, rows[:, n - halo:].contiguous(), rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, bot, rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    parts = [p for p in (top, rows, bot) if p is not None]
    return torch.cat(parts, dim=1)


def gather_rows(local, n_rows, dst=0, group=None, device=None):
    """Gather per-rank result slabs (dict of ``[rows_g, N1, ...]`` arrays) into whole maps.

    ``dst=None`` gathers on every rank (all_gather), otherwise on rank ``dst`` only (others get None).
    Slabs may differ by one row, so they are padded to the largest slab for the collective.
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [slab_bounds(n_rows, world, g) for g in range(world)]
    biggest = max(b - a for a, b in sizes)
    out = {}
    for key in sorted(local):
        arr = np.ascontiguousarray(local[key])
        t = torch.from_numpy(arr)
        if device is not None:
            t = t.to(device)
        pad = torch.zeros((biggest,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        if dst is None:
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad, group=group)
        else:
            bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
            dist.gather(pad, bufs, dst=dst, group=group)
        if bufs is not None:
            out[key] = np.concatenate([bufs[g][: sizes[g][1] - sizes[g][0]].cpu().numpy() for g in range(world)], axis=0)
    return out if (dst is None or rank == dst) else None
