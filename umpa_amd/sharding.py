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

__all__ = ["slab_bounds", "input_rows", "match_rows", "exchange_halo", "gather_rows",
           "RowShardedStack", "exchange_halos", "gather_slabs", "send_rows_to"]


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


# ------------------------------------------------------------------------------------------------
# Frames that live row-sharded on the devices (BASELINE config C4; bench.py --gpus N)
# ------------------------------------------------------------------------------------------------

class RowShardedStack:
    """This rank's part of a ``[K, H, W]`` stack whose INPUT rows are split over the ranks in contiguous,
    nearly equal blocks (``slab_bounds(H, world, rank)``: the rows an upstream producer on this GPU writes).

    The rank matches the OUTPUT rows ``slab_bounds(H - 2*padding, world, rank)`` and needs the input rows
    ``input_rows(...)`` for them: its own block plus a few boundary rows of the previous / next rank.  One
    persistent buffer holds exactly the rows [lo, hi) = own block + halo; ``own_rows()`` is where the producer
    writes, ``exchange_halos`` fills the rest from the neighbours, ``frames()`` are the K contiguous 2-D views
    a model borrows (``UMPA_HIP_F_DEVICE_FRAMES``).
    """

    def __init__(self, K, H, W, padding, world, rank, device=None, dtype=None):
        import torch
        self.K, self.H, self.W, self.padding, self.world, self.rank = int(K), int(H), int(W), int(padding), int(world), int(rank)
        n_out = self.H - 2 * self.padding
        if n_out < world:
            raise ValueError("fewer output rows than ranks")
        self.own = [slab_bounds(self.H, world, g) for g in range(world)]
        self.need = [input_rows(n_out, self.padding, world, g) for g in range(world)]
        self.out = [slab_bounds(n_out, world, g) for g in range(world)]
        for g in range(world):
            (o0, o1), (a, b) = self.own[g], self.need[g]
            # the halo of a rank comes from its direct neighbours only
            if a < o0 and (g == 0 or a < self.own[g - 1][0]):
                raise ValueError("rank %d needs rows above its upper neighbour's block" % g)
            if b > o1 and (g == world - 1 or b > self.own[g + 1][1]):
                raise ValueError("rank %d needs rows below its lower neighbour's block" % g)
        (o0, o1), (a, b) = self.own[rank], self.need[rank]
        self.lo, self.hi = min(o0, a), max(o1, b)
        self.buf = torch.empty((self.K, self.hi - self.lo, self.W), dtype=dtype or torch.float64, device=device)

    # rows [r0, r1) of the whole image as a view of the buffer
    def rows(self, r0, r1):
        assert self.lo <= r0 <= r1 <= self.hi
        return self.buf[:, r0 - self.lo: r1 - self.lo]

    def own_rows(self):
        return self.rows(*self.own[self.rank])

    def frames(self):
        a, b = self.need[self.rank]
        return [self.buf[k, a - self.lo: b - self.lo] for k in range(self.K)]

    def halo_plan(self):
        """[(peer, 'send'|'recv', r0, r1)]: rows this rank sends to / receives from its neighbours."""
        g, plan = self.rank, []
        (o0, o1), (a, b) = self.own[g], self.need[g]
        if a < o0:
            plan.append((g - 1, "recv", a, o0))
        if b > o1:
            plan.append((g + 1, "recv", o1, b))
        if g > 0 and self.need[g - 1][1] > self.own[g - 1][1]:          # the upper neighbour's bottom deficit
            plan.append((g - 1, "send", self.own[g - 1][1], self.need[g - 1][1]))
        if g < self.world - 1 and self.need[g + 1][0] < self.own[g + 1][0]:
            plan.append((g + 1, "send", self.need[g + 1][0], self.own[g + 1][0]))
        return plan

    def halo_bytes(self):
        return sum((r1 - r0) * self.W * self.K * self.buf.element_size() for _, kind, r0, r1 in self.halo_plan() if kind == "recv")


def exchange_halos(stacks, group=None):
    """Fill the halo rows of every ``RowShardedStack`` in ``stacks`` from the neighbour ranks: all sends and
    receives of all stacks go into ONE ``batch_isend_irecv`` group (``ncclSend``/``ncclRecv`` under RCCL, one
    xGMI link per neighbour and direction).  Under RCCL every transfer is one frame's rows -- a contiguous view of the
    persistent buffer on both sides: sent from where the producer wrote them, received straight into the rows the model
    borrows, no staging tensor and no copy.  With the gloo backend the rows are staged through the host."""
    import torch
    import torch.distributed as dist
    via_host = dist.get_backend(group) != "nccl"
    ops, landing = [], []
    for st in stacks:
        for peer, kind, r0, r1 in st.halo_plan():
            view = st.rows(r0, r1)                                  # [K, rows, W]: contiguous per frame
            if not via_host:
                for k in range(st.K):
                    ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, view[k], peer, group))
            elif kind == "send":
                ops.append(dist.P2POp(dist.isend, view.contiguous().cpu(), peer, group))
            else:
                t = torch.empty(view.shape, dtype=view.dtype, device="cpu")
                ops.append(dist.P2POp(dist.irecv, t, peer, group))
                landing.append((view, t))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for view, t in landing:
        view.copy_(t)


class Watchdog:
    """A rank that hangs in a collective (a dead peer, a wedged link) must end the job, not hold it: ``with
    wd.phase("halo", 30):`` arms a deadline for the block; if it passes, a daemon thread reports which rank sat in which
    phase for how long and ends THIS process with a non-zero exit code (``os._exit``: a fresh exit, no interpreter
    shutdown that could block on the GPU runtime, and never an exec of a process that has touched the GPU) -- the
    launcher (torch.distributed.run) then tears the other ranks down.  Process-group construction has its own
    ``timeout=`` (bench.py passes it to ``init_process_group``)."""

    def __init__(self, rank=0, exit_code=3, stream=None):
        import threading
        self.rank, self.exit_code = int(rank), int(exit_code)
        self._stream = stream
        self._lock = threading.Lock()
        self._armed = None                                          # (name, deadline, seconds)
        self._stop = False
        self.fired = None
        self._thread = threading.Thread(target=self._run, name="umpa-watchdog", daemon=True)
        self._thread.start()

    def _run(self):
        import sys
        import time
        while not self._stop:
            time.sleep(0.2)
            with self._lock:
                armed = self._armed
            if armed and time.monotonic() > armed[1]:
                msg = "[umpa watchdog] rank %d: phase %r still running after %.0f s -- ending this process (exit code %d)\n" % (
                    self.rank, armed[0], armed[2], self.exit_code)
                (self._stream or sys.stderr).write(msg)
                (self._stream or sys.stderr).flush()
                self.fired = armed[0]
                self._exit(self.exit_code)
                return

    def _exit(self, code):                                          # (a seam for the tests)
        import os
        os._exit(code)

    def phase(self, name, seconds):
        import contextlib
        import time

        @contextlib.contextmanager
        def cm():
            with self._lock:
                self._armed = (name, time.monotonic() + float(seconds), float(seconds))
            try:
                yield
            finally:
                with self._lock:
                    self._armed = None
        return cm()

    def close(self):
        self._stop = True


def describe_device(index):
    """'<name> uuid=<...> pci=<bus id>' of HIP device ``index`` (printed per rank by bench.py: which GPUs a run used)."""
    import torch
    p = torch.cuda.get_device_properties(index)
    parts = [p.name]
    for key in ("uuid", "pci_bus_id", "pci_device_id", "gcnArchName"):
        v = getattr(p, key, None)
        if v is not None:
            parts.append("%s=%s" % (key, v))
    return " ".join(str(x) for x in parts)


def gather_slabs(local, n_rows, dst=0, group=None, out=None):
    """Gather one padded result tensor per rank on ``dst``: ``local`` is ``[biggest, ...]`` (this rank's slab in its
    first rows, ``biggest`` = the largest slab) on the device the backend moves (HIP for nccl).  Returns the
    whole ``[n_rows, ...]`` tensor on ``dst`` (written into ``out`` if given), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [slab_bounds(n_rows, world, g) for g in range(world)]
    biggest = max(b - a for a, b in sizes)
    assert local.shape[0] == biggest, "pad the local slab to the largest slab (%d rows)" % biggest
    via_host = dist.get_backend(group) != "nccl" and local.is_cuda
    send = local.cpu() if via_host else local
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if out is None:
        out = torch.empty((n_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for g, (a, b) in enumerate(sizes):
        out[a:b].copy_(bufs[g][: b - a])
    return out


def send_rows_to(tensors, wholes, n_rows, lo, hi, dst=0, group=None):
    """Rows [lo, hi) of every rank's slab travel to ``dst`` NOW (asynchronously): ``tensors`` are this rank's slab
    tensors (``[slab rows (or more), ...]``), ``wholes`` the matching ``[n_rows, ...]`` tensors on ``dst`` (ignored
    elsewhere).  One send/recv group per call with exact sizes (slabs may differ by a row); ``dst`` places its own rows
    with a device copy.  Returns the work handles to wait for.  Called once per finished row piece
    (``umpa_hip_set_rows_callback``), the transfers overlap the matching of the next piece: every rank must call it with
    the same sequence of (lo, hi)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [slab_bounds(n_rows, world, g) for g in range(world)]
    via_host = dist.get_backend(group) != "nccl"
    ops, landing = [], []
    if rank == dst:
        for g, (a, b) in enumerate(sizes):
            l, h = min(lo, b - a), min(hi, b - a)
            if h <= l:
                continue
            for t, whole in zip(tensors, wholes):
                if g == rank:
                    whole[a + l: a + h].copy_(t[l:h], non_blocking=True)
                elif via_host:
                    import torch
                    buf = torch.empty(whole[a + l: a + h].shape, dtype=whole.dtype, device="cpu")
                    ops.append(dist.P2POp(dist.irecv, buf, g, group))
                    landing.append((whole[a + l: a + h], buf))
                else:
                    ops.append(dist.P2POp(dist.irecv, whole[a + l: a + h], g, group))
    else:
        a, b = sizes[rank]
        l, h = min(lo, b - a), min(hi, b - a)
        if h > l:
            for t in tensors:
                ops.append(dist.P2POp(dist.isend, t[l:h].cpu() if via_host else t[l:h], dst, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if via_host:
        for w in works:
            w.wait()
        for view, buf in landing:
            view.copy_(buf)
        return []
    return works
