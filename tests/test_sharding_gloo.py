"""
Multi-process tests of the row-slab sharding (umpa_amd/sharding.py) on CPU: torch.distributed with the
gloo backend, world_size 2 and 3.  The pixel work itself is done by the CPU oracle here (no GPU in
this leg); what is tested is the partitioning, the halo arithmetic, the neighbour halo exchange and
the gather -- the sharded result must equal the unsharded one bit for bit.
"""
import functools
import os
import socket
import sys

import time

import numpy as np
import pytest

import farm_cpu_worker

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle import cpu_model
    from umpa_amd import sharding
    from umpa_amd.synth import make_stack

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nw, ms = 2, 3
        sam, ref, _ = make_stack(61, 70, 3, ms, df=True, seed=5, amplitude=1.5)
        P = Nw + ms
        n_out = sam.shape[1] - 2 * P

        # (1) host-owned frames: every rank matches its slab, rank 0 gathers
        res, (r0, r1) = sharding.match_rows(cpu_model.port.UMPAModelDF, sam, ref, Nw, ms, world, rank, num_threads=1)
        assert res["f"].shape[0] == r1 - r0
        whole = sharding.gather_rows({k: res[k] for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls")}, n_out, dst=0)

        # (2) device-sharded frames: each rank owns a disjoint block of INPUT rows and fetches the halo
        a, b = sharding.input_rows(n_out, P, world, rank)
        own0 = a if rank == 0 else a + P               # interior ranks own [a+P, b-P); edges own their outer halo too
        own1 = b if rank == world - 1 else b - P
        halo = 2 * P                                   # neighbours' slabs start P rows inside ours
        mine = torch.from_numpy(np.ascontiguousarray(sam[:, own0:own1]))
        padded = sharding.exchange_halo(mine, halo, None)
        lo = own0 - (halo if rank > 0 else 0)
        np.testing.assert_array_equal(padded.numpy(), sam[:, lo: own1 + (halo if rank < world - 1 else 0)])
        assert lo <= a and own1 + (halo if rank < world - 1 else 0) >= b      # slab + halo is covered

        # (2b) the persistent-buffer form bench.py --gpus N uses: input rows split in nearly equal blocks, the halo
        # rows of both stacks fetched in one send/recv group straight into the model's frame buffer, result slabs
        # gathered as padded tensors
        H = sam.shape[1]
        st_s = sharding.RowShardedStack(sam.shape[0], H, sam.shape[2], P, world, rank)
        st_r = sharding.RowShardedStack(ref.shape[0], H, ref.shape[2], P, world, rank)
        o0, o1 = st_s.own[rank]
        st_s.buf.fill_(float("nan")); st_r.buf.fill_(float("nan"))
        st_s.own_rows().copy_(torch.from_numpy(sam[:, o0:o1]))
        st_r.own_rows().copy_(torch.from_numpy(ref[:, o0:o1]))
        sharding.exchange_halos([st_s, st_r])
        na, nb = st_s.need[rank]
        assert (na, nb) == (a, b)
        np.testing.assert_array_equal(torch.stack(st_s.frames()).numpy(), sam[:, a:b])
        np.testing.assert_array_equal(torch.stack(st_r.frames()).numpy(), ref[:, a:b])
        assert st_s.halo_bytes() == ((o0 - na) + (nb - o1)) * sam.shape[2] * sam.shape[0] * 8
        sub = cpu_model.port.UMPAModelDF([f.numpy() for f in st_s.frames()], [f.numpy() for f in st_r.frames()],
                                         window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
        biggest = max(q1 - q0 for q0, q1 in st_s.out)
        pad = torch.zeros((biggest,) + sub["T"].shape[1:], dtype=torch.float64)
        pad[: r1 - r0] = torch.from_numpy(sub["T"])
        whole_T = sharding.gather_slabs(pad, n_out, dst=0)

        # (3) all_gather variant
        every = sharding.gather_rows({"err": res["err"]}, n_out, dst=None)
        assert every["err"].shape[0] == n_out

        if rank == 0:
            full = cpu_model.port.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
            for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls"):
                np.testing.assert_array_equal(whole[k], full[k], err_msg=k)
            np.testing.assert_array_equal(every["err"], full["err"])
            np.testing.assert_array_equal(whole_T.numpy(), full["T"])
            open(os.path.join(tmp, "ok_%d" % world), "w").write("ok")
        else:
            assert whole is None and whole_T is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharding_matches_unsharded(world, tmp_path):
    import torch.multiprocessing as mp
    from oracle import cpu_model
    cpu_model.native("port")            # build the checker once, before forking workers
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / ("ok_%d" % world)).exists()


def test_slab_bounds_partition():
    from umpa_amd.sharding import input_rows, slab_bounds
    for n in (1, 7, 8172, 2028):
        for world in (1, 2, 3, 8):
            edges = [slab_bounds(n, world, g) for g in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[g][1] == edges[g + 1][0] for g in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    # BASELINE config C4: 8172 output rows over 8 GPUs -> 4 x 1022 + 4 x 1021, 10 halo rows each side
    assert [slab_bounds(8172, 8, g)[1] - slab_bounds(8172, 8, g)[0] for g in range(8)] == [1022] * 4 + [1021] * 4
    assert input_rows(8172, 10, 8, 3) == (3 * 1022, 4 * 1022 + 20)


def test_projection_farm_failure_modes():
    """a farm whose workers die at start-up raises instead of spinning; ids may repeat; ROI / step are
    refused (the result slots hold full-extent maps)."""
    from oracle import cpu_model
    from umpa_amd.farm import ProjectionFarm
    from umpa_amd.synth import make_stack
    cpu_model.native("port")
    Nw, ms = 2, 3
    sam, ref, _ = make_stack(40, 44, 3, ms, df=True, seed=7, amplitude=1.0)
    t0 = time.time()
    with ProjectionFarm(ref, Nw, ms, devices=[None, None], worker=functools.partial(farm_cpu_worker.run, model=("no_such_module_anywhere", "X"))) as farm:
        with pytest.raises(RuntimeError, match="farm worker|exited"):
            dict(farm.map([(0, sam), (1, sam)], timeout=60.0))
    assert time.time() - t0 < 60.0
    with ProjectionFarm(ref, Nw, ms, devices=[None], worker=farm_cpu_worker.run) as farm:
        got = list(farm.map([(5, sam), (5, 0.5 * sam), (5, sam)], num_threads=1))       # the same id three times
        assert [g[0] for g in got] == [5, 5, 5]
        want = cpu_model.port.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
        assert sum(np.array_equal(g[1]["T"], want["T"]) for g in got) == 2
        with pytest.raises(ValueError, match="full-extent"):
            dict(farm.map([(0, sam)], step=2))
        assert dict(farm.map([(9, sam)], num_threads=1))[9]["T"].shape == want["T"].shape   # still usable afterwards


def test_projection_farm_keeps_its_books_when_a_projection_fails():
    """a projection that fails in its worker ends map() with ProjectionFailed -- AFTER the projections that
    were in flight beside it have been collected and handed out; the next map() on the same farm sees its own results only."""
    from oracle import cpu_model
    from umpa_amd.farm import ProjectionFailed, ProjectionFarm
    from umpa_amd.synth import make_stack
    cpu_model.native("port")
    Nw, ms = 2, 3
    sam, ref, _ = make_stack(40, 44, 3, ms, df=True, seed=7, amplitude=1.0)
    worker = functools.partial(farm_cpu_worker.run, fail_pids=(1,))
    with ProjectionFarm(ref, Nw, ms, devices=[None, None], worker=worker, depth=2) as farm:
        seen = []
        with pytest.raises(ProjectionFailed, match="projection 1 failed"):
            for pid, res in farm.map([(0, sam), (1, sam), (2, 0.5 * sam), (3, sam)], num_threads=1):
                seen.append(pid)
        assert 1 not in seen and all(w["inflight"] == 0 for w in farm._workers) and not farm._by_seq
        got = dict(farm.map([(10, sam), (11, 0.5 * sam)], num_threads=1))
        assert sorted(got) == [10, 11]
        want = cpu_model.port.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
        np.testing.assert_array_equal(got[10]["T"], want["T"])
        assert not np.array_equal(got[11]["T"], want["T"])


def test_projection_farm_on_cpu_workers():
    """umpa_amd.farm (config C5's pattern: independent projections, shared reference) with two CPU workers."""
    from oracle import cpu_model
    from umpa_amd.farm import ProjectionFarm
    from umpa_amd.synth import make_stack
    cpu_model.native("port")
    Nw, ms = 2, 3
    stacks = [make_stack(40, 44, 3, ms, df=True, seed=100 * p, amplitude=1.0) for p in range(3)]
    ref = stacks[0][1]
    sams = {p: np.ascontiguousarray(0.9 * stacks[p][0]) for p in range(3)}
    with ProjectionFarm(ref, Nw, ms, devices=[None, None], worker=farm_cpu_worker.run) as farm:
        got = dict(farm.map(sams.items(), num_threads=1))
    assert sorted(got) == [0, 1, 2]
    for p in range(3):
        want = cpu_model.port.UMPAModelDF(sams[p], ref, window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
        for k in ("f", "T", "dx", "dy", "df", "err"):
            np.testing.assert_array_equal(got[p][k], want[k])

    # the flat-field form of umpa_multi.py:133-145: uint16 counts, dark frame, one flat and one reference stack per
    # reference acquisition, nearest reference per projection number
    rng = np.random.default_rng(3)
    refs = np.stack([stacks[0][1], stacks[1][1]])
    dark = 90.0 + rng.uniform(0, 2, size=ref.shape)
    flats = 9000.0 * (1.0 + 0.05 * rng.standard_normal((2,) + ref.shape))
    ref_nums = [0, 9]
    raws = {p: np.rint(stacks[p % 3][0] * flats[0 if p < 5 else 1] + dark).astype(np.uint16) for p in (1, 4, 5, 8)}
    with ProjectionFarm(refs, Nw, ms, devices=[None], flats=flats, dark=dark, ref_nums=ref_nums, raw_dtype=np.uint16,
                        worker=farm_cpu_worker.run) as farm:
        got = dict(farm.map(raws.items(), num_threads=1))
    for p, raw in raws.items():
        r = 0 if p < 5 else 1
        sam = (raw.astype(np.float64) - dark) / flats[r]
        want = cpu_model.port.UMPAModelDF(sam, refs[r], window_size=Nw, max_shift=ms).match(quiet=True, num_threads=1)
        for k in ("f", "T", "dx", "dy", "df", "err"):
            np.testing.assert_array_equal(got[p][k], want[k])


def _gpu_worker(rank, world, port, tmp):
    """Two ranks, one process each, both on the one GPU of the test box (the 8-GPU run is the driver's): slabs are
    matched by the HIP path and gathered over gloo; rank 0 checks against the unsharded HIP result."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      UMPA_HIP_DEVICE="0")
    import torch.distributed as dist
    from umpa_amd import model, sharding
    from umpa_amd.synth import make_stack

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nw, ms = 3, 4
        sam, ref, _ = make_stack(150, 170, 4, ms, df=True, seed=11, amplitude=2.0)
        n_out = sam.shape[1] - 2 * (Nw + ms)
        keys = ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls")
        res, (r0, r1) = sharding.match_rows(model.UMPAModelDF, sam, ref, Nw, ms, world, rank)
        whole = sharding.gather_rows({k: res[k] for k in keys}, n_out, dst=0)
        if rank == 0:
            full = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms).match(quiet=True)
            for k in keys:
                np.testing.assert_array_equal(whole[k], full[k], err_msg=k)
            assert full["err"].mean() > 0.5
            open(os.path.join(tmp, "gpu_ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_row_sharding_on_the_gpu(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "gpu_ok").exists()


def _gpu_sharded_worker(rank, world, port, tmp, backend):
    """The path of bench.py --gpus N at a small size: the frames live row-sharded on the device(s) in persistent
    buffers, the halo rows travel between the ranks, the model borrows the device rows, the result slabs are gathered
    on rank 0 -- and must equal the unsharded HIP result bit for bit."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from umpa_amd import model, sharding
    from umpa_amd.synth import make_stack

    local = rank % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nw, ms = 3, 4
        P = Nw + ms
        sam, ref, _ = make_stack(190, 210, 4, ms, df=True, seed=21, amplitude=2.0)
        K, H, W = sam.shape
        n_out, N1 = H - 2 * P, W - 2 * P
        st_s = sharding.RowShardedStack(K, H, W, P, world, rank, device=dev)
        st_r = sharding.RowShardedStack(K, H, W, P, world, rank, device=dev)
        o0, o1 = st_s.own[rank]
        st_s.buf.fill_(float("nan")); st_r.buf.fill_(float("nan"))
        st_s.own_rows().copy_(torch.from_numpy(sam[:, o0:o1]))
        st_r.own_rows().copy_(torch.from_numpy(ref[:, o0:o1]))
        sharding.exchange_halos([st_s, st_r])
        m = model.UMPAModelDF(st_s.frames(), st_r.frames(), window_size=Nw, max_shift=ms, device=local)
        res = m.match(quiet=True)
        r0, r1 = st_s.out[rank]
        assert res["T"].shape == (r1 - r0, N1)
        biggest = max(b - a for a, b in st_s.out)
        whole = {}
        for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls"):
            pad = torch.zeros((biggest, N1), dtype=torch.from_numpy(res[k]).dtype, device=dev)
            pad[: r1 - r0] = torch.from_numpy(res[k]).to(dev)
            got = sharding.gather_slabs(pad, n_out, dst=0)
            if rank == 0:
                whole[k] = got.cpu().numpy()
        # the form bench.py --gpus N times: device-resident outputs, the slab matched in row pieces, the rows of a piece
        # sent to rank 0 from the library's callback while the next piece is matched
        import ctypes
        from umpa_amd import _lib
        N0 = r1 - r0
        values = torch.zeros((biggest, N1, 5), dtype=torch.float64, device=dev)
        err = torch.zeros((biggest, N1), dtype=torch.int32, device=dev)
        whole_v = torch.full((n_out, N1, 5), float("nan"), dtype=torch.float64, device=dev) if rank == 0 else None
        whole_e = torch.full((n_out, N1), -7, dtype=torch.int32, device=dev) if rank == 0 else None
        pending, seen = [], []

        def on_rows(lo, hi, _user):
            seen.append((lo, hi))
            if hi >= N0:
                hi = biggest
            pending.extend(sharding.send_rows_to([values, err], [whole_v, whole_e], n_out, lo, hi, dst=0))

        cb = _lib.ROWS_FN(on_rows)
        lib, h = m._lib, m._handle
        lib.check(lib.set_rows_callback(h, ctypes.cast(cb, ctypes.c_void_p), None, 32), "set_rows_callback")
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), 5, None, err.data_ptr(), None, 0.0, None, None, None,
                              _lib.F_DEVICE_IO, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.set_rows_callback(h, None, None, 0)
        lib.check(rc, "match_region")
        for w in pending:
            w.wait()
        torch.cuda.synchronize()
        assert len(seen) == -(-N0 // 32) and seen[0][0] == 0 and seen[-1][1] == N0

        if rank == 0:
            full = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms, device=local).match(quiet=True)
            for k in whole:
                np.testing.assert_array_equal(whole[k], full[k], err_msg=k)
            assert full["err"].mean() > 0.5
            wv, we = whole_v.cpu().numpy(), whole_e.cpu().numpy()
            np.testing.assert_array_equal(we, full["err"])
            for n, k in enumerate(("f", "T", "dx", "dy", "df")):
                np.testing.assert_array_equal(wv[:, :, n], full[k], err_msg="pieces " + k)
            open(os.path.join(tmp, "sharded_ok_" + backend), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_device_sharded_frames_on_one_gpu(tmp_path):
    """two ranks share the one GPU of the test box; the halo and the gather travel over gloo (staged through the host)"""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_sharded_worker, args=(2, port, str(tmp_path), "gloo"), nprocs=2, join=True)
    assert (tmp_path / "sharded_ok_gloo").exists()


@pytest.mark.gpu
def test_device_sharded_frames_over_rccl(tmp_path):
    """one rank per GPU, halo exchange and gather over RCCL: needs at least two GPUs (the driver's 8-GPU node)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL variant needs two")
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_gpu_sharded_worker, args=(2, port, str(tmp_path), "nccl"), nprocs=2, join=True)
    assert (tmp_path / "sharded_ok_nccl").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["pieces", "after"])
def test_bench_sharded_leg_rehearsal(tmp_path, gather):
    """bench.py --gpus 2 end to end at a reduced size (two ranks on one GPU, gloo): the JSON line of the C4 leg, with the
    gather overlapped piece by piece and with the gather after the match.  The leg verifies itself: the gathered maps
    across the slab boundary must equal an unsharded match of regenerated input rows bit for bit (`gpu_agrees`)."""
    import json
    import subprocess
    env = dict(os.environ, UMPA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", UMPA_BENCH_GATHER=gather)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--rows", "128", "--cols", "512"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rccl_world_size"] == 2
    assert d["config"]["output_pixels_total"] == (256 - 20) * (512 - 20)
    for k in ("halo_ms", "match_ms", "gather_ms"):
        assert d["config"][k] > 0
    assert d["value"] > 0 and d["roofline"]["kernel"] in ("corr_volume", "replay_walk", "prep_maps")
    assert ("pieces" in d["config"]["gather_mode"]) == (gather == "pieces")
    g = d["gpu_agrees"]
    assert g["parity"].startswith("bit-identical") and g["boundaries"][0]["ranks"] == [0, 1] and g["boundaries"][0]["pixels"] == 16 * 492


def _farm_on(devices, port_ns):
    from umpa_amd.farm import ProjectionFarm
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 3, 4, 4, 160
    sam0, ref, _ = make_stack(n, n + 24, K, ms, df=True, seed=60, amplitude=1.0)
    sams = {p: np.ascontiguousarray(np.roll(sam0, p % 2, axis=2)) for p in range(5)}    # two different sample stacks in turn
    with ProjectionFarm(ref, Nw, ms, devices=devices) as farm:
        res = dict(farm.map(sams.items()))
        assert len(farm._workers) == len(devices)
    assert sorted(res) == list(range(5))
    for p in range(5):
        o = port_ns.UMPAModelDF(sams[p], ref, window_size=Nw, max_shift=ms)
        o.debug = False
        want = o.match(quiet=True)
        np.testing.assert_array_equal(res[p]["err"], want["err"])
        ok = want["err"] == 1
        for k in ("T", "df"):
            np.testing.assert_allclose(res[p][k][ok], want[k][ok], rtol=1e-5)
        close = np.abs(res[p]["dx"] - want["dx"]) <= 1e-5 * np.maximum(1.0, np.abs(want["dx"]))
        assert (~close & ok).sum() <= max(2, int(0.002 * ok.sum()))      # (the farm's maps carry no debug arrays to classify with)


@pytest.mark.gpu
def test_projection_farm_two_workers_on_one_gpu(port_ns):
    """ProjectionFarm(devices=[0, 0]): two worker processes (each its own StreamingMatcher) share the one GPU of the test
    box -- the multi-device farm of BASELINE config C5 in every respect but the second card -- against the CPU oracle."""
    _farm_on([0, 0], port_ns)


@pytest.mark.gpu
def test_projection_farm_on_two_devices(port_ns):
    """ProjectionFarm(devices=[0, 1]): one worker process per GPU (umpa_multi.py:261-270's Pool of workers); needs two
    visible GPUs (the driver's 8-GPU node)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the two-device farm needs two")
    _farm_on([0, 1], port_ns)


def test_watchdog_ends_a_wedged_rank():
    """sharding.Watchdog: a phase that overruns its deadline ends the process with a non-zero exit code (a wedged rank must
    end the job); a phase that finishes in time does not; run in a child process, as bench.py uses it."""
    import subprocess
    prog = ("import sys, time; sys.path.insert(0, %r)\n"
            "from umpa_amd.sharding import Watchdog\n"
            "wd = Watchdog(rank=5)\n"
            "with wd.phase('quick', 5.0): time.sleep(0.05)\n"
            "print('quick phase done', flush=True)\n"
            "with wd.phase('halo exchange', 0.4): time.sleep(30)\n"
            "print('not reached')\n") % REPO
    t0 = time.time()
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3, (out.returncode, out.stdout, out.stderr)
    assert "quick phase done" in out.stdout and "not reached" not in out.stdout
    assert "rank 5" in out.stderr and "halo exchange" in out.stderr
    assert time.time() - t0 < 20
