#!/usr/bin/env python3
"""
bench.py -- output-map Mpixels/s of the UMPA matching path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C1|C5]

One "step" is one pass of the hot path over one synthetic stack: every output pixel of the frame is matched.
Inputs are resident in HBM before the timed region starts, outputs stay in HBM (the PCIe-inclusive rate of the
host-array API is quoted in DESIGN.md, never here).

N = 1   BASELINE config C2 (2048x2048, 10 frames, Nw=5, max_shift=5, dark-field on): one `umpa_hip_match_region`
        (include/umpa_hip.h) per step.  `--config C5` times the step-scan farm shape instead (umpa_amd/farm.py).

N > 1   (launched by torch.distributed.run, one rank per GPU, RCCL) BASELINE config C4: ONE (N*1024) x 8192 image
        of 10 frames whose INPUT rows live row-sharded on the GPUs, 1024 rows per rank -- 8192 x 8192 at N = 8.
        A step is what the path does with such an image:
          1. halo exchange: every rank tops its block up with the boundary rows of its neighbours
             (sharding.exchange_halos: one ncclSend/ncclRecv group over xGMI, device tensors);
          2. match of the rank's output-row slab (1021/1022 rows at N = 8), in four row pieces;
          3. gather of the result slabs on rank 0: the rows of a piece start travelling (sharding.send_rows_to, one
             RCCL send/recv group per piece) as soon as its kernels are enqueued, overlapping the next piece.
        Per-GPU work is the same for every N (weak scaling); `value` counts the output pixels of the whole image.

Rank 0 prints ONE JSON line (see the driver contract), including
  "roofline":     the dominant kernel against the roofline that bounds it (fp64 FMA; the HBM figure beside it)
  "cpu_baseline": the reference C++ core (oracle/_ref, kind "reference") or this repo's C restatement (kind "port")
                  timed on the host cores on a bounded row sample, N = 1 only -- and the GPU maps of those rows are
                  checked against it with the parity rules of the test-suite (the run fails if they differ).
"""
import os

# Before anything loads an OpenMP runtime: the CPUs this process may run on (the runtime binds the initial thread to
# one place as soon as it starts, after which the mask reads as that one core), then the binding policy of the CPU
# baseline's thread team, which the runtime reads when it is loaded.
try:
    AFFINITY_AT_START = len(os.sched_getaffinity(0))
except Exception:
    AFFINITY_AT_START = None
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import argparse
import ctypes
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TF = 78.6            # fp64 vector = fp64 MFMA peak on MI355X: 256 CU x 64 FMA/clk x 2.4 GHz x 2 (public spec)
BLOCK_ROWS = 1024              # input rows per rank of the row-sharded image (8 x 1024 = BASELINE config C4)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, help="N=1: C2 (default), C1, C3, C5, C2mask / C2maskNoDF (C2's stack with a 95 %% random mask); N>1: C4")
    ap.add_argument("--rows", type=int, default=0, help="override frame height (debugging)")
    ap.add_argument("--cols", type=int, default=0, help="override frame width (debugging)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--force", choices=["auto", "direct", "tiled"], default="auto")
    return ap.parse_args()


def algorithmic_bytes(K, H, W, N0, N1, nparam):
    """compulsory HBM traffic of one match, fp64."""
    return 2 * K * H * W * 8 + (8 * nparam + 4) * N0 * N1


def collect_kernels(lib, h):
    """{name: (ms per launch, launches, fp64 FMAs of all launches)} from the HIP events recorded on the launch stream."""
    kernels = {}
    for q in range(lib.timing_collect(h)):
        name, tot, cnt, fma = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
        lib.timing_read(h, q, ctypes.byref(name), ctypes.byref(tot), ctypes.byref(cnt))
        lib.timing_fma(h, q, ctypes.byref(fma))
        kernels[name.value.decode()] = (tot.value / max(cnt.value, 1), cnt.value, fma.value)
    return kernels


def useful_fma(K, Nw, ms, N0, N1):
    """The table kernel's arithmetic without halos or padding: per output pixel and integer shift, K
    product FMAs and the two 1-D window filters of 2 Nw + 1 taps."""
    return float(K + 2 * (2 * Nw + 1)) * N0 * N1 * (2 * ms - 1) ** 2


def roofline(kernels, steps, abytes, config, useful=None, ms_per_step=None):
    """The dominant kernel against the roofline that bounds it.  The table kernel is fp64-FMA bound.
    `achieved` / `frac` price the ALGORITHMIC arithmetic of the restructured algorithm -- `useful`, per output pixel and integer
    shift K products + two (2 Nw + 1)-tap window filters, no halos, no padding -- x 2 flop / the kernel's HIP-event duration,
    against the 78.6 TFLOP/s fp64 peak.  What the kernel EXECUTES (halo and padding included, counted on the host from the
    launch geometry) sits beside it as `frac_executed`.  Models with masks have no closed count of useful arithmetic: there
    `frac` is the executed figure and `frac_executed` says the same.  The HBM figure the contract describes (the whole match's
    algorithmic bytes / the dominant kernel's duration, against 8 TB/s) sits under "hbm"; `traffic` = PMC bytes per launch.
    `step`: the whole step -- useful TFLOP/s and the counter traffic of all its kernels over `ms_per_step`."""
    if not kernels:
        return None
    dom = max(kernels, key=lambda k: kernels[k][0] * kernels[k][1])
    per_step = kernels[dom][1] / steps
    dur_ms = kernels[dom][0] * per_step
    traffic, traffic_all = None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp)).get(config, {})
            traffic = tj.get(dom)
            per = {k: tj[k] * kernels[k][1] / steps for k in kernels if isinstance(tj.get(k), (int, float))}
            traffic_all = sum(per.values()) if len(per) == len(kernels) else None
        except Exception:
            traffic = traffic_all = None
    gbs = abytes / (dur_ms * 1e-3) / 1e9
    hbm = dict(achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 5),
               algorithmic_bytes=abytes)
    fma = kernels[dom][2] / steps
    out = dict(kernel=dom, kernel_ms=round(dur_ms, 4), traffic=traffic,
               kernels_ms={k: round(v[0] * v[1] / steps, 4) for k, v in kernels.items()}, hbm=hbm)
    if fma > 0:
        tf_exec = 2.0 * fma / (dur_ms * 1e-3) / 1e12
        have_useful = bool(useful) and dom in ("corr_volume", "corr_march")
        tf = 2.0 * useful / (dur_ms * 1e-3) / 1e12 if have_useful else tf_exec
        out.update(bound="fp64_fma", achieved=round(tf, 3), peak=FP64_PEAK_TF, unit="TFLOP/s", frac=round(tf / FP64_PEAK_TF, 5),
                   frac_executed=round(tf_exec / FP64_PEAK_TF, 5),
                   priced="algorithmic (useful) fp64 FMAs" if have_useful else "executed fp64 issue slots (no closed useful count for this model)")
        out["fp64_fma"] = dict(useful_fma_per_match=useful if have_useful else None,
                               executed_fma_per_launch=fma / max(per_step, 1), executed_tflops=round(tf_exec, 3),
                               peak_tflops=FP64_PEAK_TF,
                               note="useful: (K + 2 (2 Nw + 1)) (2 ms - 1)^2 N0 N1 FMAs per match; executed: fp64 issue slots of the tiled "
                                    "path's dominant kernel (corr_volume: all FMAs incl. halo and padding; corr_masked: an FMA, multiply "
                                    "or add each one slot), counted on the host from the launch geometry and the device's count of "
                                    "(tile, pass) units it computed; x2 flop, / the kernel's HIP-event duration")
    else:
        out.update(bound="hbm", achieved=hbm["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=hbm["frac"])
    if ms_per_step:
        st = dict(ms=round(ms_per_step, 4),
                  algorithmic_gbs=round(abytes / (ms_per_step * 1e-3) / 1e9, 2),
                  algorithmic_hbm_frac=round(abytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5))
        if useful:
            st["useful_tflops"] = round(2.0 * useful / (ms_per_step * 1e-3) / 1e12, 3)
            st["useful_fp64_frac"] = round(st["useful_tflops"] / FP64_PEAK_TF, 5)
        if traffic_all:
            st["counter_traffic_bytes"] = int(traffic_all)
            st["counter_gbs"] = round(traffic_all / (ms_per_step * 1e-3) / 1e9, 1)
            st["counter_over_algorithmic"] = round(traffic_all / abytes, 2)
        out["step"] = st
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from umpa_amd import _lib, model
    from umpa_amd.synth import CONFIGS, make_stack

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU over RCCL ("nccl"); UMPA_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal
    backend = os.environ.get("UMPA_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never shows up ends the job after this long instead of holding it (the default is 10 minutes per
        # collective under RCCL); the per-phase deadlines of the steps are sharding.Watchdog's
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("UMPA_BENCH_PG_TIMEOUT", "180")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
        else:
            dist.init_process_group(backend=backend, timeout=pg_timeout)
    assert world == args.gpus or world == 1 and args.gpus == 1, \
        "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if world > 1:
        return sharded(args, world, rank, local, dev, backend)

    config = args.config or "C2"
    if config == "C5":
        return farm_bench(args, local)
    masked = config.startswith("C2mask")
    cfg = dict(CONFIGS["C2" if masked else config])
    if config == "C2maskNoDF":
        cfg["df"] = False
    if args.rows:
        cfg["H"] = args.rows
    if args.cols:
        cfg["W"] = args.cols
    H, W, K, Nw, ms, df = cfg["H"], cfg["W"], cfg["K"], cfg["Nw"], cfg["max_shift"], cfg["df"]
    nparam = 5 if df else 4

    t_gen = time.time()
    sam, ref, _ = make_stack(H, W, K, ms, df=df, seed=0)
    t_gen = time.time() - t_gen
    cls = model.UMPAModelDF if df else model.UMPAModelNoDF
    mask = (np.random.default_rng(1).uniform(size=sam.shape) < 0.95).astype(np.float64) if masked else None
    m = cls(sam, ref, mask_list=mask, window_size=Nw, max_shift=ms, device=local)      # H2D happens here, outside the timed region
    lib, h = m._lib, m._handle
    N0, N1 = m.extent
    npx = N0 * N1
    values = torch.zeros((N0, N1, nparam), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    ncalls = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    flags = _lib.F_DEVICE_IO | {"auto": 0, "direct": _lib.F_FORCE_DIRECT, "tiled": _lib.F_FORCE_TILED}[args.force]
    stream = torch.cuda.current_stream().cuda_stream
    cover, thr = None, 0.0
    if masked:                                              # the coverage map and threshold of model.pyx:427-431, resident like the inputs
        cm = m.coverage()
        cover, thr = torch.from_numpy(cm).to(dev), .1 * float(cm.max()) / K

    def step(with_ncalls=False):
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), nparam, None, err.data_ptr(),
                              cover.data_ptr() if cover is not None else None, thr, None, None,
                              ncalls.data_ptr() if with_ncalls else None, flags,
                              ctypes.c_void_p(stream))
        lib.check(rc, "match_region")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.timing_enable(h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.timing_enable(h, 0)
    kernels = collect_kernels(lib, h)
    path = {1: "direct", 2: "tiled", 3: "direct-staged"}.get(lib.last_path(h), "?")
    if masked and path == "tiled":
        path = "tiled (corr_masked + replay_cost)"

    step(with_ncalls=True)                                  # untimed: evaluation-count statistics of this dataset
    torch.cuda.synchronize()
    st4 = (ctypes.c_double * 4)()
    lib.check(lib.last_stats(h, st4), "last_stats")         # on-demand passes of the shift table (umpa_ondemand.h)
    nc = ncalls.cpu().numpy()
    err_h = err.cpu().numpy()

    # (the PMC traffic figures belong to a geometry: an overridden frame shape has none, except one rank's slab of C4)
    tkey = config if not (args.rows or args.cols) else ("C4slab" if (config, args.rows, args.cols) == ("C2", 1042, 8192) else "")
    roof = roofline(kernels, args.steps, algorithmic_bytes(K, H, W, N0, N1, nparam), tkey,
                    useful=None if mask is not None else useful_fma(K, Nw, ms, N0, N1), ms_per_step=dt / args.steps * 1e3)
    cpu = None
    if not args.no_cpu:
        cpu = cpu_baseline(m, sam, ref, Nw, ms, df, N0, N1, mask=mask)
    out = {
        "metric": "Mpixels/s (output map) at Nw=%d, max_shift=%d, %d frames" % (Nw, ms, K),
        "value": round(npx * args.steps / dt / 1e6, 3),
        "unit": "Mpx/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dx%d, %d frames, Nw=%d, max_shift=%d, dark-field %s" % (
                       config, H, W, K, Nw, ms, "on" if df else "off"),
                   "mask": "95 % random 0/1 mask per frame (model.pyx:257-262)" if masked else None,
                   "output_pixels_per_gpu": npx, "kernel_path": path,
                   "Ncalls_mean": round(float(nc.mean()), 3), "Ncalls_p99": int(np.percentile(nc, 99)),
                   "err_ok_fraction": round(float(err_h.mean()), 5), "parallelism": "1 GPU",
                   "table_passes_computed": int(st4[0]), "table_passes_exhaustive": int(st4[1]),
                   "pixels_replayed_twice": int(st4[2]), "tiles_repaired": int(st4[3]),
                   "input_generation_s": round(t_gen, 1)},
        "roofline": roof,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------
# N > 1: BASELINE config C4, one image row-sharded over the GPUs, RCCL halo exchange + gather
# ------------------------------------------------------------------------------------------------
def sharded(args, world, rank, local, dev, backend):
    import torch
    import torch.distributed as dist
    from umpa_amd import _lib, model, sharding
    from umpa_amd.synth import CONFIGS, make_block

    cfg = dict(CONFIGS["C4"])
    K, Nw, ms, df = cfg["K"], cfg["Nw"], cfg["max_shift"], cfg["df"]
    # which GPU every rank sits on (stderr, one line per rank; rank 0 puts the list into the JSON line), and a watchdog:
    # a rank wedged in the halo exchange, the match or the gather ends with exit code 3 instead of hanging the run
    me = "rank %d/%d local %d: %s" % (rank, world, local, sharding.describe_device(local))
    print("[bench] " + me, file=sys.stderr, flush=True)
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    wd = sharding.Watchdog(rank=rank)
    phase_s = float(os.environ.get("UMPA_BENCH_PHASE_TIMEOUT", "120"))
    H = world * (args.rows or BLOCK_ROWS)
    W = args.cols or cfg["W"]
    P = Nw + ms
    nparam = 5 if df else 4
    n_out, N1 = H - 2 * P, W - 2 * P
    coll = dev if backend == "nccl" else torch.device("cpu")        # where the timing all-reduce lives

    # this rank's block of input rows is produced on its GPU (here: generated on the host and uploaded once)
    st_s = sharding.RowShardedStack(K, H, W, P, world, rank, device=dev)
    st_r = sharding.RowShardedStack(K, H, W, P, world, rank, device=dev)
    t_gen = time.time()
    sam_b, ref_b = make_block(st_s.own[rank], H, W, K, ms, df=df, seed=100 * rank)
    st_s.own_rows().copy_(torch.from_numpy(sam_b))
    st_r.own_rows().copy_(torch.from_numpy(ref_b))
    del sam_b, ref_b
    t_gen = time.time() - t_gen
    sharding.exchange_halos([st_s, st_r])                   # the model wants valid rows at creation; timed again per step
    cls = model.UMPAModelDF if df else model.UMPAModelNoDF
    m = cls(st_s.frames(), st_r.frames(), window_size=Nw, max_shift=ms, device=local)   # borrows the device rows
    lib, h = m._lib, m._handle
    r0, r1 = st_s.out[rank]
    N0 = r1 - r0
    assert tuple(m.extent) == (N0, N1), (m.extent, N0, N1)
    biggest = max(b - a for a, b in st_s.out)
    values = torch.zeros((biggest, N1, nparam), dtype=torch.float64, device=dev)
    err = torch.zeros((biggest, N1), dtype=torch.int32, device=dev)
    ncalls = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    whole_v = torch.empty((n_out, N1, nparam), dtype=torch.float64, device=dev) if rank == 0 else None
    whole_e = torch.empty((n_out, N1), dtype=torch.int32, device=dev) if rank == 0 else None
    flags = _lib.F_DEVICE_IO | {"auto": 0, "direct": _lib.F_FORCE_DIRECT, "tiled": _lib.F_FORCE_TILED}[args.force]
    stream = torch.cuda.current_stream().cuda_stream
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]

    # The slab is matched in row pieces; as soon as the kernels of a piece are enqueued its rows start travelling
    # to rank 0 (one RCCL send/recv group per piece, behind the piece's kernels in stream order), so that the gather
    # overlaps the matching of the next piece instead of following the whole match.  UMPA_BENCH_GATHER=after: one gather
    # of the padded slabs after the match (sharding.gather_slabs), for comparison.
    overlap = os.environ.get("UMPA_BENCH_GATHER", "pieces") != "after"
    # six pieces of 192 rows at C4's 1022-row slabs: the last one (62 rows) is what the gather cannot overlap; the pieces cost the
    # match 0.05 ms (tools/piece_rate.py: 1 piece 5.24 ms, 4: 5.27, 6: 5.32, 8: 5.42, 12: 5.59)
    npieces = max(1, int(os.environ.get("UMPA_BENCH_PIECES", "6")))
    piece_rows = ((-(-biggest // npieces)) + 31) // 32 * 32          # the same on every rank
    pending = []

    def on_rows(lo, hi, _user):
        if hi >= N0:
            hi = biggest                                              # last piece: slabs differ by a row
        pending.extend(sharding.send_rows_to([values, err], [whole_v, whole_e], n_out, lo, hi, dst=0))

    cb = _lib.ROWS_FN(on_rows)

    def step(marks=None, with_ncalls=False):
        with wd.phase("step (halo exchange, match, gather)", phase_s):
            _step(marks, with_ncalls)

    def _step(marks=None, with_ncalls=False):
        nonlocal overlap
        if marks: marks[0].record()
        sharding.exchange_halos([st_s, st_r])
        if marks: marks[1].record()
        if overlap:
            lib.check(lib.set_rows_callback(h, ctypes.cast(cb, ctypes.c_void_p), None, piece_rows), "set_rows_callback")
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), nparam, None, err.data_ptr(),
                              None, 0.0, None, None, ncalls.data_ptr() if with_ncalls else None, flags,
                              ctypes.c_void_p(stream))
        if overlap:
            lib.set_rows_callback(h, None, None, 0)
        lib.check(rc, "match_region")
        if marks: marks[2].record()
        if overlap:
            for w in pending:
                w.wait()
            del pending[:]
        else:
            sharding.gather_slabs(values, n_out, dst=0, out=whole_v)
            sharding.gather_slabs(err, n_out, dst=0, out=whole_e)
        if marks: marks[3].record()

    def fence():
        with wd.phase("barrier", phase_s):
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # self-check of the piecewise gather on the first (untimed) step: every row of the whole maps must have arrived on
    # rank 0; if not, every rank falls back to the gather after the match
    if rank == 0:
        whole_e.fill_(-7)
    step()
    fence()
    good = torch.tensor([1 if (rank != 0 or bool((whole_e != -7).all().item())) else 0], dtype=torch.int32, device=coll)
    dist.all_reduce(good, op=dist.ReduceOp.MIN)
    if overlap and int(good.item()) == 0:
        overlap = False
        if rank == 0:
            print("bench.py: the piecewise gather left rows unwritten; using the gather after the match", file=sys.stderr)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    fence()
    lib.timing_enable(h, 1)
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(ev[s])
    fence()
    dt = time.perf_counter() - t0
    lib.timing_enable(h, 0)
    phases = np.array([[ev[s][q].elapsed_time(ev[s][q + 1]) for q in range(3)] for s in range(args.steps)]).mean(0)
    tmax = torch.tensor([dt] + list(phases), dtype=torch.float64, device=coll)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0].item())
    kernels = collect_kernels(lib, h)
    path = {1: "direct", 2: "tiled", 3: "direct-staged"}.get(lib.last_path(h), "?")
    step(with_ncalls=True)
    torch.cuda.synchronize()

    if rank == 0:
        nc = ncalls.cpu().numpy()
        err_all = whole_e.cpu().numpy()
        npx = n_out * N1
        agrees = None if args.no_cpu else verify_slab_boundaries(
            world, H, W, K, Nw, ms, df, P, N1, nparam, st_s, whole_v, whole_e, cls, local)
        name = "C4" if (world == 8 and H == 8192 and W == 8192) else "C4-type"
        roof = roofline(kernels, args.steps, algorithmic_bytes(K, N0 + 2 * P, W, N0, N1, nparam), "C4slab",
                        useful=useful_fma(K, Nw, ms, N0, N1), ms_per_step=dt / args.steps * 1e3)   # (rank 0's slab over the whole step)
        out = {
            "metric": "Mpixels/s (output map) at Nw=%d, max_shift=%d, %d frames" % (Nw, ms, K),
            "value": round(npx * args.steps / dt / 1e6, 3),
            "unit": "Mpx/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d frames, Nw=%d, max_shift=%d, dark-field %s; input rows sharded %d per GPU, "
                                   "halo exchange + slab match + gather on rank 0 per step (N GPUs match ONE (N x %d) x %d image: "
                                   "N = 8 is BASELINE config C4's 8192 x 8192)" % (
                                       name, H, W, K, Nw, ms, "on" if df else "off", H // world, H // world, W),
                       "output_pixels_total": npx, "output_pixels_per_gpu": N0 * N1, "kernel_path": path,
                       "parallelism": "rows x%d" % world, "comm_backend": "rccl" if backend == "nccl" else backend,
                       "rccl_world_size": dist.get_world_size(), "devices": everyone,
                       "halo_ms": round(float(tmax[1]), 4), "match_ms": round(float(tmax[2]), 4),
                       "gather_ms": round(float(tmax[3]), 4),
                       "gather_mode": "row pieces of %d rows sent while the next piece is matched; gather_ms is what is left "
                                      "after the match" % piece_rows if overlap else "after the match",
                       "halo_bytes_received_per_rank": st_s.halo_bytes() + st_r.halo_bytes(),
                       "gather_bytes_to_rank0": int((biggest * N1 * (nparam * 8 + 4)) * (world - 1)),
                       "Ncalls_mean_rank0": round(float(nc.mean()), 3),
                       "err_ok_fraction": round(float(err_all.mean()), 5),
                       "input_generation_s": round(t_gen, 1)},
            "roofline": roof,
            "cpu_baseline": None,
            "gpu_agrees": agrees,
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def verify_slab_boundaries(world, H, W, K, Nw, ms, df, P, N1, nparam, st, whole_v, whole_e, cls, local, half=8):
    """Rank 0, after the timed steps: the maps it gathered must be RIGHT, not just there.  Around (up to) two slab
    boundaries the input rows are generated again on the host (make_block of the two ranks that own them -- the inputs
    never left their GPUs otherwise), matched UNSHARDED by a model of their own on this GPU, and the gathered maps of
    the 2 * half output rows across the boundary must equal that match bit for bit (pixels are independent: sharding,
    halo exchange and gather may not change a bit).  Raises on a mismatch; returns what was compared."""
    import torch
    from umpa_amd.synth import make_block
    bounds = sorted(set([0, world - 2])) if world > 1 else []
    checked = []
    for r in bounds:
        b = st.out[r][1]                                              # first output row of rank r + 1's slab
        lo_out, hi_out = max(0, b - half), min(whole_e.shape[0], b + half)
        lo_in, hi_in = lo_out, hi_out + 2 * P                         # image rows those output rows read
        blocks = []
        for q in (r, r + 1):
            a0, a1 = st.own[q]
            if a1 <= lo_in or a0 >= hi_in:
                continue
            sb, rb = make_block(st.own[q], H, W, K, ms, df=df, seed=100 * q)
            c0, c1 = max(lo_in, a0) - a0, min(hi_in, a1) - a0
            blocks.append((np.ascontiguousarray(sb[:, c0:c1]), np.ascontiguousarray(rb[:, c0:c1])))
            del sb, rb
        sam = np.ascontiguousarray(np.concatenate([x[0] for x in blocks], axis=1))
        ref = np.ascontiguousarray(np.concatenate([x[1] for x in blocks], axis=1))
        assert sam.shape[1] == hi_in - lo_in, (sam.shape, lo_in, hi_in)
        m = cls(sam, ref, window_size=Nw, max_shift=ms, device=local)
        m.debug = False
        got = m.match(quiet=True)
        keys = ("f", "T", "dx", "dy") + (("df",) if df else ())
        gath_v = whole_v[lo_out:hi_out].cpu().numpy()
        gath_e = whole_e[lo_out:hi_out].cpu().numpy()
        if not np.array_equal(gath_e, got["err"]):
            raise AssertionError("gathered err map differs from an unsharded match across the boundary of ranks %d|%d" % (r, r + 1))
        for n, k in enumerate(keys):
            if not np.array_equal(gath_v[:, :, n], got[k], equal_nan=True):
                raise AssertionError("gathered %s map differs from an unsharded match across the boundary of ranks %d|%d" % (k, r, r + 1))
        checked.append(dict(ranks=[r, r + 1], output_rows=[int(lo_out), int(hi_out)], pixels=int((hi_out - lo_out) * N1)))
    return dict(parity="bit-identical to an unsharded match of regenerated input rows", boundaries=checked)


# ------------------------------------------------------------------------------------------------
# --config C5: the step-scan farm shape on one GPU (32 projections against one reference stack)
# ------------------------------------------------------------------------------------------------
def farm_bench(args, local):
    from umpa_amd import farm
    if args.no_cpu:
        print(json.dumps(farm.bench_c5(device=local, steps=args.steps, warmup=args.warmup)), flush=True)
        return
    # one projection of the series against the CPU checker: a band of full-width rows of its maps, held to the parity bar
    line, kept = farm.bench_c5(device=local, steps=args.steps, warmup=args.warmup, keep=7)
    from oracle import cpu_model, parity
    kind = "reference" if cpu_model.have_ref() else "port"
    ns = cpu_model.ref if kind == "reference" else cpu_model.port
    info = host_info()
    ncpu = usable_cpus(info)
    cm = ns.UMPAModelDF(np.ascontiguousarray(kept["sam"]), np.ascontiguousarray(kept["ref"]), window_size=kept["Nw"], max_shift=kept["ms"])
    cm.debug = True
    N0, N1 = cm.extent
    rows = min(N0, max(64, 4 * ncpu))
    r0 = (N0 - rows) // 2
    t = time.perf_counter()
    want = cm.match(ROI=((r0, r0 + rows, 1), (0, N1, 1)), num_threads=ncpu, quiet=True)
    dt = time.perf_counter() - t
    # the streamed maps carry no debug arrays: the same projection through a model of its own (debug arrays on) must give
    # the streamed maps bit for bit, and is then held to the parity bar against the CPU checker
    from umpa_amd import model
    gm = model.UMPAModelDF(np.ascontiguousarray(kept["sam"]), np.ascontiguousarray(kept["ref"]), window_size=kept["Nw"],
                           max_shift=kept["ms"], device=local)
    got = gm.match(ROI=((r0, r0 + rows, 1), (0, N1, 1)), quiet=True)
    for k in ("f", "T", "dx", "dy", "df", "err"):
        assert np.array_equal(kept["res"][k][r0:r0 + rows], got[k], equal_nan=True), "streamed map %s differs from a direct match" % k
    st = parity.assert_parity(got, want, kept["ms"], "bench C5 projection 7")
    line["cpu_baseline"] = dict(value=round(rows * N1 / dt / 1e6, 5), unit="Mpx/s", threads=ncpu, usable_cpus=ncpu, cores=ncpu, kind=kind,
                                sample="projection 7 of the series: %d full-width output rows (%d px) in %.1f s, OpenMP dynamic over rows as "
                                       "model.pyx:476-478" % (rows, rows * N1, dt), host=info,
                                gpu_agrees=dict(rows=rows, parity="pass", ok_pixels=st["ok"], unconverged_newton_pixels=st["unconverged"]))
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------
# cpu_baseline: the CPU checker timed on the host cores, and the GPU result checked against it
# ------------------------------------------------------------------------------------------------
def host_info():
    info = {}
    try:
        info["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception:
        info["cgroup_cpu_max"] = None
    info["os_cpu_count"] = os.cpu_count()
    info["affinity"] = AFFINITY_AT_START
    try:
        import subprocess
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            key = line.split(":")[0].strip()
            if key in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)"):
                info[key] = line.split(":", 1)[1].strip()
    except Exception:
        pass
    return info


def usable_cpus(info):
    n = info.get("affinity") or info.get("os_cpu_count") or 1
    q = info.get("cgroup_cpu_max")
    if q and not q.startswith("max"):
        try:
            quota, period = q.split()
            n = max(1, min(n, int(float(quota) / float(period))))
        except Exception:
            pass
    return n


def cpu_baseline(m, sam, ref, Nw, ms, df, N0, N1, per_run_s=2.5, mask=None):
    """The CPU checker (the reference C++ core where oracle/_ref exists) on bounded row samples of the SAME stack:
    a thread sweep with OMP_PROC_BIND=spread / OMP_PLACES=cores on first-touch-parallel copies of the inputs.
    `value` is the best of the sweep, `value_at_reference_default` the reference's default thread count
    (cpu_count()//2, model.pyx:378).  The GPU maps of the largest sample are then held to the parity bar."""
    from oracle import cpu_model, parity
    kind = "reference" if cpu_model.have_ref() else "port"
    ns = cpu_model.ref if kind == "reference" else cpu_model.port
    info = host_info()
    ncpu = usable_cpus(info)
    default_threads = max(1, (os.cpu_count() or 2) // 2)               # model.pyx:378
    # inputs re-homed by a parallel first touch (static partition over all threads)
    port = cpu_model.native("port").lib
    port.umpaor_parallel_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    port.umpaor_parallel_copy.restype = None

    def rehome(a):
        b = np.empty_like(a)
        port.umpaor_parallel_copy(b.ctypes.data, a.ctypes.data, a.size, ncpu)
        return b

    sam_c, ref_c = rehome(sam), rehome(ref)
    cls = ns.UMPAModelDF if df else ns.UMPAModelNoDF
    cm = cls(sam_c, ref_c, mask_list=mask, window_size=Nw, max_shift=ms)
    cm.debug = True

    def run(rows, threads):
        t = time.perf_counter()
        r = cm.match(ROI=((0, rows, 1), (0, N1, 1)), num_threads=threads, quiet=True)
        return time.perf_counter() - t, r

    # ~0.1 s per row and thread at C2 (0.02 Mpx/s per thread): size each run for ~per_run_s
    t_probe, _ = run(min(N0, 8), min(8, ncpu))
    rate_1 = min(N0, 8) * N1 / max(t_probe, 1e-3) / min(8, ncpu)      # px/s per thread, rough
    sweep, best, keep = [], None, None
    # thread counts beyond the usable CPUs (the reference's default cpu_count()//2 = 128 on a 16-CPU quota, say) are
    # oversubscription: they are timed too, on samples sized for the CPUs that really run
    cand = sorted(set(t for t in (8, 32, 64, 128, 256, default_threads, ncpu) if 1 <= t <= (os.cpu_count() or 8)))
    for th in cand:
        rows = int(min(N0, max(th, 16, per_run_s * rate_1 * min(th, ncpu) / N1)))
        dt, r = run(rows, th)
        rate = rows * N1 / dt / 1e6
        sweep.append(dict(threads=th, rows=rows, seconds=round(dt, 2), mpx_s=round(rate, 5)))
        if best is None or rate > best[0]:
            best = (rate, th, rows, dt)
        if keep is None or rows > keep[0]:
            keep = (rows, r)
    at_default = next((s["mpx_s"] for s in sweep if s["threads"] == default_threads), None)

    # the GPU result on the rows of the largest sample, through the host API with the debug maps, held to the
    # parity bar of the test-suite on ALL maps (err, Ncalls bit-exact; T, df, dx, dy, f to 1e-5)
    rows, want = keep
    m.debug = True
    got = m.match(ROI=((0, rows, 1), (0, N1, 1)), quiet=True)
    stats = parity.assert_parity(got, want, ms, "bench sample")       # raises (and fails the run) on a mismatch
    okpx = want["err"] == 1                                           # (none at all in C1's degenerate max_shift = 2)
    dT = float(np.max(np.abs(got["T"] - want["T"])[okpx])) if okpx.any() else 0.0
    return dict(value=round(best[0], 5), unit="Mpx/s", cores=min(best[1], ncpu), threads=best[1], kind=kind,
                sample="thread sweep on first-touch copies of the C-contiguous stacks, OMP_PROC_BIND=%s OMP_PLACES=%s; best: "
                       "first %d output rows (%d px) in %.1f s; OpenMP dynamic over rows as model.pyx:476-478" % (
                           os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_PLACES"), best[2], best[2] * N1, best[3]),
                value_at_reference_default=at_default, reference_default_threads=default_threads,
                sweep=sweep, host=info, usable_cpus=ncpu,
                gpu_agrees=dict(rows=rows, parity="pass", ok_pixels=stats["ok"], unconverged_newton_pixels=stats["unconverged"],
                                max_abs_dT=dT))


if __name__ == "__main__":
    main()
