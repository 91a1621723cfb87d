#!/usr/bin/env python3
"""Rate of the general (non-tiled) path on one GPU, device-resident: BASELINE config C2's stack (a) forced onto the
general kernels, (b) with a 95 % random mask, (c) as a 2 x 2 sample-stepping stack (four positions, 2 or 3 frames each).
Each as the library routes it ("auto": the tiled path where it applies -- for the stepping stack on the rectangle every
frame contributes to, path 4), with the staged general kernel (windows out of LDS) and with the plain one (through L1)."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umpa_amd import _lib, model
from umpa_amd.synth import make_stack

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K, Nw, ms = 10, 5, 5
sam, ref, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)
rng = np.random.default_rng(1)
cases = {
    "C2 stack, general kernels": dict(sam=sam, ref=ref),
    "C2 stack + 95% mask": dict(sam=sam, ref=ref, mask_list=(rng.uniform(size=sam.shape) < 0.95).astype(np.float64)),
    "2x2 sample stepping (4 positions)": dict(
        sam=[np.ascontiguousarray(sam[k, :H - 64, :W - 64]) for k in range(K)],
        ref=[np.ascontiguousarray(ref[k, :H - 64, :W - 64]) for k in range(K)],
        pos_list=[np.array([64 * ((k // 2) % 2), 64 * (k % 2)]) for k in range(K)]),
}
out = {}
for name, c in cases.items():
    kw = {k: v for k, v in c.items() if k in ("mask_list", "pos_list")}
    m = model.UMPAModelDF(c["sam"], c["ref"], window_size=Nw, max_shift=ms, **kw)
    lib, h = m._lib, m._handle
    N0, N1 = m.extent
    dev = torch.device("cuda", 0)
    values = torch.zeros((N0, N1, 5), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    cover = None
    thr = 0.0
    if not m._trivial_coverage():
        cm = m.coverage()
        cover = torch.from_numpy(cm).to(dev)
        thr = .1 * cm.max() / K
    for tag, flags in (("auto", 0), ("staged", _lib.F_FORCE_DIRECT), ("plain", _lib.F_FORCE_DIRECT | _lib.F_FORCE_PLAIN_DIRECT)):
        def step():
            rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), 5, None, err.data_ptr(),
                                  cover.data_ptr() if cover is not None else None, float(thr), None, None, None,
                                  _lib.F_DEVICE_IO | flags, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            lib.check(rc, "match_region")
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        out["%s | %s" % (name, tag)] = dict(ms=round(dt * 1e3, 2), mpx_s=round(N0 * N1 / dt / 1e6, 1), path=lib.last_path(h),
                                            err_ok=round(float(err.float().mean()), 4))
        print(name, tag, out["%s | %s" % (name, tag)], flush=True)
print(json.dumps(out))
