#!/usr/bin/env python3
"""Rate of masked models on one GPU, device-resident: BASELINE config C2's stack with a 95 % random mask, dark-field
and plain model, as the library routes it (the tiled path: corr_masked + replay_cost) and forced onto the general kernel."""
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
mask = (rng.uniform(size=sam.shape) < 0.95).astype(np.float64)
out = {}
for df in (True, False):
    cls = model.UMPAModelDF if df else model.UMPAModelNoDF
    m = cls(sam, ref, mask_list=mask, window_size=Nw, max_shift=ms)
    lib, h = m._lib, m._handle
    N0, N1 = m.extent
    np_ = 5 if df else 4
    dev = torch.device("cuda", 0)
    values = torch.zeros((N0, N1, np_), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    cm = m.coverage()
    cover = torch.from_numpy(cm).to(dev)
    thr = .1 * cm.max() / K
    for tag, flags, reps in (("auto", 0, 3), ("general", _lib.F_FORCE_DIRECT, 1)):
        if tag == "general" and len(sys.argv) > 2 and sys.argv[2] == "fast":
            continue
        def step():
            rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), np_, None, err.data_ptr(),
                                  cover.data_ptr(), float(thr), None, None, None,
                                  _lib.F_DEVICE_IO | flags, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            lib.check(rc, "match_region")
        step(); torch.cuda.synchronize()
        lib.timing_enable(h, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lib.timing_enable(h, 0)
        kern = {}
        for q in range(lib.timing_collect(h)):
            name, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
            lib.timing_read(h, q, ctypes.byref(name), ctypes.byref(tot), ctypes.byref(cnt))
            kern[name.value.decode()] = round(tot.value / reps, 3)
        key = "%s + 95%% mask | %s" % ("DF" if df else "NoDF", tag)
        out[key] = dict(ms=round(dt * 1e3, 2), mpx_s=round(N0 * N1 / dt / 1e6, 1), path=lib.last_path(h),
                        err_ok=round(float(err.float().mean()), 4), kernels_ms=kern)
        print(key, out[key], flush=True)
print(json.dumps(out))
