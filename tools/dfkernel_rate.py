#!/usr/bin/env python3
"""Rate of the kernel-dark-field model (UMPAModelDFKernel, Model.cpp:997-1238) on one GPU: an E_dfkernel-type stack
(2 frames, Nw=2, max_shift=4, per-pixel blur a=c=0.1, b=0) at 512 x 512, host-array API.
UMPA_HIP_DFK_NO_REUSE=1 recomputes the 289-tap blur at every evaluation (round 1), the default fills the pixel's
blurred footprint once."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpa_amd import model
from umpa_amd.synth import make_stack

# `c2`: BASELINE config C2's parameters (2048 x 2048, 10 frames, Nw=5, max_shift=5) on a band of 32 full-width output rows
# -- the blur alone is 289 x 19 x 19 x 10 = 1.04 M FMAs per PIXEL there (4.3e12 for the whole image: minutes per match)
c2 = len(sys.argv) > 1 and sys.argv[1] == "c2"
n = 2048 if c2 else (int(sys.argv[1]) if len(sys.argv) > 1 else 512)
K, Nw, ms = (10, 5, 5) if c2 else (2, 2, 4)
sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=40, amplitude=1.5, order=1)
m = model.UMPAModelDFKernel(sam, ref, window_size=Nw, max_shift=ms)
m.debug = "ncalls"
N0, N1 = m.extent
kw = dict(quiet=True)
if c2:
    N0 = 32
    kw["ROI"] = ((1000, 1032, 1), (0, N1, 1))
abc = np.zeros((N0, N1, 3)); abc[..., 0] = 0.1; abc[..., 2] = 0.1
r = m.match(abc=abc, **kw)
reps = 1 if c2 else 3
t0 = time.perf_counter()
for _ in range(reps):
    r = m.match(abc=abc, **kw)
dt = (time.perf_counter() - t0) / reps
import ctypes
lib, h = m._lib, m._handle
lib.timing_enable(h, 1)
m.match(abc=abc, **kw)
lib.timing_enable(h, 0)
kern = {}
for q in range(lib.timing_collect(h)):
    nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
    lib.timing_read(h, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
    kern[nm.value.decode()] = round(tot.value, 3)
F = 2 * (Nw + ms - 1) + 1
print(json.dumps(dict(reuse=not os.environ.get("UMPA_HIP_DFK_NO_REUSE"), tiles=not os.environ.get("UMPA_HIP_DFK_NO_TILES"),
                      blur_fma_per_px=289 * F * F * K, blur_tfma_s=round(289.0 * F * F * K * N0 * N1 / dt / 1e12, 3), size=n, frames=K, Nw=Nw, max_shift=ms, output_px=N0 * N1,
                      ms=round(dt * 1e3, 2), mpx_s=round(N0 * N1 / dt / 1e6, 2), err_ok=round(float(r["err"].mean()), 4),
                      Ncalls_mean=round(float(r["debug_Ncalls"].mean()), 2), kernels_ms=kern)))
