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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, Nw, ms = 2, 2, 4
sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=40, amplitude=1.5, order=1)
m = model.UMPAModelDFKernel(sam, ref, window_size=Nw, max_shift=ms)
m.debug = "ncalls"
N0, N1 = m.extent
abc = np.zeros((N0, N1, 3)); abc[..., 0] = 0.1; abc[..., 2] = 0.1
r = m.match(abc=abc, quiet=True)
t0 = time.perf_counter()
for _ in range(3):
    r = m.match(abc=abc, quiet=True)
dt = (time.perf_counter() - t0) / 3
print(json.dumps(dict(reuse=not os.environ.get("UMPA_HIP_DFK_NO_REUSE"), size=n, frames=K, Nw=Nw, max_shift=ms, output_px=N0 * N1,
                      ms=round(dt * 1e3, 2), mpx_s=round(N0 * N1 / dt / 1e6, 2), err_ok=round(float(r["err"].mean()), 4),
                      Ncalls_mean=round(float(r["debug_Ncalls"].mean()), 2))))
