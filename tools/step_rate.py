import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from umpa_amd import _lib, model
from umpa_amd.synth import make_stack
H = W = 2048
K, Nw, ms = 10, 5, 5
sam, ref, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)
m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
lib, h = m._lib, m._handle
E0, E1 = m.extent
dev = torch.device("cuda", 0)
for step in (3, 4, 5, 6):
    N0, N1 = (E0 + step - 1) // step, (E1 + step - 1) // step
    values = torch.zeros((N0, N1, 5), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    for tag, flags in (("auto", 0), ("direct", _lib.F_FORCE_DIRECT)):
        def run():
            rc = lib.match_region(h, 0, step, N0, 0, step, N1, values.data_ptr(), 5, None, err.data_ptr(), None, 0.0, None, None, None,
                                  _lib.F_DEVICE_IO | flags, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            lib.check(rc, "match_region")
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print("step %d %s: %.2f ms, %.1f Mpx/s requested, path %d" % (step, tag, dt * 1e3, N0 * N1 / dt / 1e6, lib.last_path(h)), flush=True)
