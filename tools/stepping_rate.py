#!/usr/bin/env python3
"""A 2 x 2 sample-stepping stack (four positions, C2's parameters) on one GPU, device-resident: time per match and the
kernels it is made of (the rectangles with a constant set of contributing frames each run on the tiled path)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umpa_amd import _lib, model
from umpa_amd.synth import make_stack

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K, Nw, ms = 10, 5, 5
sam, ref, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)
m = model.UMPAModelDF([np.ascontiguousarray(sam[k, :H - 64, :W - 64]) for k in range(K)],
                      [np.ascontiguousarray(ref[k, :H - 64, :W - 64]) for k in range(K)], window_size=Nw, max_shift=ms,
                      pos_list=[np.array([64 * ((k // 2) % 2), 64 * (k % 2)]) for k in range(K)])
lib, h = m._lib, m._handle
N0, N1 = m.extent
dev = torch.device("cuda", 0)
values = torch.zeros((N0, N1, 5), dtype=torch.float64, device=dev)
err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
cm = m.coverage()
cover = torch.from_numpy(cm).to(dev)
thr = .1 * cm.max() / K
for tag, flags in (("auto", 0), ("general kernels", _lib.F_FORCE_DIRECT)):
    def step():
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), 5, None, err.data_ptr(), cover.data_ptr(), float(thr),
                              None, None, None, _lib.F_DEVICE_IO | flags, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.check(rc, "match_region")
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    lib.timing_enable(h, 1)
    step(); torch.cuda.synchronize()
    parts = []
    for q in range(lib.timing_collect(h)):
        nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
        lib.timing_read(h, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
        parts.append("%s %.2f ms / %d" % (nm.value.decode(), tot.value, cnt.value))
    lib.timing_enable(h, 0)
    print("%s: %.2f ms, %.1f Mpx/s, path %d, ok %.4f | %s" % (tag, dt * 1e3, N0 * N1 / dt / 1e6, lib.last_path(h),
                                                              float(err.float().mean()), "; ".join(parts)), flush=True)
