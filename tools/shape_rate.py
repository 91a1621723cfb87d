#!/usr/bin/env python3
"""corr_volume per workgroup shape (UMPA_HIP_CORR_SHAPE) for C2's stack at other windows / search ranges:
tools/shape_rate.py [Nw:ms ...]  (device-resident, HIP-event time of the table kernel per match)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from umpa_amd import _lib, model
from umpa_amd.synth import make_stack

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(5, 3), (5, 4), (5, 6), (5, 7), (3, 4), (7, 6)]
H = W = 2048
K = 10
dev = torch.device("cuda", 0)
for Nw, ms in cases:
    sam, ref, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)
    m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
    lib, h = m._lib, m._handle
    N0, N1 = m.extent
    values = torch.zeros((N0, N1, 5), dtype=torch.float64, device=dev)
    err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
    out = []
    for shape in (0, 1, 5, 6):
        os.environ["UMPA_HIP_CORR_SHAPE"] = str(shape)
        def step():
            rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), 5, None, err.data_ptr(), None, 0.0, None, None, None,
                                  _lib.F_DEVICE_IO | _lib.F_FORCE_TILED, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            return rc
        if step() < 0:
            out.append("shape %d: n/a" % shape); continue
        torch.cuda.synchronize()
        lib.timing_enable(h, 1)
        for _ in range(5): step()
        torch.cuda.synchronize()
        t = {}
        for q in range(lib.timing_collect(h)):
            nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
            lib.timing_read(h, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
            t[nm.value.decode()] = tot.value / 5
        lib.timing_enable(h, 0)
        out.append("shape %d: corr %.3f (all %.3f)" % (shape, t.get("corr_volume", 0), sum(t.values())))
    print("Nw %d ms %d: %s" % (Nw, ms, "; ".join(out)), flush=True)
    del m
