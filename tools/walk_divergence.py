#!/usr/bin/env python3
"""How much of replay_walk's time is the spread of walk lengths inside a wave?

replay_walk gives every lane of a 64-lane wave one pixel (64 consecutive pixels of a row); the wave runs until its
LONGEST walk has ended.  This tool matches a BASELINE configuration on the GPU, takes the per-pixel number of cost
evaluations (the `Ncalls` debug array, Optim.cpp:267) and prints, for groups of 64 consecutive pixels of a row,
mean(max over the group) / mean(all pixels): the factor by which a wave outlasts its average lane.

    python tools/walk_divergence.py [C2|C3] ...
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from umpa_amd import _lib, model
    from umpa_amd.synth import CONFIGS, make_stack
    for config in (sys.argv[1:] or ["C2"]):
        cfg = CONFIGS[config]
        H, W, K, Nw, ms, df = cfg["H"], cfg["W"], cfg["K"], cfg["Nw"], cfg["max_shift"], cfg["df"]
        sam, ref, _ = make_stack(H, W, K, ms, df=df, seed=0)
        cls = model.UMPAModelDF if df else model.UMPAModelNoDF
        m = cls(sam, ref, window_size=Nw, max_shift=ms, device=0)
        lib, h = m._lib, m._handle
        N0, N1 = m.extent
        nparam = 5 if df else 4
        dev = torch.device("cuda", 0)
        values = torch.zeros((N0, N1, nparam), dtype=torch.float64, device=dev)
        err = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
        ncalls = torch.zeros((N0, N1), dtype=torch.int32, device=dev)
        rc = lib.match_region(h, 0, 1, N0, 0, 1, N1, values.data_ptr(), nparam, None, err.data_ptr(), None, 0.0, None, None,
                              ncalls.data_ptr(), _lib.F_DEVICE_IO, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        lib.check(rc, "match_region")
        torch.cuda.synchronize()
        nc = ncalls.cpu().numpy().astype(np.float64)
        n64 = (N1 // 64) * 64
        g = nc[:, :n64].reshape(N0, -1, 64)
        print("%s: %d x %d pixels, evaluations per pixel mean %.2f, p50 %d, p99 %d, max %d" % (
            config, N0, N1, nc.mean(), np.percentile(nc, 50), np.percentile(nc, 99), nc.max()))
        print("   per 64-pixel wave: mean of the wave maximum %.2f = %.2f x the mean lane (mean of wave minimum %.2f)" % (
            g.max(axis=2).mean(), g.max(axis=2).mean() / nc.mean(), g.min(axis=2).mean()))
        for q in (16, 32):
            gq = nc[:, :n64].reshape(N0, -1, q)
            print("   groups of %d: mean maximum %.2f = %.2f x" % (q, gq.max(axis=2).mean(), gq.max(axis=2).mean() / nc.mean()))
        del m


if __name__ == "__main__":
    main()
