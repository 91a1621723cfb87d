#!/usr/bin/env python3
"""End-to-end rate of the host-array API (UMPAModelDF(...).match()) on one GPU: includes H2D of the
frame stacks, the kernels and D2H of the result maps.  Quoted in DESIGN.md; never bench.py's `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpa_amd import model
from umpa_amd.synth import CONFIGS, make_stack

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
sam, ref, _ = make_stack(cfg["H"], cfg["W"], cfg["K"], cfg["max_shift"], df=cfg["df"], seed=0, order=1)
cls = model.UMPAModelDF if cfg["df"] else model.UMPAModelNoDF
for debug in (False, True):
    for rep in range(3):
        t0 = time.perf_counter()
        m = cls(sam, ref, window_size=cfg["Nw"], max_shift=cfg["max_shift"])
        t1 = time.perf_counter()
        m.debug = debug
        r = m.match(quiet=True)
        t2 = time.perf_counter()
        del m
    npx = r["f"].size
    print("debug=%s: create (H2D %.0f MB) %.1f ms, match (+D2H) %.1f ms, end-to-end %.2f Mpx/s" % (
        debug, 2 * sam.nbytes / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, npx / (t2 - t0) / 1e6))
