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
        times = []
        for again in range(3):                       # the same model again: scratch buffers and pinned arrays exist
            ta = time.perf_counter()
            r = m.match(quiet=True)
            times.append((time.perf_counter() - ta) * 1e3)
        del m
    npx = r["f"].size
    print("debug=%s: create (H2D %.0f MB) %.1f ms, first match (+D2H, allocates scratch) %.1f ms, later matches %s ms, "
          "end-to-end (create + later match) %.2f Mpx/s" % (
              debug, 2 * sam.nbytes / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, " ".join("%.1f" % t for t in times),
              npx / ((t1 - t0) + min(times) * 1e-3) / 1e6))
