#!/usr/bin/env python3
"""Frames beyond 2 GiB: the tiled path's LDS-DMA addresses a frame with a 32-bit byte offset (unsigned, up to 4 GiB).
A ROI at the bottom of a 16400 x 16400 frame (2.15 GB: offsets above 2^31) on the tiled path against the general kernel."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpa_amd import model, _lib

N = 16400
rng = np.random.default_rng(3)
base = rng.random((N + 8, N + 8))
ref = np.ascontiguousarray(base[4:4 + N, 4:4 + N])
sam = np.ascontiguousarray(base[5:5 + N, 3:3 + N]) * 0.9            # a shift of (1, -1)
res = {}
for tag, force in (("tiled", _lib.F_FORCE_TILED), ("direct", _lib.F_FORCE_DIRECT)):
    m = model.UMPAModelNoDF([sam], [ref], window_size=2, max_shift=3)
    m._force = force
    E0, E1 = m.extent
    res[tag] = m.match(ROI=((E0 - 96, E0, 1), (E1 - 4000, E1, 1)), quiet=True)
    print(tag, "path", m._lib.last_path(m._handle), "ok share", float(res[tag]["err"].mean()), flush=True)
    del m
a, b = res["tiled"], res["direct"]
assert np.array_equal(a["err"], b["err"])
ok = b["err"] == 1
assert ok.mean() > 0.05                                              # (white noise: most walks leave the search box; both paths agree on which)
for k in ("dx", "dy", "T", "f"):
    d = np.abs(a[k] - b[k])[ok] / np.maximum(1.0, np.abs(b[k][ok]))
    print(k, "max diff", d.max())
    assert np.mean(d > 1e-6) < 2e-3
print("dx median", np.median(b["dx"][ok]), "dy median", np.median(b["dy"][ok]))
print("big frame check OK")
