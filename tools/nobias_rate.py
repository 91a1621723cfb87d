#!/usr/bin/env python3
"""the standard science call `align.UMPA_nobias` (sample-vs-reference match, reference-vs-reference
match, subtract, repair) on BASELINE config C2's stack, host arrays in, host maps out: one model for both matches
(reference stack and its maps stay resident, the sample stack is swapped in: umpa_amd/align.py) against the reference's
two models (align.py:98-117).  Per-kernel times of the one-model form show the reference-side maps being reused."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from umpa_amd import align, model
from umpa_amd.synth import make_stack
import ctypes

H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K, Nw, ms = 10, 5, 5
sam, ref, _ = make_stack(H, W, K, ms, df=True, seed=0, order=1)


def two_models():
    pm = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
    pb = model.UMPAModelDF(ref, ref, window_size=Nw, max_shift=ms)
    pm.debug = pb.debug = False
    r, b = pm.match(quiet=True), pb.match(quiet=True)
    r["dx"] = align.correct_bad_pixels(r["dx"] - b["dx"], ms)
    r["dy"] = align.correct_bad_pixels(r["dy"] - b["dy"], ms)
    return r


def one_model():
    return align.UMPA_nobias(sam, ref, window=Nw, shift=ms)


model.UMPAModelBase.debug = False                     # the helpers return the maps only; 332 B/px of debug arrays otherwise
out = {}
for name, fn in (("two models (reference's structure)", two_models), ("one model (umpa_amd.align.UMPA_nobias)", one_model)):
    fn()
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = fn()
        t.append(time.perf_counter() - t0)
    out[name] = dict(ms=round(min(t) * 1e3, 1), mpx_s=round(r["dx"].size / min(t) / 1e6, 1))
    print(name, out[name], flush=True)
a, b = two_models(), one_model()
assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in ("dx", "dy", "T", "df", "err")), "the two forms differ"

# per-kernel split of the one-model form: the second match recomputes only the sample side of prep_maps
pm = model.UMPAModelDF(ref, ref, window_size=Nw, max_shift=ms)
pm.debug = False
lib, h = pm._lib, pm._handle
split = {}
for tag, frames in (("bias match (ref vs ref)", None), ("sample match (ref maps reused)", sam)):
    if frames is not None:
        pm.update_frames(sam_list=frames)
    lib.timing_enable(h, 1)
    pm.match(quiet=True)
    lib.timing_enable(h, 0)
    k = {}
    for q in range(lib.timing_collect(h)):
        nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
        lib.timing_read(h, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
        k[nm.value.decode()] = round(tot.value, 3)
    split[tag] = k
    print(tag, k, flush=True)
out["kernels_ms"] = split
out["speedup"] = round(out["two models (reference's structure)"]["ms"] / out["one model (umpa_amd.align.UMPA_nobias)"]["ms"], 2)
print(json.dumps(out))
