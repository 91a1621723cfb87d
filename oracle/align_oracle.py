"""
TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md section 2): numpy restatement of the reference's
bad-pixel repair, `UMPA/align.py:661-732`.  Imported by tests/ only; the product path
(`umpa_amd/align.py` -> `umpa_hip_correct_bad_pixels`) never touches it.

Pinned by tests/golden/G_align.npz (outputs of the imported, unmodified reference;
generator tests/golden/make_golden_align.py).
"""
import numpy as np


def correct_bad_pixels(img_in, th=None, iterations=1, dims=(-2, -1), p=0.5):
    img = np.array(img_in, copy=True)                                  # align.py:700
    if th is None:                                                     # align.py:703-706
        bounds = (np.percentile(img, p), np.percentile(img, 100 - p))
    else:
        bounds = (-th, th)
    lo, hi = min(bounds), max(bounds)
    where = np.nonzero((img < lo) | (img > hi))                        # align.py:708-709: fixed for all passes
    if where[0].size == 0:                                             # align.py:711-712
        return img
    for _ in range(iterations):
        neighbours = []
        for d in dims:                                                 # two neighbours per axis, reflected at the ends
            n = img.shape[d]
            for step in (-1, 1):
                j = where[d] + step
                j = np.where(j < 0, 1, j)                              # align.py:723: |i - 1|
                j = np.where(j == n, n - 2, j)                         # align.py:728
                at = list(where)
                at[d] = j
                neighbours.append(img[tuple(at)])                      # gathered before anything is written
        img[where] = np.median(np.stack(neighbours), axis=0)           # align.py:732
    return img
