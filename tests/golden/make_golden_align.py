#!/usr/bin/env python3
"""
Golden fixtures for the callers next to the matching path:
`UMPA.align.correct_bad_pixels`, `UMPA.align.UMPA_normal`, `UMPA.align.UMPA_nobias`
(reference UMPA/align.py:12-117, 661-732).  Run in the BUILD container only, after
tests/golden/make_golden.py has made the scratch build of the reference under /tmp/umpa_oracle.

Only data is written: input arrays and the reference's output arrays (tests/golden/G_align.npz).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)


def main():
    import make_golden as MG
    MG.build_reference(False)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, MG.SCRATCH)
    from UMPA import align as RA                                     # the unmodified reference
    from umpa_amd.synth import make_stack

    out = {}
    rng = np.random.default_rng(4242)

    def hot(shape, nbad, amp):
        yy, xx = np.meshgrid(np.linspace(0, 3, shape[-2]), np.linspace(0, 2, shape[-1]), indexing="ij")
        a = np.broadcast_to(np.sin(yy) * np.cos(xx), shape) + 0.05 * rng.standard_normal(shape)
        a = np.ascontiguousarray(a)
        flat = a.reshape(-1)
        idx = rng.choice(flat.size, nbad, replace=False)
        flat[idx] = amp * rng.choice([-1.0, 1.0], nbad) * (1.0 + rng.random(nbad))
        return a

    a2 = hot((37, 41), 60, 5.0)
    # corners, edges and a clump of adjacent bad pixels
    a2[0, 0] = 9.0; a2[0, 40] = -7.0; a2[36, 0] = 8.0; a2[36, 40] = 6.5; a2[0, 17] = 5.5; a2[20, 0] = -6.0
    a2[10:12, 10:13] = 7.0
    out["bp_a2"] = a2
    out["bp_a2_th3_it1"] = RA.correct_bad_pixels(a2, 3.0)
    out["bp_a2_th3_it3"] = RA.correct_bad_pixels(a2, 3.0, iterations=3)
    out["bp_a2_auto"] = RA.correct_bad_pixels(a2)                     # percentile thresholds, p = 0.5
    out["bp_a2_auto_p2"] = RA.correct_bad_pixels(a2, p=2.0)
    out["bp_a2_cols"] = RA.correct_bad_pixels(a2, 3.0, dims=(-1,))
    out["bp_a2_rows"] = RA.correct_bad_pixels(a2, 3.0, dims=(0,))
    out["bp_a2_none"] = RA.correct_bad_pixels(a2, 100.0)
    a3 = hot((3, 16, 18), 40, 4.0)
    out["bp_a3"] = a3
    out["bp_a3_th2"] = RA.correct_bad_pixels(a3, 2.0, iterations=2)

    # the two help functions on a small stack
    sam, ref, _ = make_stack(48, 56, 4, 3, df=True, seed=77)
    out["hf_sam"], out["hf_ref"] = sam, ref
    roi = (slice(2, 30, 1), slice(None, None, 1))
    for name, fn, kw in [("normal", RA.UMPA_normal, {}), ("nobias", RA.UMPA_nobias, {}),
                         ("nobias_ref", RA.UMPA_nobias, dict(assign_coordinates="ref")),
                         ("normal_roi", RA.UMPA_normal, dict(ROI=roi))]:
        r = fn(sam, ref, window=2, shift=3, num_threads=1, **kw)
        for k in ("dx", "dy", "T", "df", "err"):
            out["hf_%s_%s" % (name, k)] = r[k]
    np.savez_compressed(os.path.join(HERE, "G_align.npz"), **out)
    print("wrote G_align.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
