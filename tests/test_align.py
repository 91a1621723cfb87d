"""
The callers next to the matching path: `correct_bad_pixels`,
`UMPA_normal`, `UMPA_nobias` of the reference's UMPA/align.py.

CPU: the numpy restatement (oracle/align_oracle.py) against the golden outputs of the imported reference
(tests/golden/G_align.npz, generator make_golden_align.py).  GPU: `umpa_amd.align` -- which runs
`umpa_hip_correct_bad_pixels` and the HIP models through the C ABI -- against the same goldens
(bit-exact for the repair, 1e-5 for the maps) and against the oracle on larger seeded inputs.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

G = np.load(os.path.join(GOLDEN, "G_align.npz"))

BP_CASES = [
    ("bp_a2_th3_it1", "bp_a2", dict(th=3.0)),
    ("bp_a2_th3_it3", "bp_a2", dict(th=3.0, iterations=3)),
    ("bp_a2_auto", "bp_a2", dict()),
    ("bp_a2_auto_p2", "bp_a2", dict(p=2.0)),
    ("bp_a2_cols", "bp_a2", dict(th=3.0, dims=(-1,))),
    ("bp_a2_rows", "bp_a2", dict(th=3.0, dims=(0,))),
    ("bp_a2_none", "bp_a2", dict(th=100.0)),
    ("bp_a3_th2", "bp_a3", dict(th=2.0, iterations=2)),
]


@pytest.mark.parametrize("want,src,kw", BP_CASES, ids=[c[0] for c in BP_CASES])
def test_oracle_bad_pixels_match_reference_golden(want, src, kw):
    from oracle.align_oracle import correct_bad_pixels
    got = correct_bad_pixels(G[src], **kw)
    assert np.array_equal(got, G[want])
    assert got is not G[src]


def test_golden_actually_repairs_something():
    assert (G["bp_a2_th3_it1"] != G["bp_a2"]).sum() == 70
    assert np.abs(G["bp_a2_th3_it3"]).max() < np.abs(G["bp_a2"]).max()
    assert np.array_equal(G["bp_a2_none"], G["bp_a2"])


# ------------------------------------------------------------------------------------------------ GPU

@pytest.fixture(scope="module")
def hip_align():
    from umpa_amd import _lib, align
    if _lib.hip().device_count() < 1:
        pytest.fail("no HIP device: the GPU tests cannot run (there is no CPU fallback)")
    return align


@pytest.mark.gpu
@pytest.mark.parametrize("want,src,kw", BP_CASES, ids=[c[0] for c in BP_CASES])
def test_hip_bad_pixels_match_reference_golden(hip_align, want, src, kw):
    src_before = G[src].copy()
    got = hip_align.correct_bad_pixels(G[src], **kw)
    assert got.dtype == np.float64 and got.shape == G[want].shape and got.flags.c_contiguous
    assert np.array_equal(got, G[want])                                # bit-exact
    assert np.array_equal(G[src], src_before)                          # the input is not touched


@pytest.mark.gpu
def test_hip_bad_pixels_against_oracle_large_and_odd(hip_align):
    from oracle.align_oracle import correct_bad_pixels as ref
    rng = np.random.default_rng(99)
    for shape, kw in [((513, 1027), dict(th=2.5, iterations=2)), ((2, 3, 65, 33), dict(th=2.0)),
                      ((7, 2), dict(th=1.0, iterations=4)), ((64, 64), dict(th=2.0, dims=(0, 1))),
                      ((5, 40, 30), dict(th=2.0, dims=(0, 2))), ((1000,), dict(th=2.0, dims=(0,)))]:
        a = rng.standard_normal(shape)
        a[rng.random(shape) < 0.01] = np.nan                           # numpy.median propagates NaN
        assert np.array_equal(hip_align.correct_bad_pixels(a, **kw), ref(a, **kw), equal_nan=True), (shape, kw)
    a32 = rng.standard_normal((40, 50)).astype(np.float32) * 2
    got = hip_align.correct_bad_pixels(a32, 2.0)
    assert got.dtype == np.float32 and np.array_equal(got, ref(a32, 2.0))


@pytest.mark.gpu
def test_hip_bad_pixels_argument_errors(hip_align):
    with pytest.raises(NotImplementedError):
        hip_align.correct_bad_pixels(np.zeros((4, 4, 4)), 1.0, dims=(0, 1, 2))
    with pytest.raises(IndexError):
        hip_align.correct_bad_pixels(np.ones((4, 1)) * 5, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name,fn,kw", [
    ("normal", "UMPA_normal", {}), ("nobias", "UMPA_nobias", {}),
    ("nobias_ref", "UMPA_nobias", dict(assign_coordinates="ref")),
    ("normal_roi", "UMPA_normal", dict(ROI=(slice(2, 30, 1), slice(None, None, 1)))),
])
def test_hip_help_functions_match_reference_golden(hip_align, name, fn, kw):
    res = getattr(hip_align, fn)(G["hf_sam"], G["hf_ref"], window=2, shift=3, num_threads=1, **kw)
    want = {k: G["hf_%s_%s" % (name, k)] for k in ("dx", "dy", "T", "df", "err")}
    assert np.array_equal(res["err"], want["err"])
    ok = want["err"] == 1
    for k in ("T", "df"):
        assert np.allclose(res[k][ok], want[k][ok], rtol=1e-5, atol=0), k
    # dx / dy went through the repair: a pixel within 1e-5 of the threshold may fall on the other side
    for k in ("dx", "dy"):
        close = np.abs(res[k] - want[k]) <= 1e-5 * np.maximum(1.0, np.abs(want[k]))
        assert close.mean() > 0.98, (k, close.mean())
        assert np.abs(res[k]).max() <= max(3.0, np.abs(want[k]).max()) + 1e-9
