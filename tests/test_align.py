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
    """The full bar on the repaired dx / dy.  A pixel of the final map may differ from the reference's only if that is
    EXPLAINED: its raw value -- or, because a repaired pixel is the median of its neighbours, the raw value of one of its
    four neighbours -- is (a) a pixel the parity checker classifies as ill-posed in the raw match (unconverged Newton,
    counted and capped there), or (b) within 1e-5 of the repair threshold, so that it may fall on either side of it.
    Everything else must agree to 1e-5, and the explained set must stay small."""
    from conftest import assert_parity
    from oracle import cpu_model
    from umpa_amd import model
    shift = 3
    res = getattr(hip_align, fn)(G["hf_sam"], G["hf_ref"], window=2, shift=shift, num_threads=1, **kw)
    want = {k: G["hf_%s_%s" % (name, k)] for k in ("dx", "dy", "T", "df", "err")}
    assert np.array_equal(res["err"], want["err"])
    ok = want["err"] == 1
    for k in ("T", "df"):
        assert np.allclose(res[k][ok], want[k][ok], rtol=1e-5, atol=0), k

    # the raw (un-repaired) maps of both sides: the HIP model and the CPU oracle, each as the helper builds them
    def raw(ns):
        out = []
        pairs = [(G["hf_sam"], "sam")] if fn == "UMPA_normal" else [(G["hf_ref"], "bias"), (G["hf_sam"], "sam")]
        for stack, role in pairs:
            m = ns.UMPAModelDF(stack, G["hf_ref"], window_size=2, max_shift=shift)
            if role == "sam":
                m.assign_coordinates = kw.get("assign_coordinates", "sam")
            out.append(m.match(ROI=kw.get("ROI", (slice(None), slice(None))), quiet=True))
        return out
    raw_g, raw_o = raw(model), raw(cpu_model.port)
    suspect = np.zeros(want["dx"].shape, dtype=bool)
    for g, o in zip(raw_g, raw_o):
        assert_parity(g, o, shift, "align %s raw" % name)           # raises unless every differing pixel is ill-posed (and few)
        for k in ("dx", "dy"):
            suspect |= ~(np.abs(g[k] - o[k]) <= 1e-5 * np.maximum(1.0, np.abs(o[k])))
    for k in ("dx", "dy"):
        r = raw_o[-1][k] - (raw_o[0][k] if len(raw_o) == 2 else 0.0)
        suspect |= np.abs(np.abs(r) - shift) <= 1e-5 * shift          # may fall on either side of the threshold
    # one repair pass: a pixel's value depends on its four neighbours (edges reflect)
    near = suspect.copy()
    pad = np.pad(suspect, 1, mode="reflect")
    near |= pad[:-2, 1:-1] | pad[2:, 1:-1] | pad[1:-1, :-2] | pad[1:-1, 2:]
    assert near.mean() < 0.03, near.mean()
    for k in ("dx", "dy"):
        close = np.abs(res[k] - want[k]) <= 1e-5 * np.maximum(1.0, np.abs(want[k]))
        assert np.all(close | near), (k, int((~close & ~near).sum()))
        assert np.abs(res[k]).max() <= max(3.0, np.abs(want[k]).max()) + 1e-9


@pytest.mark.gpu
def test_nobias_takes_device_tensors_and_reports_shape_errors(hip_align):
    """the one-model shortcut of UMPA_nobias needs frames the library owns; frames that live on the GPU
    already (torch tensors, borrowed) must still work, and a shape mismatch must raise the reference's message."""
    import torch
    sam, ref = G["hf_sam"], G["hf_ref"]
    want = hip_align.UMPA_nobias(sam, ref, window=2, shift=3)
    dev = torch.device("cuda", 0)
    ts = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in sam]
    tr = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in ref]
    got = hip_align.UMPA_nobias(ts, tr, window=2, shift=3)
    for k in ("dx", "dy", "T", "df", "err"):
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    with pytest.raises(RuntimeError, match="Incompatible shape"):
        hip_align.UMPA_nobias(sam[:, :-2, :].copy(), ref, window=2, shift=3)
