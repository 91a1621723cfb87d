"""
Parity at BASELINE.json's full C2 size (2048 x 2048, 10 frames, Nw=5, max_shift=5, dark-field) through
size-independent properties -- the CPU oracle would need minutes per case here:

  * two independent GPU implementations (tiled fast path vs the general direct kernel, which follows
    the reference's own term order) must produce the same walk for all 4.1 Mpx and the same maps;
  * an exactly translated, scaled stack must come back as that integer shift, T = scale, df = 1;
  * ROI / step results are slices of the full result (pixels are independent).
bench.py additionally compares the GPU maps with the reference C++ core on every run (`gpu_agrees`).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H = W = 2048
K, NW, MS = 10, 5, 5


@pytest.fixture(scope="module")
def stack():
    from umpa_amd.synth import make_stack
    return make_stack(H, W, K, MS, df=True, seed=0, order=1)[:2]


def _model(sam, ref, force, debug="ncalls", cls="UMPAModelDF"):
    from umpa_amd import _lib, model
    m = getattr(model, cls)(sam, ref, window_size=NW, max_shift=MS)
    m.debug = debug
    m._force = {"tiled": _lib.F_FORCE_TILED, "direct": _lib.F_FORCE_DIRECT}[force]
    return m


def test_tiled_and_direct_agree_on_all_of_C2(stack):
    sam, ref = stack
    a = _model(sam, ref, "tiled").match(quiet=True)
    b = _model(sam, ref, "direct").match(quiet=True)
    assert a["f"].shape == (H - 2 * (NW + MS),) * 2
    np.testing.assert_array_equal(a["err"], b["err"])
    np.testing.assert_array_equal(a["debug_Ncalls"], b["debug_Ncalls"])
    ok = b["err"] == 1
    assert ok.mean() > 0.99
    for k in ("T", "df"):
        assert np.max(np.abs(a[k] - b[k])[ok] / np.abs(b[k][ok])) < 1e-9, k
    bad = ~ok
    for k in ("dx", "dy"):                                   # failed pixels report integers (or the start value): exact
        np.testing.assert_array_equal(a[k][bad], b[k][bad])
    # sub-pixel maps: 1e-5 bar; the only admissible misses are unconverged-Newton pixels (conftest.assert_parity),
    # of which a smooth field has a handful
    miss = np.zeros_like(ok)
    for k in ("dx", "dy"):
        miss |= ok & ~(np.abs(a[k] - b[k]) <= 1e-5 * np.maximum(1.0, np.abs(b[k])))
    miss |= ok & ~(np.abs(a["f"] - b["f"]) <= 1e-5 * np.abs(b["f"]))
    assert miss.sum() <= 1e-4 * ok.sum(), "%d of %d pixels differ" % (miss.sum(), ok.sum())


@pytest.mark.parametrize("force", ["tiled", "direct"])
@pytest.mark.parametrize("cls", ["UMPAModelDF", "UMPAModelNoDF"])
def test_exact_translation_is_recovered(stack, force, cls):
    _, ref = stack
    dy, dx, T0 = 1, -2, 0.75
    sam = np.ascontiguousarray(T0 * np.roll(ref, (-dy, -dx), axis=(1, 2)))      # sam[i,j] = T0 * ref[i+dy, j+dx]
    m = _model(sam, ref, force, cls=cls)
    m.sub_pixel_mode = 0
    m.ROI = ((0, 256, 1), (0, 2028, 1)) if force == "direct" else None           # the direct kernel is ~30x slower
    r = m.match(quiet=True)
    inner = (slice(16, -16), slice(16, -16))                                     # np.roll wraps around at the border
    assert r["err"][inner].all()
    assert (r["dy"][inner] == dy).all() and (r["dx"][inner] == dx).all()
    np.testing.assert_allclose(r["T"][inner], T0, rtol=1e-10)
    if cls == "UMPAModelDF":
        np.testing.assert_allclose(r["df"][inner], 1.0, rtol=1e-8)


def test_roi_and_step_are_slices_of_the_full_result(stack):
    sam, ref = stack
    m = _model(sam, ref, "tiled", debug=False)
    full = m.match(quiet=True)
    m.ROI = None
    roi = m.match(ROI=((100, 900, 1), (37, 2001, 1)), quiet=True)                # unit step: still the tiled path
    for k in ("f", "T", "dx", "dy", "df", "err"):
        np.testing.assert_array_equal(roi[k], full[k][100:900, 37:2001])
    m2 = _model(sam, ref, "direct", debug=False)
    part = m2.match(ROI=((3, 2028, 97), (5, 2028, 53)), quiet=True)              # stepped: general kernel
    np.testing.assert_array_equal(part["err"], full["err"][3::97, 5::53])
    ok = part["err"] == 1
    np.testing.assert_allclose(part["T"][ok], full["T"][3::97, 5::53][ok], rtol=1e-9)
