"""
Host-side logic of umpa_amd.model (the part of the reference that lived in Cython: ROI / extent / step
arithmetic, window, validation, result packing), exercised on CPU through the oracle backend.
The numbers it is compared with come from the reference goldens (A_small) -- see test_oracle_golden.py
for the full sweep; here the API behaviour is pinned.
"""
import numpy as np
import pytest

from conftest import Case


@pytest.fixture(scope="module")
def case():
    return Case("A_small")


def test_extent_padding_window(port_ns, case):
    m = port_ns.UMPAModelDF(case.sam, case.ref, window_size=2, max_shift=4)
    assert m.padding == 6 and m.extent == (52, 60) and m.sh == (52, 60) and m.Na == 3
    assert m.ROI == ((0, 52, 1), (0, 60, 1))
    w = np.multiply.outer(np.hamming(5), np.hamming(5))
    np.testing.assert_allclose(m.window, w / w.sum(), rtol=0, atol=1e-17)
    assert m.assign_coordinates == "sam" and m.sub_pixel_mode == -1 and m.Nw == 2 and m.max_shift == 4
    m.assign_coordinates = "nonsense"                  # prints, does not change (model.pyx:732-742)
    assert m.assign_coordinates == "sam"
    m.assign_coordinates = "ref"
    assert m.assign_coordinates == "ref"
    c0, c1 = m.coords()
    assert c0[0] == 6 and c1[-1] == 6 + 59
    k = port_ns.UMPAModelDFKernel(case.sam, case.ref, window_size=2, max_shift=4)
    assert k.padding == 14                             # safe_crop 8 (model.pyx:904)


def test_roi_step_arithmetic(port_ns, case):
    m = port_ns.UMPAModelNoDF(case.sam, case.ref, window_size=2, max_shift=4)
    m.debug = False
    full = m.match(quiet=True)
    assert set(full) == {"err", "f", "T", "dx", "dy"}
    m.ROI = None
    r = m.match(step=3, quiet=True)
    assert r["f"].shape == (18, 20) and m.ROI == ((0, 52, 3), (0, 60, 3))
    np.testing.assert_array_equal(r["dx"], full["dx"][::3, ::3])
    m.ROI = None
    r = m.match(ROI=(slice(4, 30), slice(10, 50, 2)), quiet=True)
    np.testing.assert_array_equal(r["T"], full["T"][4:30, 10:50:2])
    r = m.match(ROI=((5, 40, 2), (3, 50, 3)), quiet=True)
    np.testing.assert_array_equal(r["T"], full["T"][5:40:2, 3:50:3])
    assert m.set_step(4) == ((5, 40, 4), (3, 50, 4))
    with pytest.raises(RuntimeError, match="exceeds the reconstructible extent"):
        m.match(ROI=((0, 60, 1), (0, 60, 1)), quiet=True)      # the reference would read out of bounds
    with pytest.raises(RuntimeError, match="should not be specified simultaneously"):
        m._convert_ROI_slice(ROI=((0, 5, 1), (0, 5, 1)), step=2)


def test_start_shift_and_subpixel_modes(port_ns, case):
    m = port_ns.UMPAModelDF(case.sam, case.ref, window_size=2, max_shift=4)
    r = m.match(dxdy=(5, 0), quiet=True)                       # first call already out of bounds
    assert not r["err"].any() and (r["debug_Ncalls"] == 1).all()
    assert (r["dy"] == 5).all() and (r["dx"] == 0).all()       # dxdy[0] is the ROW shift and comes back as dy
    m.sub_pixel_mode = 0
    r0 = m.match(quiet=True)
    ok = r0["err"] == 1
    assert np.all(r0["dx"][ok] == np.round(r0["dx"][ok])) and set(np.unique(r0["f"][ok])) <= {0.0, 1.0}


def test_validation_errors(port_ns, case):
    with pytest.raises(RuntimeError, match="not C-contiguous"):
        port_ns.UMPAModelDF(case.sam[:, ::2, ::2], case.ref[:, ::2, ::2])
    with pytest.raises(RuntimeError, match="Incompatible shape"):
        port_ns.UMPAModelDF(case.sam, case.ref[:, :, :-2].copy())
    with pytest.raises(RuntimeError, match="Unexpected length for position list"):
        port_ns.UMPAModelDF(case.sam, case.ref, pos_list=[np.array([0, 0])])
    with pytest.raises(RuntimeError, match="Negative frame positions"):
        port_ns.UMPAModelDF(case.sam, case.ref, pos_list=[np.array([0, 0]), np.array([-1, 0]), np.array([0, 0])])
    m = port_ns.UMPAModelDF(case.sam, case.ref, window_size=2, max_shift=4)
    with pytest.raises(RuntimeError, match="non-negative"):
        m.Nw = -1
    with pytest.raises(RuntimeError, match="does not fit the padding"):
        m.Nw = 3
    with pytest.raises(RuntimeError, match="wrong shape"):
        m._match(input_values=np.zeros((3, 3, 5)), quiet=True)
    k = port_ns.UMPAModelDFKernel(case.sam, case.ref, window_size=2, max_shift=4)
    with pytest.raises(RuntimeError, match="abc array has to be provided"):
        k.match(quiet=True)


def test_match_unbiased_and_list_inputs(port_ns, case, monkeypatch):
    """speckle_matching.match / match_unbiased semantics with the model classes swapped for the CPU checker."""
    from umpa_amd import model, speckle_matching
    monkeypatch.setattr(model, "UMPAModelDF", port_ns.UMPAModelDF)
    monkeypatch.setattr(model, "UMPAModelNoDF", port_ns.UMPAModelNoDF)
    frames = [np.ascontiguousarray(f) for f in case.sam]
    refs = [np.ascontiguousarray(f) for f in case.ref]
    r = speckle_matching.match(frames, refs, 2, step=2, max_shift=1)          # max_shift ignored: built with 4
    assert r["f"].shape == (26, 30) and "df" in r
    ru = speckle_matching.match_unbiased(frames, refs, 2, step=2)
    rr = speckle_matching.match(refs, refs, 2, step=2)
    np.testing.assert_allclose(ru["dx"], r["dx"] - rr["dx"])
    np.testing.assert_allclose(ru["dy"], r["dy"] - rr["dy"])
    rn = speckle_matching.match_unbiased(frames, refs, 2, step=2, df=False, bias=False)
    assert "df" not in rn


def test_degenerate_sizes(port_ns):
    """Frames no larger than twice the padding leave nothing to reconstruct: a clean error, not a crash."""
    a = np.ones((2, 12, 30))
    m = port_ns.UMPAModelDF(a, a, window_size=2, max_shift=4)           # padding 6: extent (0, 18)
    assert m.extent == (0, 18)
    with pytest.raises(RuntimeError, match="Empty ROI"):
        m.match(quiet=True)
    b = np.ones((1, 13, 13))
    m = port_ns.UMPAModelNoDF(b, b, window_size=2, max_shift=4)         # exactly one output pixel
    assert m.extent == (1, 1)
    r = m.match(quiet=True)
    assert r["f"].shape == (1, 1) and r["err"].dtype == np.int32


def test_checker_geometry_is_its_own_and_agrees_with_the_candidates(port_ns):
    """oracle/cpu_model.py computes extent, ROI / step / slice conversion and pixel counts itself (the
    checker must not share that layer with the candidate).  The two implementations are compared on random forms -- tuples,
    slices with negative / open bounds, steps, set_step -- and on sample-stepping extents."""
    from umpa_amd import model
    rng = np.random.default_rng(12)
    cand_cls, chk_cls = model.UMPAModelDF, port_ns.UMPAModelDF
    assert chk_cls._calculate_extent is not cand_cls._calculate_extent and chk_cls._counts is not cand_cls._counts
    assert chk_cls._convert_ROI_slice is not cand_cls._convert_ROI_slice and chk_cls._set_ROI is not cand_cls._set_ROI

    def host(cls, pos, shapes, padding):                            # the state the geometry methods read, without a native model
        h = object.__new__(type("Host", (cls,), {"__del__": lambda self: None}))
        h._pos_list, h._shape_list, h._padding = pos, shapes, padding
        h._set_ROI(None)
        return h

    for trial in range(200):
        K = int(rng.integers(1, 5))
        pos = rng.integers(0, 9, size=(K, 2)); pos -= pos.min(axis=0)
        shapes = [(int(rng.integers(40, 70)), int(rng.integers(40, 70))) for _ in range(K)]
        pad = int(rng.integers(2, 9))
        a, b = host(cand_cls, [tuple(p) for p in pos], shapes, pad), host(chk_cls, [tuple(p) for p in pos], shapes, pad)
        assert a._calculate_extent() == b._calculate_extent()
        n0, n1 = a._calculate_extent()

        def rnd(n):
            if rng.random() < 0.5:
                lo = int(rng.integers(0, n - 2)); hi = int(rng.integers(lo + 1, n + 1))
                return (lo, hi, int(rng.integers(1, 5)))
            pick = lambda: None if rng.random() < 0.3 else int(rng.integers(-n, n + 3))
            return slice(pick(), pick(), int(rng.integers(1, 5)))
        roi = (rnd(n0), rnd(n1))
        ra, rb = a._convert_ROI_slice(ROI=roi), b._convert_ROI_slice(ROI=roi)
        assert [len(range(*t)) for t in ra] == [len(range(*t)) for t in rb], (roi, ra, rb)
        assert [list(range(*t))[:3] for t in ra] == [list(range(*t))[:3] for t in rb], (roi, ra, rb)
        if all(len(range(*t)) for t in ra):
            assert cand_cls._counts(*ra) == chk_cls._counts(*rb) == tuple(len(range(*t)) for t in ra)
        step = int(rng.integers(1, 6))
        sa, sb = a._convert_ROI_slice(step=step), b._convert_ROI_slice(step=step)
        assert [list(range(*t)) for t in sa] == [list(range(*t)) for t in sb]
        a._set_ROI(roi); b._set_ROI(roi)
        assert [list(range(*t)) for t in a._ROI] == [list(range(*t)) for t in b._ROI]
