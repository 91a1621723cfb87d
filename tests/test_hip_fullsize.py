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


# ------------------------------------------------------------------------------------------------
# the other BASELINE configurations at (or near) their sizes, against the CPU oracle on row samples
# ------------------------------------------------------------------------------------------------
def _oracle_rows(port_ns, cls, sam, ref, Nw, ms, rows):
    m = getattr(port_ns, cls)(sam, ref, window_size=Nw, max_shift=ms)
    m.debug = True
    N1 = m.extent[1]
    return {r0: m.match(ROI=((r0, r1, 1), (0, N1, 1)), quiet=True) for r0, r1 in rows}


def test_C4_slab_full_width_against_the_oracle(port_ns):
    """One rank's slab of BASELINE config C4 (8192 columns, 1022 output rows + halo, 10 frames, Nw=5, max_shift=5):
    the whole slab on the GPU, three row bands of it on the CPU oracle."""
    from conftest import assert_parity
    from umpa_amd import model
    from umpa_amd.synth import make_block
    Nw, ms, K, W = 5, 5, 10, 8192
    P = Nw + ms
    sam, ref = make_block((3 * 1022, 4 * 1022 + 2 * P), 8192, W, K, ms, seed=300)      # rank 3's rows of the C4 image
    m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
    m.debug = True
    N0, N1 = m.extent
    assert (N0, N1) == (1022, 8172)
    bands = [(0, 3), (509, 512), (1019, 1022)]
    want = _oracle_rows(port_ns, "UMPAModelDF", sam, ref, Nw, ms, bands)
    for r0, r1 in bands:                                            # the GPU matches each band as an ROI of the slab model...
        got = m.match(ROI=((r0, r1, 1), (0, N1, 1)), quiet=True)
        assert_parity(got, want[r0], ms, "C4 slab rows %d-%d" % (r0, r1))
    m.debug = False                                                 # ...and the whole slab once: its bands must be those
    m.ROI = None                                                    # (match(ROI=...) leaves the ROI set, as in the reference)
    full = m.match(quiet=True)
    assert full["T"].shape == (1022, 8172)
    for r0, r1 in bands:
        np.testing.assert_array_equal(full["err"][r0:r1], want[r0]["err"])
        ok = want[r0]["err"] == 1
        np.testing.assert_allclose(full["T"][r0:r1][ok], want[r0]["T"][ok], rtol=1e-5)


def test_C3_parameters_at_1536_multi_chunk_table(port_ns, monkeypatch):
    """BASELINE config C3's parameters (20 frames, Nw=7, max_shift=8) on 1536 x 1536 with the default table budget
    (4 GiB: 225 planes of 1506 columns give 1568-row chunks, so a 1506-row region is one chunk) and with a 1 GiB budget
    (four chunks): same maps, and row bands of them against the CPU oracle."""
    from conftest import assert_parity
    from umpa_amd import model
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 7, 8, 20, 1536
    sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=7, order=1)
    m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
    m.debug = True
    N0, N1 = m.extent
    a = m.match(quiet=True)
    assert m._lib.last_path(m._handle) == 2
    monkeypatch.setenv("UMPA_HIP_TABLE_MB", "1024")
    b = m.match(quiet=True)
    for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    bands = [(0, 2), (700, 702), (N0 - 2, N0)]
    want = _oracle_rows(port_ns, "UMPAModelDF", sam, ref, Nw, ms, bands)
    for r0, r1 in bands:
        got = {k: (v[r0:r1] if isinstance(v, np.ndarray) and v.shape[:1] == (N0,) else v) for k, v in a.items()}
        assert_parity(got, want[r0], ms, "C3 params 1536 rows %d-%d" % (r0, r1))


def test_C3_full_size_default_and_chunked_table(port_ns, monkeypatch):
    """BASELINE config C3 at its full size (4096 x 4096, 20 frames, Nw=7, max_shift=8, dark-field on): the whole image with
    the default table budget (two row chunks) and with a 4 GiB budget (eight): identical maps; three 2-row bands of full
    width -- first rows, a chunk seam region, last rows -- against the CPU oracle, matched as ROIs (with the debug arrays the
    parity rules classify by) and read out of the whole-image maps."""
    from conftest import assert_parity
    from umpa_amd import model
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 7, 8, 20, 4096
    sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=11, order=1)
    m = model.UMPAModelDF(sam, ref, window_size=Nw, max_shift=ms)
    m.debug = "ncalls"
    N0, N1 = m.extent
    assert (N0, N1) == (4066, 4066)
    a = m.match(quiet=True)
    assert m._lib.last_path(m._handle) == 2
    monkeypatch.setenv("UMPA_HIP_TABLE_MB", "4096")
    b = m.match(quiet=True)
    monkeypatch.delenv("UMPA_HIP_TABLE_MB")
    for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    del b
    assert a["err"].mean() > 0.9
    bands = [(0, 2), (2303, 2305), (N0 - 2, N0)]                    # (2304 = the default budget's chunk seam at this width)
    want = _oracle_rows(port_ns, "UMPAModelDF", sam, ref, Nw, ms, bands)
    m.debug = True
    for r0, r1 in bands:
        got = m.match(ROI=((r0, r1, 1), (0, N1, 1)), quiet=True)
        assert_parity(got, want[r0], ms, "C3 full size rows %d-%d" % (r0, r1))
        for k in ("f", "T", "dx", "dy", "df", "err", "debug_Ncalls"):     # the whole-image match holds the same numbers
            np.testing.assert_array_equal(a[k][r0:r1], got[k], err_msg="%s rows %d-%d" % (k, r0, r1))


def test_C5_projection_full_size_through_the_streaming_matcher(port_ns):
    """One projection of BASELINE config C5 at its full size (2048 x 2048, 5 frames, Nw=5, max_shift=5, dark-field on) through
    StreamingMatcher -- uint16 counts, flat / dark correction fused into the upload, asynchronous match, maps in page-locked
    memory -- against the CPU oracle on (proj - dark) / flat, a 64-row band of full width; and twice in a row (the second
    projection reuses the reference-side maps)."""
    from conftest import assert_parity
    from umpa_amd.farm import StreamingMatcher
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 5, 5, 5, 2048
    sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=21, order=1)
    rng = np.random.default_rng(5)
    dark = 100.0 + rng.uniform(0, 2, size=(K, n, n))
    flat = 20000.0 * (1.0 + 0.05 * rng.standard_normal((K, n, n)))
    raws = [np.rint(np.roll(sam, p, axis=2) * flat + dark).astype(np.uint16) for p in range(2)]
    sm = StreamingMatcher(ref[None], Nw, ms, df=True, device=0, flats=flat[None], dark=dark, ref_nums=[0], debug=True)
    bufs = []
    for r in raws:
        bufs.append(sm.input_buffer(np.uint16))
        bufs[-1][...] = r
    N1 = n - 2 * (Nw + ms)
    r0, r1 = 980, 1044
    for pid, res in sm.run(enumerate(bufs)):
        corrected = (raws[pid].astype(np.float64) - dark) / flat
        o = port_ns.UMPAModelDF(corrected, ref, window_size=Nw, max_shift=ms)
        o.debug = True
        want = o.match(ROI=((r0, r1, 1), (0, N1, 1)), quiet=True)
        got = {k: (v[r0:r1] if isinstance(v, np.ndarray) else v) for k, v in res.items()}
        st = assert_parity(got, want, ms, "C5 full size p%d" % pid)
        assert st["ok"] > 0.7 * 64 * N1                                # (uint16 counts: about 80 % of the walks end inside the search range)


def test_streaming_matcher_survives_an_abandoned_series(port_ns):
    """A consumer that leaves StreamingMatcher.run() early (break) leaves matches in flight; the generator's clean-up waits
    for them, so that the next series on the same matcher pairs every wait with its own match: the second
    series' maps equal those of a fresh matcher bit for bit."""
    from umpa_amd.farm import StreamingMatcher
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 3, 4, 3, 256
    sam, ref, _ = make_stack(n, n, K, ms, df=True, seed=31, order=1)
    projs = [np.ascontiguousarray(np.roll(sam, p, axis=2)) for p in range(5)]
    sm = StreamingMatcher(ref[None], Nw, ms, df=True, device=0)
    for pid, res in sm.run(enumerate(projs)):
        break                                                       # two matches are in flight at this point
    second = {pid: {k: np.array(v) for k, v in res.items() if isinstance(v, np.ndarray)} for pid, res in sm.run(enumerate(projs))}
    fresh = StreamingMatcher(ref[None], Nw, ms, df=True, device=0)
    first = {pid: {k: np.array(v) for k, v in res.items() if isinstance(v, np.ndarray)} for pid, res in fresh.run(enumerate(projs))}
    assert sorted(second) == sorted(first) == list(range(5))
    for pid in first:
        for k in first[pid]:
            np.testing.assert_array_equal(second[pid][k], first[pid][k], err_msg="p%d %s" % (pid, k))
    sync = sm.model.match(quiet=True)                               # and the model takes a synchronous match again
    assert sync["err"].shape == first[0]["err"].shape
