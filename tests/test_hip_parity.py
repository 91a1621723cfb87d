"""
GPU parity tests: the HIP path, called through the C ABI (umpa_amd.model -> libumpa_hip.so),
against (a) the golden vectors frozen from the reference and (b) the CPU oracle on the same
seeded inputs.  Bar: err / Ncalls / integer minimum bit-exact, float maps <= 1e-5 relative
(conftest.assert_parity).  Run with `pytest -m gpu` on the MI355X box.
"""
import os

import ctypes

import numpy as np
import pytest

from conftest import ALL_CASES, GOLDEN, Case, assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip_ns():
    from umpa_amd import _lib, model
    if _lib.hip().device_count() < 1:
        pytest.fail("no HIP device: the GPU tests cannot run (there is no CPU fallback)")
    return model


def _variants(names):
    out = []
    for name in names:
        for n in range(len(Case(name).variants)):
            out.append((name, n))
    return out


GPU_CASES = list(ALL_CASES)


@pytest.mark.parametrize("name,n", _variants(GPU_CASES))
def test_hip_matches_reference_golden(hip_ns, name, n):
    case = Case(name)
    got, m = case.run(hip_ns, n)
    want = case.expected(n)
    v = case.variants[n]
    assert_parity(got, want, case.max_shift, "%s v%d" % (name, n), subpx=v.get("subpx", -1))
    ok = want["err"] == 1
    if "debug_d" in want:
        np.testing.assert_allclose(got["debug_d"], want["debug_d"], rtol=1e-8, atol=1e-13)
        np.testing.assert_allclose(got["debug_a"] * ok[..., None], want["debug_a"], rtol=1e-8, atol=1e-13)
    if "coverage" in want:
        np.testing.assert_array_equal(got["coverage"], want["coverage"])


@pytest.mark.parametrize("name", ["B_walks", "F7_C1_ms4", "F8_C2_crop", "F8_C3_crop"])
@pytest.mark.parametrize("force", ["direct", "tiled"])
def test_both_kernels_agree_with_golden(hip_ns, name, force):
    """The general direct kernel and the tiled fast path must each meet the bar on their own."""
    from umpa_amd import _lib
    case = Case(name)
    for n, v in enumerate(case.variants):
        cls = getattr(hip_ns, v["model"])
        m = cls(case.sam, case.ref, window_size=case.Nw, max_shift=case.max_shift)
        m.assign_coordinates = v.get("assign", "sam")
        m.sub_pixel_mode = v.get("subpx", -1)
        m._force = _lib.F_FORCE_DIRECT if force == "direct" else _lib.F_FORCE_TILED
        try:
            got = m.match(quiet=True)
        except _lib.NativeError as e:
            if force == "tiled" and "does not cover" in str(e):
                pytest.skip("tiled path does not cover this configuration yet")
            raise
        assert m._lib.last_path(m._handle) in ((1, 3) if force == "direct" else (2,))
        assert_parity(got, case.expected(n), case.max_shift, "%s v%d %s" % (name, n, force),
                      subpx=v.get("subpx", -1))


def test_dfkernel_single_pixel_and_mask(hip_ns, port_ns):
    """Kernel-dark-field model: min()/cost() entry points and the mask-weighted blur (Utils.cpp:103-117)."""
    case = Case("E_dfkernel")
    rng = np.random.default_rng(3)
    mask = (rng.uniform(size=case.sam.shape) < 0.9).astype(np.float64)
    for mk in (None, mask):
        g = hip_ns.UMPAModelDFKernel(case.sam, case.ref, mask_list=mk, window_size=case.Nw, max_shift=case.max_shift)
        o = port_ns.UMPAModelDFKernel(case.sam, case.ref, mask_list=mk, window_size=case.Nw, max_shift=case.max_shift)
        assert g.padding == case.Nw + case.max_shift + 8
        for (i, j) in [(15, 16), (20, 30), (33, 37)]:
            np.testing.assert_allclose(g.cost(i, j, 1, -2, 0.2, 0.03, 0.1), o.cost(i, j, 1, -2, 0.2, 0.03, 0.1), rtol=1e-9)
            np.testing.assert_allclose(g.min(i, j, 0.1, 0.0, 0.1), o.min(i, j, 0.1, 0.0, 0.1), rtol=1e-6, atol=1e-9)
        sh = ((g.sh[0] + 2) // 3, (g.sh[1] + 2) // 3)
        abc = np.zeros(sh + (3,)); abc[..., 0] = 0.15; abc[..., 2] = 0.1
        got, want = g.match(abc=abc, step=3, quiet=True), o.match(abc=abc, step=3, quiet=True)
        assert_parity(got, want, case.max_shift, "DFKernel mask=%s" % (mk is not None))


@pytest.mark.parametrize("cfg", [
    dict(H=200, W=260, K=5, Nw=4, ms=5, df=True, amp=2.5),
    dict(H=160, W=300, K=3, Nw=2, ms=3, df=False, amp=1.0),
    dict(H=130, W=150, K=8, Nw=6, ms=7, df=True, amp=4.5),
    dict(H=90, W=90, K=1, Nw=1, ms=4, df=True, amp=2.0),
    dict(H=64, W=80, K=2, Nw=0, ms=4, df=False, amp=2.0),
    dict(H=77, W=91, K=3, Nw=3, ms=4, df=True, amp=2.0),       # odd output width: unaligned table rows in the tiled path
    dict(H=101, W=67, K=2, Nw=6, ms=3, df=False, amp=0.3),
    dict(H=120, W=131, K=3, Nw=8, ms=5, df=True, amp=2.0),      # widest window the tiled path is built for
    dict(H=70, W=75, K=18, Nw=2, ms=4, df=True, amp=1.5),       # more frames than replay_walk keeps in registers     # 32x32-tile shape of the tiled path (Nw > 5), small search range
    dict(H=64, W=72, K=25, Nw=2, ms=3, df=True, amp=1.0),       # odd frame count beyond the templated ones: generic replay_walk on the frame-pair maps
    dict(H=64, W=70, K=26, Nw=3, ms=3, df=True, amp=1.0),       # even, same path
])
def test_hip_matches_oracle_on_seeded_inputs(hip_ns, port_ns, cfg):
    from umpa_amd.synth import make_stack
    sam, ref, _ = make_stack(cfg["H"], cfg["W"], cfg["K"], cfg["ms"], df=cfg["df"], seed=123, amplitude=cfg["amp"])
    name = "UMPAModelDF" if cfg["df"] else "UMPAModelNoDF"
    for assign in ("sam", "ref"):
        g = getattr(hip_ns, name)(sam, ref, window_size=cfg["Nw"], max_shift=cfg["ms"])
        o = getattr(port_ns, name)(sam, ref, window_size=cfg["Nw"], max_shift=cfg["ms"])
        g.assign_coordinates = o.assign_coordinates = assign
        got, want = g.match(quiet=True), o.match(quiet=True)
        st = assert_parity(got, want, cfg["ms"], "%s %s %s" % (name, assign, cfg))
        assert st["ok"] > 0


def test_single_pixel_entry_points(hip_ns, port_ns):
    case = Case("A_small")
    for name in ("UMPAModelDF", "UMPAModelNoDF"):
        g = getattr(hip_ns, name)(case.sam, case.ref, window_size=2, max_shift=4)
        o = getattr(port_ns, name)(case.sam, case.ref, window_size=2, max_shift=4)
        for (i, j) in [(6, 6), (20, 33), (57, 65), (31, 8)]:
            np.testing.assert_allclose(g.min(i, j), o.min(i, j), rtol=1e-6, atol=1e-9)
            for (sx, sy) in [(0, 0), (1, -2), (-3, 3), (2.4, -0.6)]:
                np.testing.assert_allclose(g.cost(i, j, sx, sy), o.cost(i, j, sx, sy), rtol=1e-10)


def test_cost_kats_and_bound_errors(hip_ns):
    import ctypes as C
    z = np.load(os.path.join(GOLDEN, "F2_cost.npz"))
    case = Case("A_small")
    for mdl in ("UMPAModelNoDF", "UMPAModelDF"):
        for assign in ("sam", "ref"):
            m = getattr(hip_ns, mdl)(case.sam, case.ref, window_size=2, max_shift=4)
            m.assign_coordinates = assign
            want = z["%s_%s" % (mdl, assign)]
            for n, (i, j, si, sj) in enumerate(z["pts"][:80]):
                if abs(si) >= 4 or abs(sj) >= 4:
                    vals = np.zeros(3)
                    st = m._lib.cost(m._handle, int(i), int(j), int(si), int(sj), vals.ctypes.data_as(C.POINTER(C.c_double)))
                    assert st & 2 and not st & 1
                else:
                    np.testing.assert_allclose(m.cost(int(i), int(j), float(si), float(sj)), want[n], rtol=1e-10, atol=1e-16)


def test_subpixel_kats_on_device(hip_ns):
    z = np.load(os.path.join(GOLDEN, "F1_subpixel.npz"))
    for n in range(0, 256, 4):
        pos, v = hip_ns.spmq(z["a"][n])               # reference naming: spmq -> spmin
        np.testing.assert_allclose(pos, z["spmin_pos"][n], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(v, z["spmin_val"][n], rtol=1e-7)
        pos, v = hip_ns.spm(z["a"][n])                # spm -> spmin_quad
        np.testing.assert_allclose(pos, z["quad_pos"][n], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(v, z["quad_val"][n], rtol=1e-9)


def test_api_conventions(hip_ns):
    """Result dictionary layout, dtypes and error behaviour of the reference API."""
    import umpa_amd
    case = Case("A_small")
    r = umpa_amd.match(case.sam, case.ref, 2, step=2, max_shift=1)      # max_shift is ignored, as in the reference
    assert list(r.keys()) == ['err', 'debug_d', 'debug_a', 'debug_Ncalls', 'f', 'T', 'dx', 'dy', 'df']
    assert r['f'].dtype == np.float64 and r['err'].dtype == np.int32 and r['f'].flags.c_contiguous
    m = hip_ns.UMPAModelDF(case.sam, case.ref, window_size=2)
    assert m.max_shift == 4 and m.padding == 6 and m.extent == (64 - 12, 72 - 12)
    assert r['f'].shape == ((m.extent[0] + 1) // 2, (m.extent[1] + 1) // 2)
    r2 = umpa_amd.match(case.sam, case.ref, 2, step=2, df=False)
    assert 'df' not in r2
    rb = umpa_amd.match_unbiased(case.sam, case.ref, 2, step=2)          # shares the resident reference stack
    rr = umpa_amd.match(case.ref, case.ref, 2, step=2)
    np.testing.assert_array_equal(rb['dx'], r['dx'] - rr['dx'])
    np.testing.assert_array_equal(rb['dy'], r['dy'] - rr['dy'])
    np.testing.assert_array_equal(rb['T'], r['T'])
    ru = umpa_amd.match_unbiased(case.sam, case.ref, 2, step=2, bias=(0.25, -0.5))
    np.testing.assert_allclose(ru['dx'], r['dx'] - 0.25)
    np.testing.assert_allclose(ru['dy'], r['dy'] + 0.5)
    with pytest.raises(RuntimeError, match="not C-contiguous"):
        hip_ns.UMPAModelDF(case.sam[:, :, ::2], case.ref[:, :, ::2])
    with pytest.raises(RuntimeError, match="Incompatible shape"):
        hip_ns.UMPAModelDF(case.sam, case.ref[:, :-1, :].copy())
    with pytest.raises(RuntimeError, match="Positions should start at 0"):
        hip_ns.UMPAModelDF(case.sam, case.ref, pos_list=[np.array([1, 1])] * 3)
    # float32 input is converted, not silently misread
    r32 = hip_ns.UMPAModelDF(case.sam.astype(np.float32), case.ref.astype(np.float32), window_size=2).match(step=2, quiet=True)
    assert np.abs(r32['T'] - r['T'])[r['err'] == 1].max() < 1e-3


def test_pixels_are_independent_at_full_width(hip_ns):
    """Size-independent property: a stepped / ROI match equals slices of the full match."""
    from umpa_amd.synth import make_stack
    sam, ref, _ = make_stack(300, 2048, 4, 5, df=True, seed=9, amplitude=2.5, order=1)
    from umpa_amd import _lib
    m = hip_ns.UMPAModelDF(sam, ref, window_size=4, max_shift=5)
    m.debug = False
    m._force = _lib.F_FORCE_DIRECT        # one kernel for all three runs: the results must then be bit-identical
    full = m.match(quiet=True)
    m.ROI = None
    part = m.match(step=3, quiet=True)
    for k in ("f", "T", "dx", "dy", "df", "err"):
        np.testing.assert_array_equal(part[k], full[k][::3, ::3])
    m.ROI = None
    roi = m.match(ROI=((10, 200, 2), (1000, 2030, 5)), quiet=True)
    for k in ("f", "T", "dx", "dy", "df", "err"):
        np.testing.assert_array_equal(roi[k], full[k][10:200:2, 1000:2030:5])


def test_update_frames_and_projection_farm(hip_ns):
    """Extension used by the projection farm: swap the sample stack, keep the reference resident."""
    from umpa_amd.farm import ProjectionFarm
    from umpa_amd.synth import make_stack
    Nw, ms = 3, 4
    stacks = [make_stack(80, 96, 4, ms, df=True, seed=10 * p, amplitude=1.5) for p in range(3)]
    ref = stacks[0][1]
    sams = {p: np.ascontiguousarray(stacks[p][0]) for p in range(3)}
    want = {p: hip_ns.UMPAModelDF(sams[p], ref, window_size=Nw, max_shift=ms).match(quiet=True) for p in range(3)}
    m = hip_ns.UMPAModelDF(sams[0], ref, window_size=Nw, max_shift=ms)
    for p in (1, 2, 0):
        m.update_frames(sam_list=sams[p])
        got = m.match(quiet=True)
        for k in ("f", "T", "dx", "dy", "df", "err"):
            np.testing.assert_array_equal(got[k], want[p][k])
    with pytest.raises(RuntimeError, match="shapes"):
        m.update_frames(sam_list=sams[0][:, :-2])
    with ProjectionFarm(ref, Nw, ms, devices=[0]) as farm:
        res = dict(farm.map(sams.items()))
    for p in range(3):
        for k in ("f", "T", "dx", "dy", "df", "err"):
            np.testing.assert_array_equal(res[p][k], want[p][k])


def test_C5_shape_step_scan_against_the_oracle(hip_ns, port_ns):
    """BASELINE config C5's parameters (5 frames, Nw=5, max_shift=5, dark-field on) on 512 x 512 projections: the streaming
    pipeline (uint16 counts uploaded into the back buffer + fused flat correction while the previous projection is matched,
    nearest-reference switching, results in page-locked memory) and the worker-process farm, both against the CPU oracle
    on what the reference script computes: (proj - dark) / flat[refnum] matched against ref[refnum] (umpa_multi.py:133-150)."""
    from umpa_amd.farm import ProjectionFarm, StreamingMatcher, nearest_reference
    from umpa_amd.synth import make_stack
    Nw, ms, K, n = 5, 5, 5, 512
    base = [make_stack(n, n, K, ms, df=True, seed=40 + 10 * r, order=1) for r in range(2)]
    refs = np.stack([b[1] for b in base])
    rng = np.random.default_rng(9)
    dark = 100.0 + rng.uniform(0, 2, size=(K, n, n))
    flats = 20000.0 * (1.0 + 0.05 * rng.standard_normal((2, K, n, n)))
    ref_nums = [0, 5]
    raws, want = {}, {}
    for p in range(6):
        r = nearest_reference(p, ref_nums)
        raws[p] = np.rint(np.roll(base[r][0], p, axis=2) * flats[r] + dark).astype(np.uint16)
        sam = (raws[p].astype(np.float64) - dark) / flats[r]
        m = port_ns.UMPAModelDF(sam, refs[r], window_size=Nw, max_shift=ms)
        m.debug = True
        want[p] = m.match(ROI=((100, 140, 1), (0, n - 2 * (Nw + ms), 1)), quiet=True)       # a 40-row band per projection
    sm = StreamingMatcher(refs, Nw, ms, df=True, device=0, flats=flats, dark=dark, ref_nums=ref_nums, debug=True)
    bufs = {}
    for p in raws:
        bufs[p] = sm.input_buffer(np.uint16)
        bufs[p][...] = raws[p]
    seen = []
    for pid, res in sm.run((p, bufs[p]) for p in range(6)):
        seen.append(pid)
        got = {k: (v[100:140] if isinstance(v, np.ndarray) else v) for k, v in res.items()}
        assert_parity(got, want[pid], ms, "C5 stream p%d" % pid)
    assert seen == list(range(6))
    with ProjectionFarm(refs, Nw, ms, devices=[0], flats=flats, dark=dark, ref_nums=ref_nums, raw_dtype=np.uint16) as farm:
        res = dict(farm.map(raws.items()))
    assert sorted(res) == list(range(6))
    for p in range(6):
        # the farm returns the maps only (no debug arrays): bit-exact walk outcome, maps to the bar on all but the
        # unconverged-Newton pixels the streaming leg has just classified
        np.testing.assert_array_equal(res[p]["err"][100:140], want[p]["err"])
        ok = want[p]["err"] == 1
        for k in ("T", "df"):
            np.testing.assert_allclose(res[p][k][100:140][ok], want[p][k][ok], rtol=1e-5)
        close = np.abs(res[p]["dx"][100:140] - want[p]["dx"]) <= 1e-5 * np.maximum(1.0, np.abs(want[p]["dx"]))
        assert (~close & ok).sum() <= 4


@pytest.mark.parametrize("name", ["B_walks", "D_stepping", "A_small"])
def test_staged_and_plain_direct_kernels_are_identical(hip_ns, name):
    """The general kernel with the windows served from LDS (path 3) sums the same terms in the same order as the one
    that reads them through L1 (path 1): bit-identical maps, for sample stepping with unequal shapes, both coordinate
    conventions, steps and start shifts.  (Masked models never take the staged kernel: umpa_masked.h is their path.)"""
    from umpa_amd import _lib
    case = Case(name)
    for n, v in enumerate(case.variants):
        out = {}
        for tag, force in (("staged", _lib.F_FORCE_DIRECT), ("plain", _lib.F_FORCE_DIRECT | _lib.F_FORCE_PLAIN_DIRECT)):
            _, m0 = None, None
            orig = getattr(hip_ns, v["model"])._force
            try:
                getattr(hip_ns, v["model"])._force = force
                got, m = case.run(hip_ns, n)
                out[tag] = (got, m._lib.last_path(m._handle))
            finally:
                getattr(hip_ns, v["model"])._force = orig
        assert out["plain"][1] == 1
        if out["staged"][1] != 3:
            continue                                                # region too small or too spread out for the staged kernel
        for k in ("f", "T", "dx", "dy", "err", "debug_Ncalls", "debug_d", "debug_a"):
            np.testing.assert_array_equal(out["staged"][0][k], out["plain"][0][k], err_msg="%s v%d %s" % (name, n, k))


@pytest.mark.parametrize("variant", ["plain", "ref_step2_dxdy", "roi_inside"])
def test_sample_stepping_on_the_tiled_path(hip_ns, port_ns, variant):
    """Frames of one shape at different positions (sample stepping): the rectangle every frame contributes to runs on the
    tiled path (per-frame positions folded into the staging addresses), the border strips on the general kernels
    (path 4); a ROI inside the fully covered rectangle is the tiled path alone (path 2).  All against the CPU oracle."""
    from umpa_amd.synth import make_stack
    Nw, ms, K = 3, 4, 6
    pos = [np.array(p) for p in [(0, 0), (0, 14), (9, 0), (9, 14), (4, 7), (0, 0)]]
    frames = [make_stack(150, 170, 1, ms, df=True, seed=300 + k, amplitude=2.0) for k in range(K)]
    sam = [np.ascontiguousarray(f[0][0]) for f in frames]
    ref = [np.ascontiguousarray(f[1][0]) for f in frames]
    kw = dict(window_size=Nw, max_shift=ms, pos_list=pos)
    g, o = hip_ns.UMPAModelDF(sam, ref, **kw), port_ns.UMPAModelDF(sam, ref, **kw)
    mk = dict(quiet=True)
    if variant == "ref_step2_dxdy":
        g.assign_coordinates = o.assign_coordinates = "ref"
        mk.update(step=2, dxdy=(1, -1))
    elif variant == "roi_inside":
        mk.update(ROI=((12, 120, 1), (16, 150, 1)))
    got, want = g.match(**mk), o.match(**mk)
    assert g._lib.last_path(g._handle) == (2 if variant == "roi_inside" else 4)
    st = assert_parity(got, want, ms, "stepping tiled %s" % variant)
    assert st["ok"] > 1000
    np.testing.assert_array_equal(g.coverage(), o.coverage())


@pytest.mark.parametrize("df,assign,masked", [(True, "sam", False), (False, "ref", False), (True, "ref", False),
                                              (True, "sam", True), (False, "ref", True)])
def test_sample_stepping_border_rectangles_on_the_tiled_path(hip_ns, port_ns, df, assign, masked):
    """A 2 x 2 grid of positions: the region falls into rectangles with a constant set of contributing frames.  The large
    ones (the centre with every frame, the four edges with half of them, the corners with a quarter) each run on the tiled
    path with their subset's frames -- the cost still divided by the model's frame count --, the slivers (a frame's last
    contributing column, small rectangles) on the general kernels.  Against the CPU oracle, with a coverage threshold."""
    import ctypes
    from umpa_amd.synth import make_stack
    Nw, ms, K = 2, 3, 8
    pos = [np.array(p) for p in [(0, 0), (0, 150), (120, 0), (120, 150), (0, 0), (120, 150), (0, 150), (120, 0)]]
    frames = [make_stack(330, 380, 1, ms, df=True, seed=500 + k, amplitude=1.5) for k in range(K)]
    sam = [np.ascontiguousarray(f[0][0]) for f in frames]
    ref = [np.ascontiguousarray(f[1][0]) for f in frames]
    kw = dict(window_size=Nw, max_shift=ms, pos_list=pos)
    if masked:                                                      # the masked table kernel takes the same rectangles
        rng = np.random.default_rng(21)
        kw["mask_list"] = [(rng.random(a.shape) < 0.93).astype(np.float64) for a in sam]
    name = "UMPAModelDF" if df else "UMPAModelNoDF"
    g, o = getattr(hip_ns, name)(sam, ref, **kw), getattr(port_ns, name)(sam, ref, **kw)
    g.assign_coordinates = o.assign_coordinates = assign
    lib, h = g._lib, g._handle
    lib.timing_enable(h, 1)
    got, want = g.match(quiet=True), o.match(quiet=True)
    assert lib.last_path(h) == 4
    launches = {}
    for q in range(lib.timing_collect(h)):
        nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
        lib.timing_read(h, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
        launches[nm.value.decode()] = cnt.value
    lib.timing_enable(h, 0)
    assert launches.get("corr_masked" if masked else "corr_volume", 0) >= 9, launches      # centre + 4 edges + 4 corners
    st = assert_parity(got, want, ms, "stepping rectangles %s %s%s" % (name, assign, " masked" if masked else ""))
    assert st["ok"] > 100000
    # the same through a ROI with steps and a start shift (the rectangles move with the region's origin and step)
    mk = dict(quiet=True, ROI=((3, 400, 2), (5, 480, 1)), dxdy=(1, -1))
    got, want = g.match(**mk), o.match(**mk)
    assert_parity(got, want, ms, "stepping rectangles ROI %s %s%s" % (name, assign, " masked" if masked else ""))
    if masked:
        np.testing.assert_array_equal(g.coverage(), o.coverage())
    # a second match of the same model takes the cached descriptor lists
    got2 = g.match(**mk)
    for k in ("f", "dx", "dy", "T", "err"):
        np.testing.assert_array_equal(got2[k], got[k])


def test_plain_c_host_of_the_c_abi(hip_ns, tmp_path):
    """examples/c_host.c: the boundary is a C ABI -- a host with no Python and no PyTorch in the process."""
    import subprocess
    from conftest import REPO
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-O2", "-I" + os.path.join(REPO, "include"), os.path.join(REPO, "examples", "c_host.c"),
                    "-o", exe, "-L" + os.path.join(REPO, "umpa_amd"), "-lumpa_hip",
                    "-Wl,-rpath," + os.path.join(REPO, "umpa_amd"), "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "path = 2" in out.stdout                    # the tiled fast path served it


def test_stepped_and_chunked_tiled_path_matches_direct(hip_ns, monkeypatch):
    """Small steps run on the tiled path over the dense grid; a tiny table budget forces several row chunks."""
    from umpa_amd import _lib
    from umpa_amd.synth import make_stack
    monkeypatch.setenv("UMPA_HIP_TABLE_MB", "16")
    sam, ref, _ = make_stack(400, 300, 4, 4, df=True, seed=77, amplitude=1.5, order=1)
    res = {}
    for force in ("tiled", "direct"):
        m = hip_ns.UMPAModelDF(sam, ref, window_size=3, max_shift=4)
        m.debug = "ncalls"
        m._force = _lib.F_FORCE_TILED if force == "tiled" else _lib.F_FORCE_DIRECT
        res[force] = [m.match(ROI=((3, 380, 2), (5, 280, 3)), quiet=True), m.match(step=3, quiet=True),
                      m.match(ROI=((0, 386, 1), (0, 286, 1)), quiet=True), m.match(step=5, quiet=True),
                      m.match(ROI=((2, 385, 7), (1, 285, 4)), quiet=True)]
        assert m._lib.last_path(m._handle) in ((2,) if force == "tiled" else (1, 3))
    for a, b in zip(res["tiled"], res["direct"]):
        np.testing.assert_array_equal(a["err"], b["err"])
        np.testing.assert_array_equal(a["debug_Ncalls"], b["debug_Ncalls"])
        ok = b["err"] == 1
        np.testing.assert_allclose(a["T"][ok], b["T"][ok], rtol=1e-9)
        np.testing.assert_allclose(a["df"][ok], b["df"][ok], rtol=1e-9)
        assert np.mean(np.abs(a["dx"] - b["dx"])[ok] > 1e-6) < 2e-3
    with pytest.raises(_lib.NativeError, match="does not cover"):
        m = hip_ns.UMPAModelDF(sam, ref, window_size=3, max_shift=4)
        m._force = _lib.F_FORCE_TILED
        m.match(step=10, quiet=True)                      # 100 dense pixels per output pixel: left to the direct kernel


def test_degenerate_sizes_on_device(hip_ns, port_ns):
    """One-pixel and one-row outputs, constant frames (zero-variance windows): same answers as the oracle."""
    from umpa_amd.synth import make_stack
    sam, ref, _ = make_stack(13, 13, 2, 4, df=True, seed=1, amplitude=0.5)
    for name in ("UMPAModelDF", "UMPAModelNoDF"):
        g = getattr(hip_ns, name)(sam, ref, window_size=2, max_shift=4).match(quiet=True)
        o = getattr(port_ns, name)(sam, ref, window_size=2, max_shift=4).match(quiet=True)
        assert g["f"].shape == (1, 1)
        np.testing.assert_array_equal(g["err"], o["err"])
        np.testing.assert_array_equal(g["debug_Ncalls"], o["debug_Ncalls"])
    sam, ref, _ = make_stack(15, 90, 3, 4, df=True, seed=2, amplitude=0.5)       # 3 output rows
    g = hip_ns.UMPAModelDF(sam, ref, window_size=2, max_shift=4).match(quiet=True)
    o = port_ns.UMPAModelDF(sam, ref, window_size=2, max_shift=4).match(quiet=True)
    assert_parity(g, o, 4, "3-row output")
    with pytest.raises(RuntimeError, match="Empty ROI"):
        hip_ns.UMPAModelDF(np.ones((2, 12, 30)), np.ones((2, 12, 30)), window_size=2, max_shift=4).match(quiet=True)


def test_reference_side_maps_are_reused_and_invalidated(hip_ns):
    """UMPA_HIP_F_REUSE_REF_MAPS: a second match of one model recomputes only the sample-side maps; swapping the
    sample stack keeps the reference side, swapping the reference stack or the window drops it.  Every result must
    equal the one of a freshly built model, bit for bit."""
    from umpa_amd.synth import make_stack
    samA, refA, _ = make_stack(120, 140, 6, 4, df=True, seed=5, amplitude=1.5)
    samB, refB, _ = make_stack(120, 140, 6, 4, df=True, seed=9, amplitude=2.0)

    def fresh(sam, ref, Nw=3):
        return hip_ns.UMPAModelDF(sam, ref, window_size=Nw, max_shift=4).match(quiet=True)

    def same(a, b):
        for k in ("err", "debug_Ncalls", "T", "df", "dx", "dy", "f"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k

    m = hip_ns.UMPAModelDF(samA, refA, window_size=3, max_shift=4)
    first = m.match(quiet=True)
    assert m._lib.last_path(m._handle) == 2                            # the tiled path, where the maps live
    same(m.match(quiet=True), first)                                   # second match: reference side reused
    m.update_frames(sam_list=samB)
    same(m.match(quiet=True), fresh(samB, refA))                       # new sample stack, old reference maps
    m.update_frames(ref_list=refB)
    same(m.match(quiet=True), fresh(samB, refB))                       # new reference stack: recomputed
    m.Nw = 2
    same(m.match(quiet=True), _with_nw(hip_ns, samB, refB, 2))        # new window: every map recomputed


def _with_nw(hip_ns, sam, ref, nw):
    m = hip_ns.UMPAModelDF(sam, ref, window_size=3, max_shift=4)
    m.Nw = nw
    return m.match(quiet=True)


@pytest.mark.parametrize("name", ["C_mask", "C_mask_ones"])
def test_masked_goldens_on_the_tiled_path(hip_ns, name):
    """Models with masks run on the tiled path (corr_masked + replay_cost, umpa_masked.h): the goldens frozen from the
    reference, with the path forced so that a silent fall-back to the general kernel fails."""
    from umpa_amd import _lib
    case = Case(name)
    for n, v in enumerate(case.variants):
        cls = getattr(hip_ns, v["model"])
        orig = cls._force
        try:
            cls._force = _lib.F_FORCE_TILED
            got, m = case.run(hip_ns, n)
        finally:
            cls._force = orig
        assert m._lib.last_path(m._handle) == 2
        assert_parity(got, case.expected(n), case.max_shift, "%s v%d" % (name, n), subpx=v.get("subpx", -1))


@pytest.mark.parametrize("cfg", [
    dict(H=150, W=170, K=4, Nw=5, ms=5, df=True, amp=2.5, kind="binary"),
    dict(H=150, W=170, K=4, Nw=5, ms=5, df=False, amp=2.5, kind="binary"),
    dict(H=120, W=131, K=3, Nw=3, ms=4, df=True, amp=2.0, kind="weights"),     # real-valued weights, odd output width
    dict(H=120, W=131, K=3, Nw=3, ms=4, df=False, amp=2.0, kind="weights"),
    dict(H=90, W=100, K=10, Nw=1, ms=3, df=True, amp=1.0, kind="binary"),       # 3x3 windows: enough frames to keep the fits well-posed
    dict(H=100, W=90, K=5, Nw=2, ms=2, df=False, amp=0.5, kind="binary"),
    dict(H=110, W=120, K=3, Nw=4, ms=6, df=True, amp=3.0, kind="blocks"),      # whole regions masked out: coverage threshold
    dict(H=130, W=140, K=2, Nw=6, ms=3, df=True, amp=1.0, kind="weights"),
    dict(H=140, W=150, K=3, Nw=7, ms=4, df=True, amp=2.0, kind="binary"),      # two column offsets per pass
    dict(H=140, W=150, K=2, Nw=8, ms=3, df=False, amp=1.0, kind="weights"),
    dict(H=100, W=100, K=11, Nw=2, ms=4, df=True, amp=2.0, kind="binary"),     # odd frame count: the frame-pair layout of the means
])
def test_masked_models_against_the_oracle(hip_ns, port_ns, cfg):
    """corr_masked / replay_cost against the CPU oracle: binary masks, real-valued weights, masked-out blocks, both
    models, both coordinate conventions, steps and start shifts; every window half-width the kernel is built for."""
    from umpa_amd import _lib
    from umpa_amd.synth import make_stack
    sam, ref, _ = make_stack(cfg["H"], cfg["W"], cfg["K"], cfg["ms"], df=cfg["df"], seed=77, amplitude=cfg["amp"], order=1)
    rng = np.random.default_rng(5)
    if cfg["kind"] == "binary":
        mask = (rng.random(sam.shape) < 0.9).astype(np.float64)
    elif cfg["kind"] == "weights":
        mask = rng.uniform(0.0, 1.0, size=sam.shape) * (rng.random(sam.shape) < 0.95)
    else:
        mask = np.ones(sam.shape)
        mask[:, 30:60, 40:80] = 0.0
        mask[0, :, :20] = 0.0
    name = "UMPAModelDF" if cfg["df"] else "UMPAModelNoDF"
    for assign, mk in (("sam", dict()), ("ref", dict(step=2, dxdy=(1, -1)))):
        g = getattr(hip_ns, name)(sam, ref, mask_list=mask, window_size=cfg["Nw"], max_shift=cfg["ms"])
        o = getattr(port_ns, name)(sam, ref, mask_list=mask, window_size=cfg["Nw"], max_shift=cfg["ms"])
        g.assign_coordinates = o.assign_coordinates = assign
        g._force = _lib.F_FORCE_TILED
        got, want = g.match(quiet=True, **mk), o.match(quiet=True, **mk)
        assert g._lib.last_path(g._handle) == 2
        st = assert_parity(got, want, cfg["ms"], "masked %s %s %s" % (name, assign, cfg))
        assert st["ok"] > 0 or cfg["ms"] <= 2               # (max_shift 2 has no room for the 4 x 4 gather: every walk fails, as in BASELINE config C1)


@pytest.mark.parametrize("cfg", [
    dict(H=420, W=520, K=4, Nw=3, ms=6, df=True, amp=4.5, assign="sam", mask=False, mk=dict()),
    dict(H=420, W=520, K=3, Nw=2, ms=8, df=False, amp=6.0, assign="ref", mask=False, mk=dict(step=2, dxdy=(1, -1))),
    dict(H=300, W=700, K=5, Nw=5, ms=5, df=True, amp=3.5, assign="sam", mask=False, mk=dict()),
    dict(H=330, W=400, K=4, Nw=4, ms=5, df=True, amp=3.5, assign="sam", mask=True, mk=dict()),
    dict(H=330, W=400, K=4, Nw=4, ms=5, df=False, amp=3.5, assign="ref", mask=True, mk=dict(step=2)),
    # windows of 13 / 15 pixels: corr_march's units are (strip, band, pass); the prediction is a lattice of sample pixels
    # (od_run_chunk_march); bands of 64 rows so that these small images have several
    dict(H=420, W=520, K=4, Nw=7, ms=8, df=True, amp=6.0, assign="sam", mask=False, mk=dict(), march_rows=64),
    dict(H=380, W=300, K=3, Nw=6, ms=6, df=False, amp=4.5, assign="ref", mask=False, mk=dict(step=2, dxdy=(1, -1)), march_rows=64),
    dict(H=300, W=410, K=5, Nw=7, ms=10, df=True, amp=7.5, assign="ref", mask=False, mk=dict(), march_rows=96),
])
def test_on_demand_table_passes_change_nothing(hip_ns, monkeypatch, cfg):
    """umpa_ondemand.h: only the (tile, pass) units of the shift table that walks read are computed -- seed tiles predict,
    parked pixels and repair rounds make up for every misprediction.  The maps must be those of the exhaustive table
    bit for bit, debug arrays included; the counters show that passes were left out and that walks were parked."""
    import ctypes
    from umpa_amd import _lib
    from umpa_amd.synth import make_stack
    sam, ref, _ = make_stack(cfg["H"], cfg["W"], cfg["K"], cfg["ms"], df=cfg["df"], seed=31, amplitude=cfg["amp"], order=1)
    mask = None
    if cfg["mask"]:
        mask = (np.random.default_rng(2).random(sam.shape) < 0.93).astype(np.float64)
    cls = hip_ns.UMPAModelDF if cfg["df"] else hip_ns.UMPAModelNoDF
    out, stats = {}, {}
    if "march_rows" in cfg:
        monkeypatch.setenv("UMPA_HIP_MARCH_OD_ROWS", str(cfg["march_rows"]))
    for od in ("0", "1", "lattice"):
        # "lattice": corr_volume's / corr_masked's stages with the sample-lattice prediction (corr_march's only one) instead of seed tiles
        if od == "lattice" and "march_rows" in cfg:
            continue
        monkeypatch.setenv("UMPA_HIP_ONDEMAND", "0" if od == "0" else "1")
        monkeypatch.setenv("UMPA_HIP_OD_PRED", "lattice" if od == "lattice" else "seed")
        m = cls(sam, ref, mask_list=mask, window_size=cfg["Nw"], max_shift=cfg["ms"])
        m.assign_coordinates = cfg["assign"]
        m._force = _lib.F_FORCE_TILED
        out[od] = m.match(quiet=True, **cfg["mk"])
        st = (ctypes.c_double * 4)()
        m._lib.check(m._lib.last_stats(m._handle, st), "last_stats")
        stats[od] = list(st)
    for od in out:
        for k in out["0"]:
            assert np.array_equal(out["0"][k], out[od][k], equal_nan=True), (od, k)
    assert stats["0"][0] == stats["0"][1] and stats["0"][2] == 0          # exhaustive: every unit, nothing parked
    if "march_rows" in cfg:
        # (strip, band, pass) units: a pass is two or three row offsets deep and every walk's 4 x 4 gather reaches two rows past its
        # minimum, so on images this small every unit may be needed; what is pinned is the bookkeeping and the repair rounds
        assert 0 < stats["1"][0] <= stats["1"][1], stats
    else:
        assert stats["1"][1] == stats["0"][1] and stats["1"][0] < stats["1"][1], stats   # (harsh fields on small images: few passes go unread)
    assert stats["1"][2] > 0, stats                                       # some walks were parked and run again


@pytest.mark.parametrize("df", [True, False])
def test_masks_and_sample_stepping_together(hip_ns, port_ns, df):
    """Masks AND frames at different positions: the rectangle every frame contributes to runs on the masked tiled path
    (frame positions folded into the staging addresses of frames and masks), the border strips on the general kernels
    (path 4); against the CPU oracle, coverage map (mask-weighted) included."""
    from umpa_amd.synth import make_stack
    Nw, ms, K = 3, 4, 5
    pos = [np.array(p) for p in [(0, 0), (0, 12), (8, 0), (8, 12), (4, 6)]]
    frames = [make_stack(140, 160, 1, ms, df=df, seed=500 + k, amplitude=2.0) for k in range(K)]
    sam = [np.ascontiguousarray(f[0][0]) for f in frames]
    ref = [np.ascontiguousarray(f[1][0]) for f in frames]
    rng = np.random.default_rng(8)
    mask = [(rng.random(s.shape) < 0.92).astype(np.float64) for s in sam]
    name = "UMPAModelDF" if df else "UMPAModelNoDF"
    kw = dict(window_size=Nw, max_shift=ms, pos_list=pos, mask_list=mask)
    g, o = getattr(hip_ns, name)(sam, ref, **kw), getattr(port_ns, name)(sam, ref, **kw)
    got, want = g.match(quiet=True), o.match(quiet=True)
    assert g._lib.last_path(g._handle) == 4
    st = assert_parity(got, want, ms, "masks + stepping %s" % name)
    assert st["ok"] > 1000
    np.testing.assert_array_equal(g.coverage(), o.coverage())
    g.ROI = None
    roi = ((14, 110, 1), (18, 130, 1))                              # inside the fully covered rectangle: the tiled path alone
    got, want = g.match(ROI=roi, quiet=True), o.match(ROI=roi, quiet=True)
    assert g._lib.last_path(g._handle) == 2
    assert_parity(got, want, ms, "masks + stepping ROI %s" % name)


@pytest.mark.parametrize("cfg", [
    dict(H=210, W=333, K=5, Nw=6, ms=5, df=True, assign="sam", mk=dict()),
    dict(H=180, W=260, K=3, Nw=7, ms=8, df=True, assign="ref", mk=dict(step=2, dxdy=(1, -1))),
    dict(H=200, W=215, K=4, Nw=7, ms=3, df=False, assign="sam", mk=dict()),                      # odd region width, few shifts
    dict(H=400, W=190, K=7, Nw=6, ms=9, df=True, assign="sam", mk=dict(), table_mb=64),           # 17 x 17 shifts, several row chunks
    dict(H=230, W=250, K=4, Nw=6, ms=4, df=True, assign="sam", mk=dict(), pos=[(0, 0), (0, 11), (9, 0), (9, 11)]),   # sample stepping
])
def test_wide_windows_take_the_marching_table_kernel(hip_ns, port_ns, monkeypatch, cfg):
    """Windows of 13 and 15 pixels: the shift table comes from corr_march (umpa_march.h: column strips marched down the rows,
    row filter as a register ring, column filter by DPP shifts, strip-blocked table) instead of corr_volume.  Against the CPU
    oracle to the full bar, and against corr_volume (UMPA_HIP_MARCH=0): the same walk on every pixel -- the two kernels sum in
    different orders, so the maps agree to rounding, not bit for bit."""
    from umpa_amd.synth import make_stack
    Nw, ms, K = cfg["Nw"], cfg["ms"], cfg["K"]
    kw = dict(window_size=Nw, max_shift=ms)
    if cfg.get("pos"):
        frames = [make_stack(cfg["H"], cfg["W"], 1, ms, df=cfg["df"], seed=500 + k, amplitude=min(2.0, ms - 1.5), order=1) for k in range(K)]
        sam = [np.ascontiguousarray(f[0][0]) for f in frames]
        ref = [np.ascontiguousarray(f[1][0]) for f in frames]
        kw["pos_list"] = [np.array(p) for p in cfg["pos"]]
    else:
        sam, ref, _ = make_stack(cfg["H"], cfg["W"], K, ms, df=cfg["df"], seed=77, amplitude=min(3.0, ms - 1.5), order=1)
    name = "UMPAModelDF" if cfg["df"] else "UMPAModelNoDF"
    if cfg.get("table_mb"):
        monkeypatch.setenv("UMPA_HIP_TABLE_MB", str(cfg["table_mb"]))
    res = {}
    for march in ("1", "0"):
        monkeypatch.setenv("UMPA_HIP_MARCH", march)
        g = getattr(hip_ns, name)(sam, ref, **kw)
        g.assign_coordinates = cfg["assign"]
        g.debug = True
        g._lib.timing_enable(g._handle, 1)
        res[march] = g.match(quiet=True, **cfg["mk"])
        g._lib.timing_enable(g._handle, 0)
        assert g._lib.last_path(g._handle) in (2, 4)
        names = set()
        for q in range(g._lib.timing_collect(g._handle)):
            nm, tot, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_int()
            g._lib.timing_read(g._handle, q, ctypes.byref(nm), ctypes.byref(tot), ctypes.byref(cnt))
            names.add(nm.value.decode())
        assert ("corr_march" in names) == (march == "1") and ("corr_volume" in names) == (march == "0"), names
    o = getattr(port_ns, name)(sam, ref, **kw)
    o.assign_coordinates = cfg["assign"]
    o.debug = True
    want = o.match(quiet=True, **cfg["mk"])
    st = assert_parity(res["1"], want, ms, "march %s" % cfg)
    assert st["ok"] > 500
    for k in ("err", "debug_Ncalls"):
        np.testing.assert_array_equal(res["1"][k], res["0"][k], err_msg=k)
    ok = res["0"]["err"] == 1
    np.testing.assert_allclose(res["1"]["T"][ok], res["0"]["T"][ok], rtol=1e-9)
    np.testing.assert_allclose(res["1"]["debug_d"], res["0"]["debug_d"], rtol=1e-9, atol=1e-14)
