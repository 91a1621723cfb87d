"""
Seeded random sweep of the HIP path against the CPU oracle (GPU box only): shapes, frame counts, window
sizes, search ranges, both models, both coordinate conventions, the three sub-pixel modes, steps, start
shifts, masks, and both kernel paths.  Same bar as everywhere (conftest.assert_parity): err / Ncalls /
integer minimum bit-exact, float maps <= 1e-5.
"""
import os

import numpy as np
import pytest

from conftest import assert_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def namespaces():
    from umpa_amd import _lib, model
    from oracle import cpu_model
    if _lib.hip().device_count() < 1:
        pytest.fail("no HIP device: the GPU tests cannot run (there is no CPU fallback)")
    return model, cpu_model.port


SCALE = int(os.environ.get("UMPA_FUZZ_SCALE", "1"))       # soak runs: larger images (several tiles and passes per launch)


def _configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for q in range(n):
        Nw = int(rng.integers(1, 9))
        ms = int(rng.integers(1, 8))
        P = Nw + ms
        c = dict(Nw=Nw, ms=ms, K=int(rng.choice([1, 2, 3, 5, 7, 10, 13, 16, 17, 20, 24, 25, 29])),
                 H=2 * P + SCALE * int(rng.integers(8, 90)), W=2 * P + SCALE * int(rng.integers(8, 110)),
                 df=bool(rng.integers(0, 2)), assign=str(rng.choice(["sam", "ref"])),
                 subpx=int(rng.choice([-1, -1, 0, 1])), step=int(rng.choice([1, 1, 1, 2, 3])),
                 dxdy=None if rng.random() < 0.7 else (int(rng.integers(-1, 2)), int(rng.integers(-1, 2))),
                 mask=bool(rng.random() < 0.2), force=int(rng.choice([0, 0, 0, 2])),      # 2 = UMPA_HIP_F_FORCE_DIRECT
                 amp=float(rng.uniform(0.2, max(0.3, ms - 1.2))), seed=1000 + q)
        # (one frame, 3x3 windows and masks -- windows fitted exactly, costs of rounding-noise size -- are no longer re-drawn:
        #  the library sends such models to the general kernel, test_exact_fit_windows_follow_the_reference_order below)
        out.append(c)
    return out


# Windows of 3x3 or 5x5 pixels on few frames give flat cost valleys: 0.4-0.7 % of the ok pixels end on a Newton iteration
# the reference itself has not converged (soak runs, seeds 777001 and 20261003 at scale 3); every one of them is still
# classified pixel by pixel (oracle/parity.py), only the count cap for cases without a recorded count is wider there.
# A dark-field fit from one or two frames is flatter still (soak seed 99173: 1.6-1.9 % at 3x3 windows, on the general kernel as
# on the tiled path).  A single frame under a 3x3 window without dark-field is the flattest plain case (soak seed 777123 at scale 2,
# case 40: 8 of 567 ok pixels = 1.4 %, the same 8 with either pixel mapping of replay_walk).
# (three frames under a 3x3 window with dark-field: soak seed 1234567 at scale 2, sample-stepping case s16: 37 of 2932 = 1.26 %)
def _illposed_share(c):
    if c["df"] and (c["K"] <= 2 or (c["K"] <= 3 and c["Nw"] == 1)):
        return 0.025
    if c["K"] == 1 and c["Nw"] == 1:
        return 0.02
    return 0.012 if c["Nw"] <= 2 else None


# UMPA_FUZZ_N / UMPA_FUZZ_SEED: a longer or different sweep for soak runs (the committed default is what the suite runs)
CONFIGS = _configs(int(os.environ.get("UMPA_FUZZ_N", "60")), int(os.environ.get("UMPA_FUZZ_SEED", "20261003")))


@pytest.mark.parametrize("c", CONFIGS, ids=["%02d" % q for q in range(len(CONFIGS))])
def test_random_configuration(namespaces, c):
    from umpa_amd.synth import make_stack
    hip_ns, port_ns = namespaces
    sam, ref, _ = make_stack(c["H"], c["W"], c["K"], c["ms"], df=c["df"], seed=c["seed"], amplitude=c["amp"], order=1)
    mask = None
    if c["mask"]:
        rng = np.random.default_rng(c["seed"])
        mask = (rng.random(sam.shape) < 0.93).astype(np.float64)
    name = "UMPAModelDF" if c["df"] else "UMPAModelNoDF"
    models = []
    for ns in (hip_ns, port_ns):
        m = getattr(ns, name)(sam, ref, mask_list=mask, window_size=c["Nw"], max_shift=c["ms"])
        m.assign_coordinates = c["assign"]
        m.sub_pixel_mode = c["subpx"]
        models.append(m)
    g, o = models
    g._force = c["force"]
    kw = dict(step=c["step"], quiet=True)
    if c["dxdy"] is not None:
        kw["dxdy"] = c["dxdy"]
    got, want = g.match(**kw), o.match(**kw)
    assert_parity(got, want, c["ms"], str(c), subpx=c["subpx"], allow_illposed=_illposed_share(c))


def _stepping_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for q in range(n):
        Nw = int(rng.integers(1, 7))
        ms = int(rng.integers(2, 6))
        P = Nw + ms
        K = int(rng.integers(2, 6))
        span = int(rng.integers(0, 16))
        pos = rng.integers(0, span + 1, size=(K, 2))
        pos -= pos.min(axis=0)                              # per-axis minimum 0 (model.pyx:274-279)
        out.append(dict(Nw=Nw, ms=ms, K=K, pos=[tuple(int(v) for v in p) for p in pos],
                        H=2 * P + SCALE * int(rng.integers(20, 110)), W=2 * P + SCALE * int(rng.integers(20, 130)),
                        df=bool(rng.integers(0, 2)), assign=str(rng.choice(["sam", "ref"])),
                        step=int(rng.choice([1, 1, 2])), amp=float(rng.uniform(0.2, max(0.3, ms - 1.2))), seed=5000 + q))
    return out


STEPPING = _stepping_configs(int(os.environ.get("UMPA_FUZZ_N", "24")) // 2 + 12, int(os.environ.get("UMPA_FUZZ_SEED", "20261004")) + 1)


@pytest.mark.parametrize("c", STEPPING, ids=["s%02d" % q for q in range(len(STEPPING))])
def test_random_sample_stepping(namespaces, c):
    """Frames of one shape at random positions: the fully covered rectangle on the tiled path, the border strips on
    the general kernels (or everything on them when there is no such rectangle), coverage map included."""
    from umpa_amd.synth import make_stack
    hip_ns, port_ns = namespaces
    frames = [make_stack(c["H"], c["W"], 1, c["ms"], df=c["df"], seed=c["seed"] + 17 * k, amplitude=c["amp"], order=1)
              for k in range(c["K"])]
    sam = [np.ascontiguousarray(f[0][0]) for f in frames]
    ref = [np.ascontiguousarray(f[1][0]) for f in frames]
    name = "UMPAModelDF" if c["df"] else "UMPAModelNoDF"
    kw = dict(window_size=c["Nw"], max_shift=c["ms"], pos_list=[np.array(p) for p in c["pos"]])
    g, o = getattr(hip_ns, name)(sam, ref, **kw), getattr(port_ns, name)(sam, ref, **kw)
    g.assign_coordinates = o.assign_coordinates = c["assign"]
    got, want = g.match(step=c["step"], quiet=True), o.match(step=c["step"], quiet=True)
    assert_parity(got, want, c["ms"], "stepping " + str(c), allow_illposed=_illposed_share(c))
    np.testing.assert_array_equal(g.coverage(), o.coverage())


@pytest.mark.parametrize("df", [False, True], ids=["NoDF", "DF"])
def test_exact_fit_windows_follow_the_reference_order(namespaces, df):
    """The divergence a soak run of the sweep above met, pinned.  One frame, 3x3 windows and a sparse mask:
    many windows hold no more valid pixel pairs than the model has parameters (1 without dark-field, 2 with) and are fitted
    EXACTLY -- every cost on such a pixel is rounding noise around zero (|f| < 1e-20 against 1e-4 elsewhere) and the walk
    compares noise with noise: where it goes (up to the 500-call cap) depends on the order of the sums, not on the data.
    The reference-order general kernel is what the library uses for such models (path 1, never the masked tiled path); on
    every pixel whose costs are NOT noise both GPU paths and the oracle agree to the full bar, walk included.  The forced
    table-based path is allowed to differ on the noise pixels only -- that difference is the documented one."""
    from umpa_amd import _lib
    from umpa_amd.synth import make_stack
    hip_ns, port_ns = namespaces
    Nw, ms, K, n = 1, 3, 1, 64
    sam, ref, _ = make_stack(n, n + 9, K, ms, df=df, seed=4242, amplitude=1.2, order=1)
    mask = (np.random.default_rng(7).random(sam.shape) < 0.7).astype(np.float64)
    name = "UMPAModelDF" if df else "UMPAModelNoDF"
    o = getattr(port_ns, name)(sam, ref, mask_list=mask, window_size=Nw, max_shift=ms)
    o.debug = True
    want = o.match(quiet=True)
    g = getattr(hip_ns, name)(sam, ref, mask_list=mask, window_size=Nw, max_shift=ms)
    g.debug = True
    got = g.match(quiet=True)
    assert g._lib.last_path(g._handle) in (1, 3)                    # the general kernel, by the library's own choice
    # pixels whose walk only ever saw costs above the noise floor: every known cell of the oracle's 5x5 memo
    # (a few dozen pixels meet a window without ANY valid pair: cost 0/0 = NaN.  The reference itself never returns from some
    #  of those -- its centre steps between two memoised cells for ever; the GPU walk and the oracle end such a walk after
    #  UMPA_MOVE_CAP moves without a cost call, umpa_walk.h -- so this test also pins that nothing hangs)
    d = want["debug_d"]
    has_nan = np.isnan(d).any(axis=-1)
    low = np.where(d >= 0, d, np.inf).min(axis=-1)
    solid = (low > 1e-12) & (want["err"] == 1) & ~has_nan
    noise = (low < 1e-20) & ~has_nan
    assert solid.sum() > 400 and noise.sum() > 100 and has_nan.sum() > 10      # the case shows all three kinds
    # "solid" is judged on the walk's LAST 5x5 neighbourhood; a walk may have crossed noise cells before that, so a handful of
    # the solid pixels still follow the noise: at most 1 % of them may differ in their walk, the others to the full bar
    def same_walk(res, tag):
        same = (res["err"] == want["err"]) & (res["debug_Ncalls"] == want["debug_Ncalls"])
        assert (~same & solid).sum() <= 0.01 * solid.sum(), "%s: %d of %d solid pixels walk differently" % (tag, (~same & solid).sum(), solid.sum())
        np.testing.assert_allclose(res["T"][solid & same], want["T"][solid & same], rtol=1e-5)
        return same
    same_walk(got, "general kernel")
    g._force = _lib.F_FORCE_TILED
    tiled = g.match(quiet=True)
    assert g._lib.last_path(g._handle) == 2
    same_t = same_walk(tiled, "table-based path")
    differs = ~same_t
    print("exact-fit windows (%s): %d noise pixels, walk length differs on %d of them on the table-based path, on %d on the general kernel"
          % (name, int(noise.sum()), int((differs & noise).sum()), int(((got["debug_Ncalls"] != want["debug_Ncalls"]) & noise).sum())))
