"""
tests/golden/I_unstable.npz: the evidence behind the one allowance of the parity bar (oracle/parity.py).

Each crop is a small input stack whose output pixel (1, 1) two builds of the UNMODIFIED reference (its own flags
-O3 -ffast-math, and -O2 without -ffast-math; generator tests/golden/make_golden_unstable.py) answer differently by more
than 1e-5 px although err, Ncalls and the walk are bit-identical: the reference's unclamped Newton iteration
(Optim.cpp:41-130) does not survive rounding noise there.  The tests assert (CPU) that the stored outputs really differ,
that the checker's classifiers flag exactly such pixels, and that the oracle port reproduces the walk; (GPU) that the HIP
path, held to the full bar against either build, misses it only on pixels the checker classifies.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity

Z = np.load(os.path.join(GOLDEN, "I_unstable.npz"))
META = json.loads(str(Z["meta"]))
N = META["n"]


def _crop(n, tag):
    pre = "c%d_%s_" % (n, tag)
    return {k[len(pre):]: Z[k] for k in Z.files if k.startswith(pre)}


def test_two_reference_builds_disagree_on_these_pixels():
    assert N >= 6 and META["builds"]["fast"] != META["builds"]["strict"]
    worst = 0.0
    for n in range(N):
        a, b = _crop(n, "fast"), _crop(n, "strict")
        assert a["err"][1, 1] == 1 and b["err"][1, 1] == 1
        assert a["debug_Ncalls"][1, 1] == b["debug_Ncalls"][1, 1]
        # the 4x4 neighbourhoods the two fits start from agree to rounding: the disagreement is the iteration's
        np.testing.assert_allclose(a["debug_a"][1, 1], b["debug_a"][1, 1], rtol=1e-9, atol=1e-15)
        d = max(abs(a["dx"][1, 1] - b["dx"][1, 1]), abs(a["dy"][1, 1] - b["dy"][1, 1]))
        assert d > 1e-5, (n, d)
        worst = max(worst, d)
    assert worst > 1e-3          # (up to a few hundredths of a pixel)


def test_checker_classifies_them_and_the_port_walks_the_same_way(port_ns):
    from oracle import parity
    flagged = 0
    for n in range(N):
        a = _crop(n, "fast")
        da, dd = a["debug_a"][1, 1], a["debug_d"][1, 1]
        flagged += bool(parity.newton_unconverged(da, dd) or parity.newton_unstable(da, dd))
        m = port_ns.UMPAModelDF(Z["c%d_sam" % n], Z["c%d_ref" % n], window_size=META["Nw"], max_shift=META["max_shift"])
        got = m.match(quiet=True, num_threads=1)
        for ref in (a, _crop(n, "strict")):
            np.testing.assert_array_equal(got["err"], ref["err"])
            np.testing.assert_array_equal(got["debug_Ncalls"], ref["debug_Ncalls"])
            assert_parity(got, ref, META["max_shift"], "I_unstable c%d" % n, allow_illposed=1.0)   # every miss must be classified
    assert flagged == N, "%d of %d unstable pixels are flagged by newton_unconverged / newton_unstable" % (flagged, N)


@pytest.mark.gpu
def test_hip_on_the_unstable_crops():
    from umpa_amd import model
    for n in range(N):
        m = model.UMPAModelDF(Z["c%d_sam" % n], Z["c%d_ref" % n], window_size=META["Nw"], max_shift=META["max_shift"])
        got = m.match(quiet=True)
        for tag in ("fast", "strict"):
            assert_parity(got, _crop(n, tag), META["max_shift"], "I_unstable c%d %s" % (n, tag), allow_illposed=1.0)
