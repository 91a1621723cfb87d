"""
The oracle against the golden vectors frozen from the unmodified reference
(tests/golden/make_golden.py).  CPU only.  `port` = oracle/umpa_oracle.c (this repo's
restatement), `ref` = oracle/_ref/libumpa_ref.so (the reference C++ core compiled in place,
present only where /root/reference was available at build time).
"""
import os

import numpy as np
import pytest

from conftest import ALL_CASES, GOLDEN, Case, assert_parity


def _variants():
    out = []
    for name in ALL_CASES:
        for n in range(len(Case(name).variants)):
            out.append((name, n))
    return out


@pytest.mark.parametrize("name,n", _variants())
def test_port_matches_reference_golden(port_ns, name, n):
    case = Case(name)
    got, _ = case.run(port_ns, n)
    want = case.expected(n)
    stats = assert_parity(got, want, case.max_shift, "%s v%d" % (name, n), subpx=case.variants[n].get("subpx", -1))
    # the restatement is fp64 and follows the same evaluation order: it must be far inside the bar
    ok = want["err"] == 1
    for k in ("T", "df"):
        if k in want and ok.any():
            assert np.max(np.abs(got[k] - want[k])[ok]) < 1e-10
    if "debug_d" in want:
        np.testing.assert_allclose(got["debug_d"], want["debug_d"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(got["debug_a"] * ok[..., None], want["debug_a"], rtol=1e-9, atol=1e-13)
    if "coverage" in want:
        np.testing.assert_array_equal(got["coverage"], want["coverage"])
    assert stats["ok"] == int(ok.sum())


@pytest.mark.parametrize("name", ["A_small", "B_walks", "C_mask", "D_stepping", "E_dfkernel", "F8_C2_crop"])
def test_ref_shim_matches_reference_golden(ref_ns, name):
    case = Case(name)
    for n in range(len(case.variants)):
        got, _ = case.run(ref_ns, n)
        assert_parity(got, case.expected(n), case.max_shift, "%s v%d" % (name, n), f_on_failed=False)


def test_subpixel_kats(port_ns, ref_ns):
    """F1: spmin / spmin_quad known answers (reference model.spmq / model.spm, model.pyx:31-80)."""
    import ctypes as C
    from oracle import cpu_model
    z = np.load(os.path.join(GOLDEN, "F1_subpixel.npz"))
    dp = C.POINTER(C.c_double)
    for which in ("port", "ref"):
        lib = cpu_model.native(which)
        for n in range(len(z["a"])):
            a = np.ascontiguousarray(z["a"][n])
            pos = np.zeros(2)
            v = lib.spmin(a.ctypes.data_as(dp), pos.ctypes.data_as(dp))
            np.testing.assert_allclose(pos, z["spmin_pos"][n], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(v, z["spmin_val"][n], rtol=1e-9, atol=1e-15)
            pos = np.zeros(2)
            v = lib.spmin_quad(a.ctypes.data_as(dp), pos.ctypes.data_as(dp))
            np.testing.assert_allclose(pos, z["quad_pos"][n], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(v, z["quad_val"][n], rtol=1e-9, atol=1e-15)


def test_cost_kats(port_ns):
    """F2: single cost() evaluations, including |shift| == max_shift (bound error)."""
    z = np.load(os.path.join(GOLDEN, "F2_cost.npz"))
    case = Case("A_small")
    for mdl in ("UMPAModelNoDF", "UMPAModelDF"):
        for assign in ("sam", "ref"):
            m = getattr(port_ns, mdl)(case.sam, case.ref, window_size=2, max_shift=4)
            m.assign_coordinates = assign
            want = z["%s_%s" % (mdl, assign)]
            for n, (i, j, si, sj) in enumerate(z["pts"]):
                if abs(si) >= 4 or abs(sj) >= 4:
                    lib = m._lib
                    vals = np.zeros(3)
                    import ctypes as C
                    st = lib.cost(m._handle, int(i), int(j), int(si), int(sj), vals.ctypes.data_as(C.POINTER(C.c_double)))
                    assert st & 2 and not st & 1          # bound_error, not ok
                else:
                    got = m.cost(int(i), int(j), float(si), float(sj))
                    np.testing.assert_allclose(got, want[n], rtol=1e-10, atol=1e-16)


def test_quad_table_is_pinv():
    """The integer table of spmin_quad (Optim.cpp:169-174) is 400*pinv of the paraboloid design matrix."""
    g = np.array([-1.0, 0, 1, 2])
    I, J = np.meshgrid(g, g, indexing="ij")
    i, j = I.ravel(), J.ravel()
    A = np.stack([np.ones(16), i, j, i * i, i * j, j * j], 1)
    P = np.linalg.pinv(A) * 400
    assert np.allclose(P, np.round(P), atol=1e-9)
    assert np.allclose(P[3], 25 * (i * i - i - 1)) and np.allclose(P[5], 25 * (j * j - j - 1))
    assert np.allclose(P[4], 4 * (2 * i - 1) * (2 * j - 1))
    assert np.allclose(P[2].reshape(4, 4), P[1].reshape(4, 4).T)
