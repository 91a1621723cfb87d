import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """UMPA_RECORD_UNCONVERGED=1: dump the unconverged-Newton counts every assert_parity call of this run saw
    (label -> count) under gpurun_out/, to be merged into tests/golden/unconverged_observed.json."""
    if not os.environ.get("UMPA_RECORD_UNCONVERGED"):
        return
    from oracle import parity
    seen = {}
    for label, ok, inside, unc in parity.OBSERVED:
        seen[label] = max(seen.get(label, 0), unc)
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    tag = "gpu" if "gpu" in (session.config.getoption("-m") or "") and "not gpu" not in (session.config.getoption("-m") or "") else "cpu"
    json.dump(seen, open(os.path.join(out, "unconverged_%s.json" % tag), "w"), indent=0, sort_keys=True)


# ----------------------------------------------------------------------------- golden cases

class Case:
    """One tests/golden/<name>.npz: inputs + the reference's outputs for a list of variants."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        self.variants = self.meta["variants"]
        self.Nw, self.max_shift, self.pos = self.meta["Nw"], self.meta["max_shift"], self.meta["pos"]

        def frames(key):
            if self.meta["ragged"]:
                n = 0
                out = []
                while "%s_%d" % (key, n) in z:
                    out.append(np.ascontiguousarray(z["%s_%d" % (key, n)], dtype=np.float64))
                    n += 1
                return out or None
            if key not in z:
                return None
            return np.ascontiguousarray(z[key], dtype=np.float64)

        self.sam, self.ref, self.mask = frames("sam"), frames("ref"), frames("mask")
        gen = self.meta.get("generator")
        if gen:                                                     # inputs stored as the parameters of a generator
            from umpa_amd import synth
            kw = {k: v for k, v in gen.items() if k != "func"}
            self.sam, self.ref = getattr(synth, gen["func"])(**kw)
        self.z = z

    def expected(self, n):
        pre = "v%d_" % n
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}

    def run(self, ns, n, debug=True):
        """Run variant ``n`` with the model classes found in namespace ``ns``
        (umpa_amd.model, oracle.cpu_model.port, oracle.cpu_model.ref)."""
        v = self.variants[n]
        cls = getattr(ns, v["model"])
        kw = dict(window_size=self.Nw, max_shift=self.max_shift)
        if self.mask is not None:
            kw["mask_list"] = self.mask
        if self.pos is not None:
            kw["pos_list"] = [np.array(p) for p in self.pos]
        m = cls(self.sam, self.ref, **kw)
        m.debug = debug
        m.assign_coordinates = v.get("assign", "sam")
        m.sub_pixel_mode = v.get("subpx", -1)
        if "Nw_set" in v:
            m.Nw = v["Nw_set"]
        mk = dict(quiet=True)
        if "dxdy" in v:
            mk["dxdy"] = tuple(v["dxdy"])
        if "step" in v:
            mk["step"] = v["step"]
        if "ROI" in v:
            roi = v["ROI"]
            mk["ROI"] = tuple(slice(*r) for r in roi) if v.get("ROI_kind") == "slice" else tuple(tuple(r) for r in roi)
        if v["model"] == "UMPAModelDFKernel":
            s0, s1 = m._convert_ROI_slice(mk.get("ROI"), mk.get("step"))
            sh = (1 + (s0[1] - s0[0] - 1) // s0[2], 1 + (s1[1] - s1[0] - 1) // s1[2])
            abc = np.zeros(sh + (3,))
            abc[..., 0], abc[..., 1], abc[..., 2] = v["abc"]
            if v.get("abc_ramp"):
                abc[..., 0] += np.linspace(0, 0.2, sh[1])[None, :]
            mk["abc"] = abc
        out = m.match(**mk)
        if v.get("coverage"):
            out["coverage"] = m.coverage()
        return out, m


ALL_CASES = ["A_small", "B_walks", "C_mask", "C_mask_ones", "D_stepping", "E_dfkernel",
             "F7_C1_ms2", "F7_C1_ms4", "F8_C2_crop", "F8_C3_crop", "H_cap"]


# ----------------------------------------------------------------------------- the parity bar
# (lives in oracle/parity.py so that bench.py's cpu_baseline leg and __graft_entry__.smoke() apply the same rules)
from oracle.parity import RTOL, assert_parity, newton_unconverged  # noqa: E402,F401


@pytest.fixture(scope="session")
def port_ns():
    from oracle import cpu_model
    cpu_model.native("port")
    return cpu_model.port


@pytest.fixture(scope="session")
def ref_ns():
    from oracle import cpu_model
    if not cpu_model.have_ref():
        pytest.skip("oracle/_ref/libumpa_ref.so not built (needs /root/reference)")
    return cpu_model.ref
