"""The C-ABI library loads on a CPU-only box and exports every symbol include/umpa_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared():
    hdr = open(os.path.join(REPO, "include", "umpa_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(umpa_hip_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    from umpa_amd import _lib
    assert _declared() == sorted("umpa_hip_" + s for s in _lib.HIP_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    if not os.path.exists(g.HIP_LIB):
        g.build()
    lib = ctypes.CDLL(g.HIP_LIB)
    for name in _declared():
        assert hasattr(lib, name), name
    from umpa_amd import _lib
    h = _lib.hip()
    assert b"gfx950" in h.version()
    assert h.device_count() >= 0


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product fails loudly; it never computes on the CPU."""
    import numpy as np
    from umpa_amd import _lib, model
    if _lib.hip().device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.ones((3, 40, 40))
    with pytest.raises(RuntimeError, match="no HIP device"):
        model.UMPAModelDF(a, a, window_size=2, max_shift=3)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "umpa_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "umpaor_" not in txt and "umparef_" not in txt, f


def test_reference_import_name_alias():
    """`import UMPA` / `from UMPA import model` / `UMPA.match` resolve to umpa_amd: callers written against the reference
    (speckle_matching.py:9, umpa_multi.py:149) keep their import lines.  The alias holds no code of its own."""
    import UMPA
    from UMPA import align, model
    from UMPA.model import UMPAModelDF
    import umpa_amd
    assert UMPA.match is umpa_amd.match and UMPA.match_unbiased is umpa_amd.match_unbiased
    assert model.UMPAModelNoDF is umpa_amd.model.UMPAModelNoDF and UMPAModelDF is umpa_amd.UMPAModelDF
    assert align.UMPA_nobias is umpa_amd.align.UMPA_nobias
    for f in os.listdir(os.path.join(REPO, "UMPA")):
        if f.endswith(".py"):
            txt = open(os.path.join(REPO, "UMPA", f)).read()
            assert "def " not in txt and "class " not in txt and "oracle" not in txt, f
