"""The reference-side binding of INTEGRATION.md section B must COMPILE: the `.pxd` printed there (what would replace
UMPA/Model.pxd:30-68 for the HIP backend) is cut out of the document, cythonised together with a small caller
(examples/cython_binding/hip_backend.pyx: the roles of model.pyx:764-770, 476-492, 305-309) against include/umpa_hip.h, linked
with libumpa_hip.so and imported.  Without a GPU the constructor fails with the library's own "no HIP device" error -- which
proves the call went through the C ABI; with one (`-m gpu`) the Cython caller's maps equal the ctypes binding's bit for bit."""
import os
import re
import subprocess
import sys
import sysconfig

import numpy as np
import pytest

from conftest import REPO


def _build(tmp):
    import __graft_entry__ as g
    if not os.path.exists(g.HIP_LIB):
        g.build()
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cython\n(.*?)```", doc, flags=re.S)
    pxd = [b for b in blocks if "cdef extern from \"umpa_hip.h\"" in b]
    assert len(pxd) == 1, "INTEGRATION.md section B holds exactly one .pxd"
    open(os.path.join(tmp, "HipModel.pxd"), "w").write(pxd[0])
    src = open(os.path.join(REPO, "examples", "cython_binding", "hip_backend.pyx")).read()
    open(os.path.join(tmp, "hip_backend.pyx"), "w").write(src)
    subprocess.run([sys.executable, "-m", "cython", "-3", "hip_backend.pyx", "-o", "hip_backend.c"], cwd=tmp, check=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    libdir = os.path.dirname(g.HIP_LIB)
    cmd = ["gcc", "-O1", "-fPIC", "-shared", "-Wno-deprecated-declarations", "hip_backend.c", "-o", "hip_backend" + ext,
           "-I" + os.path.join(REPO, "include"), "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(),
           "-L" + libdir, "-l:libumpa_hip.so", "-Wl,-rpath," + libdir]
    subprocess.run(cmd, cwd=tmp, check=True)
    return tmp


_RUN = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from umpa_amd import _lib
_lib.hip()                                   # one HIP runtime per process: the library's loader goes first (DESIGN.md 1)
import hip_backend
from umpa_amd.synth import make_stack
sam, ref, _ = make_stack(72, 96, 4, 4, df=True, seed=3, amplitude=2.0)
try:
    m = hip_backend.HipModelDF(sam, ref, 3, 4)
except RuntimeError as e:
    print("RuntimeError:", e); sys.exit(0 if "no HIP device" in str(e) or "device" in str(e).lower() else 3)
got = m.match()
from umpa_amd import model
want = model.UMPAModelDF(sam, ref, window_size=3, max_shift=4).match(quiet=True)
for k in ("f", "T", "dx", "dy", "df", "err"):
    assert np.array_equal(got[k], want[k]), k
print("cython == ctypes on", got["err"].size, "pixels,", int(got["err"].sum()), "ok")
"""


def test_reference_side_cython_binding_compiles_and_links(tmp_path):
    tmp = _build(str(tmp_path))
    out = subprocess.run([sys.executable, "-c", _RUN % (tmp, REPO)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "RuntimeError:" in out.stdout or "cython == ctypes" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_reference_side_cython_binding_matches_the_ctypes_binding(tmp_path):
    tmp = _build(str(tmp_path))
    out = subprocess.run([sys.executable, "-c", _RUN % (tmp, REPO)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "cython == ctypes" in out.stdout, out.stdout + out.stderr
