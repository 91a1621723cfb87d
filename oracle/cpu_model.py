"""
oracle/cpu_model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python faces of the two CPU checkers, with the same class API as ``umpa_amd.model``:

  * ``port``  -> oracle/libumpa_oracle.so   (this repo's plain-C restatement, umpa_oracle.c)
  * ``ref``   -> oracle/_ref/libumpa_ref.so (the reference C++ core compiled in place)

They reuse the host-side marshalling of ``umpa_amd.model`` (ROI arithmetic, window, result
packing) by overriding only the native-backend hook.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this module; nothing under umpa_amd/ does.
"""
import os
import subprocess

from umpa_amd import _lib, model

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(_HERE, "libumpa_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libumpa_ref.so")

_cache = {}


def build(quiet=True):
    """(Re)build the checkers with oracle/Makefile (gcc only; the _ref target needs /root/reference)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def native(which):
    if which not in _cache:
        path, prefix = {"port": (PORT_PATH, "umpaor_"), "ref": (REF_PATH, "umparef_")}[which]
        if which == "port" and not os.path.exists(path):
            build()
        _cache[which] = _lib.Native(path, prefix, False)
    return _cache[which]


def have_ref():
    return os.path.exists(REF_PATH)


def _classes(which):
    def _native(self):
        return native(which)

    ns = {}
    for base in (model.UMPAModelNoDF, model.UMPAModelDF, model.UMPAModelDFKernel):
        ns[base.__name__] = type(base.__name__ + "_" + which, (base,), {"_native": _native})
    return ns


port = type("port", (), _classes("port"))
ref = type("ref", (), _classes("ref"))
