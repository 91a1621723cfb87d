"""
oracle/cpu_model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python faces of the two CPU checkers, with the same class API as ``umpa_amd.model``:

  * ``port``  -> oracle/libumpa_oracle.so   (this repo's plain-C restatement, umpa_oracle.c)
  * ``ref``   -> oracle/_ref/libumpa_ref.so (the reference C++ core compiled in place)

They reuse the argument marshalling and result packing of ``umpa_amd.model`` by overriding the
native-backend hook -- but NOT its geometry: extent, ROI / step / slice conversion and pixel counts
are computed here a second time, independently (``_OwnGeometry``: Python ``range`` objects instead of
the candidate's closed forms), following ``UMPA/model.pyx:531-623, 414-415`` (the window is
``numpy.hamming``'s on both sides, as in the reference itself, ``model.pyx:692``).  A slip in the candidate's geometry therefore shows up
as a disagreement with the checker, not as a shared mistake (tests/test_host_logic.py compares the two
on random ROI forms; the golden variants pin both).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; nothing under umpa_amd/ does.
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


class _OwnGeometry:
    """The checker's own host geometry (model.pyx:531-623, 414-415), written without looking at umpa_amd.model's."""

    def _calculate_extent(self):                                    # model.pyx:531-549: max over frames of pos + shape, minus 2 padding
        top0 = top1 = 0
        for (p0, p1), (h, w) in zip(self._pos_list, self._shape_list):
            top0 = max(top0, int(p0) + int(h))
            top1 = max(top1, int(p1) + int(w))
        return top0 - 2 * int(self._padding), top1 - 2 * int(self._padding)

    @staticmethod
    def _as_triple(x, n):
        if isinstance(x, slice):
            r = range(n)[x]                                         # what the slice selects of 0 .. n-1
            return (r.start, r.stop, r.step)
        a, b, c = x
        return (int(a), int(b), int(c))

    def _convert_ROI_slice(self, ROI=None, step=None):              # model.pyx:551-588
        n0, n1 = self._calculate_extent()
        if ROI is not None:
            if step is not None:
                raise RuntimeError('Step and ROI should not be specified simultaneously.')
            return self._as_triple(ROI[0], n0), self._as_triple(ROI[1], n1)
        (a0, b0, c0), (a1, b1, c1) = self._ROI
        if step is None:
            return (int(a0), int(b0), int(c0)), (int(a1), int(b1), int(c1))
        return self._as_triple(slice(a0, b0, step), n0), self._as_triple(slice(a1, b1, step), n1)

    def _set_ROI(self, ROI=None):                                   # model.pyx:590-616
        n0, n1 = self._calculate_extent()
        self._ROI = ((0, n0, 1), (0, n1, 1)) if ROI is None else (self._as_triple(ROI[0], n0), self._as_triple(ROI[1], n1))

    @staticmethod
    def _counts(s0, s1):                                            # model.pyx:414-415: how many pixels start:stop:step visits
        if s0[2] < 1 or s1[2] < 1:
            raise RuntimeError('ROI steps must be positive.')
        return len(range(s0[0], s0[1], s0[2])), len(range(s1[0], s1[1], s1[2]))


def _classes(which):
    def _native(self):
        return native(which)

    ns = {}
    for base in (model.UMPAModelNoDF, model.UMPAModelDF, model.UMPAModelDFKernel):
        ns[base.__name__] = type(base.__name__ + "_" + which, (_OwnGeometry, base), {"_native": _native})
    return ns


port = type("port", (), _classes("port"))
ref = type("ref", (), _classes("ref"))
