"""
ctypes binding of the C ABI declared in ``include/umpa_hip.h``.

``Native`` wraps one shared library exporting that ABI under a symbol prefix.  The
product uses exactly one instance: ``libumpa_hip.so`` (prefix ``umpa_hip_``), the HIP
library built from ``umpa_amd/csrc``.  There is no CPU fallback: if the library is
missing, or no HIP device is present, loading / model creation raises.
(The CPU checkers under ``oracle/`` export the same call shapes under other prefixes;
only the test-suite binds those, through ``oracle/cpu_model.py``.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "libumpa_hip.so")

ST_OK, ST_BOUND, ST_DIM, ST_POSITIVE = 1, 2, 4, 8
F_DEVICE_FRAMES = 1
F_DEVICE_IO, F_FORCE_DIRECT, F_FORCE_TILED, F_PLANAR, F_REUSE_REF_MAPS, F_USE_STAGED, F_ASYNC, F_FORCE_PLAIN_DIRECT = 1, 2, 4, 8, 16, 32, 64, 128

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_dpp = C.POINTER(_dp)

# every symbol include/umpa_hip.h declares (checked by tests/test_cabi_symbols.py)
HIP_SYMBOLS = [
    "device_count", "last_error", "version", "create", "destroy", "set_window", "set_subpx",
    "set_reference_shift", "coverage", "coverage_region", "cost", "min", "match_region",
    "spmin", "spmin_quad", "timing_enable", "timing_collect", "timing_read", "timing_fma", "last_path", "host_alloc", "host_free", "host_trim", "stage_sample", "wait", "set_rows_callback", "host_register", "host_unregister",
    "update_frames", "correct_bad_pixels", "last_stats",
]


class NativeError(RuntimeError):
    pass


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


class Native:
    """One loaded library + prefix."""

    def __init__(self, path, prefix, is_hip):
        if not os.path.exists(path):
            raise NativeError(
                "native library %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % path)
        self.path, self.prefix, self.is_hip = path, prefix, is_hip
        self.lib = C.CDLL(path)
        f = self._f
        create_args = [C.c_int, C.c_int, _ip, _dpp, _dpp, _dpp, _ip, C.c_int, _dp, C.c_int, C.c_int]
        if is_hip:
            create_args += [C.c_int, C.c_int]
        f("create", C.c_void_p, create_args)
        f("destroy", None, [C.c_void_p])
        f("set_window", C.c_int if is_hip else None, [C.c_void_p, _dp, C.c_int])
        f("set_subpx", C.c_int if is_hip else None, [C.c_void_p, C.c_int])
        f("set_reference_shift", C.c_int if is_hip else None, [C.c_void_p, C.c_int])
        f("coverage", C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int])
        f("cost", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp])
        f("min", C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _ip])
        mr = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        mr += [C.c_int, C.c_void_p] if is_hip else [C.c_int]
        f("match_region", C.c_int if is_hip else None, mr)
        if is_hip:
            f("device_count", C.c_int, [])
            f("last_error", C.c_char_p, [])
            f("version", C.c_char_p, [])
            f("coverage_region", C.c_int, [C.c_void_p] + [C.c_int] * 6 + [_dp])
            f("spmin", C.c_int, [C.c_int, _dp, _dp, _dp])
            f("spmin_quad", C.c_int, [C.c_int, _dp, _dp, _dp])
            f("timing_enable", C.c_int, [C.c_void_p, C.c_int])
            f("timing_collect", C.c_int, [C.c_void_p])
            f("timing_read", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), _dp, _ip])
            f("timing_fma", C.c_int, [C.c_void_p, C.c_int, _dp])
            f("host_alloc", C.c_void_p, [C.c_size_t])
            f("host_free", None, [C.c_void_p])
            f("host_trim", None, [])
            f("stage_sample", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
            f("wait", C.c_int, [C.c_void_p])
            f("set_rows_callback", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
            f("host_register", C.c_int, [C.c_void_p, C.c_size_t])
            f("host_unregister", C.c_int, [C.c_void_p])
            f("last_path", C.c_int, [C.c_void_p])
            f("last_stats", C.c_int, [C.c_void_p, _dp])
            f("update_frames", C.c_int, [C.c_void_p, _dpp, _dpp])
            f("correct_bad_pixels", C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p])
        else:
            f("spmin", C.c_double, [_dp, _dp])
            f("spmin_quad", C.c_double, [_dp, _dp])
            f("max_threads", C.c_int, [])

    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype, fn.argtypes = restype, argtypes
        setattr(self, name, fn)

    def error(self):
        if self.is_hip:
            return (self.last_error() or b"").decode()
        return ""

    def check(self, rc, what):
        if rc is not None and rc < 0:
            raise NativeError("%s failed (%d): %s" % (what, rc, self.error()))
        return rc


_hip = None


def _pin_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64; if both that copy
    and /opt/rocm's get loaded, the second one sees no device and streams / device pointers cannot be
    shared.  When torch is installed, load ITS runtime first (by path, so the order of `import torch`
    and of this call does not matter): libumpa_hip.so's DT_NEEDED libamdhip64.so.7 then resolves to
    the already-loaded object.  A host without torch (C, C++, cgo ...) simply gets /opt/rocm's."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def hip():
    """The product library.  Raises if it is not built; never substitutes anything else."""
    global _hip
    if _hip is None:
        _pin_hip_runtime()
        _hip = Native(HIP_LIB_PATH, "umpa_hip_", True)
    return _hip


ROWS_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)       # umpa_hip_rows_fn


def pinned_empty(shape, dtype, zero=False):
    """A numpy array in page-locked host memory from the library's pool (``umpa_hip_host_alloc``): the result maps of
    the host-array API are downloaded into such arrays at PCIe rate.  The block goes back to the pool when the last
    view of the array is gone.  Falls back to an ordinary array if pinning fails (the download is then staged)."""
    import weakref
    import numpy as np
    lib = hip()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = lib.host_alloc(max(n, 1))
    if not ptr:
        return (np.zeros if zero else np.empty)(shape, dtype=dt)
    buf = (C.c_char * max(n, 1)).from_address(ptr)
    weakref.finalize(buf, lib.host_free, ptr)
    a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
    if zero:
        a[...] = 0
    return a


class FrameSet:
    """Pointer tables for a list of frames (host ndarrays or CUDA/HIP torch tensors)."""

    def __init__(self, frames):
        self.keep = frames
        n = len(frames)
        self.table = (_dp * n)()
        for k, a in enumerate(frames):
            if hasattr(a, "data_ptr"):
                self.table[k] = C.cast(C.c_void_p(a.data_ptr()), _dp)
            else:
                self.table[k] = a.ctypes.data_as(_dp)

