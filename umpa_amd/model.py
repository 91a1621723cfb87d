"""
Host-side mirror of the reference's model API on top of the HIP library.

Classes ``UMPAModelNoDF`` / ``UMPAModelDF`` / ``UMPAModelDFKernel`` keep the names,
constructor arguments, methods, properties, result dictionaries and error behaviour of
the reference extension types (reference ``UMPA/model.pyx:116-997``), so code written
against ``UMPA.model`` runs unchanged.  All numerics happen in ``libumpa_hip.so``
(``include/umpa_hip.h``); this module only does what the Cython layer did around the C++
core: pointer marshalling, padding / extent / ROI arithmetic, the Hamming window and
result packing.

Differences from the reference, all deliberate:
 * frames that are not float64 are converted to float64 copies (the reference casts to a
   temporary and then reads freed memory);
 * ``num_threads`` is accepted and ignored (the pixel loop runs on the GPU);
 * ``ROI`` tuples and a grown ``Nw`` are range-checked (the reference reads out of bounds);
 * ``debug_a`` of a pixel whose walk ended before the 4x4 gather is zero (the reference
   leaves whatever the same OpenMP thread computed for an earlier pixel).
 * extra keyword ``device`` (HIP device index, default: ``UMPA_HIP_DEVICE`` or 0) and
   attribute ``debug`` (default True, as the reference is compiled with DEBUG=True,
   ``model.pyx:26``): set False to skip the three ``debug_*`` outputs (332 B/pixel).
"""
import os

import numpy as np

from . import _lib

__all__ = ["UMPAModelBase", "UMPAModelNoDF", "UMPAModelDF", "UMPAModelDFKernel",
           "spm", "spmq"]

NPDOUBLE = np.dtype("float64")

KIND_NODF, KIND_DF, KIND_DFKERNEL = 0, 1, 2


def _cround(x):
    """libc round(): halves away from zero (the reference's `from libc.math cimport round`, model.pyx:20);
    Python's round() goes to the even neighbour."""
    x = float(x)
    return int(np.sign(x) * np.floor(abs(x) + 0.5))


def _default_device():
    if "UMPA_HIP_DEVICE" in os.environ:
        return int(os.environ["UMPA_HIP_DEVICE"])
    if "LOCAL_RANK" in os.environ:            # one process per GPU under torch.distributed.run
        n = _lib.hip().device_count()
        return int(os.environ["LOCAL_RANK"]) % max(n, 1)
    return 0


def _spfit(a, quad):
    """reference ``model.spm`` / ``model.spmq`` (``model.pyx:31-80``) evaluated on the device."""
    if a.shape != (4, 4):
        raise RuntimeError("input array must be (4,4)")
    if a.dtype not in (np.dtype("float64"), np.dtype("float32")):
        raise RuntimeError("Unsupported data type")
    a64 = np.ascontiguousarray(a, dtype=np.float64)
    pos = np.zeros(2)
    val = np.zeros(1)
    lib = _lib.hip()
    fn = lib.spmin_quad if quad else lib.spmin
    lib.check(fn(_default_device(), _lib._ptr(a64, _lib._dp), _lib._ptr(pos, _lib._dp),
                 _lib._ptr(val, _lib._dp)), "spmin")
    if a.dtype == np.dtype("float32"):
        return pos.astype(np.float32), np.float32(val[0])
    return pos, float(val[0])


def spm(a):
    """Sub-pixel minimum of a 4x4 array by quadratic fit (``model.pyx:31-54`` -> ``spmin_quad``)."""
    return _spfit(np.asarray(a), True)


def spmq(a):
    """Sub-pixel minimum of a 4x4 array by the B-spline model (``model.pyx:57-80`` -> ``spmin``)."""
    return _spfit(np.asarray(a), False)


class UMPAModelBase:
    """Mirror of ``UMPAModelBase`` (``model.pyx:116-755``)."""

    Nparam = 0
    safe_crop = 0
    _kind = None
    debug = True

    # -- backend hook (the test-suite's CPU checkers override this; the product never does)
    def _native(self):
        return _lib.hip()

    def __init__(self, sam_list, ref_list, mask_list=None, pos_list=None,
                 window_size=2, max_shift=4, ROI=None, device=None):
        if self._kind is None:
            raise NotImplementedError(
                'UMPAModelBase is not supposed to be called directly, use one of '
                'the subclasses "UMPAModelNoDF", "UMPAModelDF", or "UMPAModelDFKernel".')
        Nw = int(window_size)
        Na = len(sam_list)
        self._handle = None
        self._lib = self._native()
        self._device = _default_device() if device is None else int(device)

        # frames and shapes (model.pyx:225-254)
        self._sam = self._prepare_frames(sam_list)
        self._shape_list = [np.array(s.shape, dtype=np.int32) for s in self._sam]
        self._sam_list = sam_list
        self._ref = self._prepare_frames(ref_list)
        for k, r in enumerate(self._ref):
            samsh = tuple(int(x) for x in self._shape_list[k])
            if samsh != tuple(r.shape):
                raise RuntimeError('Incompatible shape between sample {0} and '
                                   'reference frames {1} (entry [{2}] in the '
                                   'datasets).'.format(samsh, tuple(r.shape), k))
        self._ref_list = ref_list
        self._mask = None
        if mask_list is not None:                                   # model.pyx:257-262
            self._mask = self._prepare_frames(mask_list)
            if len(self._mask) != Na or any(tuple(m.shape) != tuple(s.shape)
                                            for m, s in zip(self._mask, self._sam)):
                raise RuntimeError('mask_list must match sam_list in length and frame shapes.')
        self._mask_list = mask_list

        # positions (model.pyx:265-283)
        if pos_list is None:
            pos_list = [np.zeros((2,), dtype=np.int32) for _ in range(Na)]
        else:
            pos_list = [np.asarray(p).astype(np.int32) for p in pos_list]
            if len(pos_list) != Na:
                raise RuntimeError(
                    'Unexpected length for position list (len(pos_list)={0}, '
                    'len(sam_list)={1})'.format(len(pos_list), Na))
        if np.any(np.array(pos_list) < 0):
            raise RuntimeError('Negative frame positions (entries in pos_list) are not allowed.')
        pmin = np.min(pos_list, axis=0)
        if not np.all(pmin == 0):
            raise RuntimeError('Positions should start at 0.')
        self._pos_list = pos_list

        self._max_shift = int(max_shift)
        self._padding = self._max_shift + Nw + self.safe_crop      # model.pyx:286
        self._Nw = Nw
        self._subpx = -1
        self._refshift = 0
        win = self._make_window(Nw)

        dims = np.ascontiguousarray(np.array(self._shape_list, dtype=np.int32).reshape(Na, 2))
        pos = np.ascontiguousarray(np.array(pos_list, dtype=np.int32).reshape(Na, 2))
        self._fs = (_lib.FrameSet(self._sam), _lib.FrameSet(self._ref),
                    _lib.FrameSet(self._mask) if self._mask is not None else None)
        on_device = hasattr(self._sam[0], "data_ptr")
        args = [self._kind, Na, _lib._ptr(dims, _lib._ip), self._fs[0].table, self._fs[1].table,
                self._fs[2].table if self._fs[2] is not None else None,
                _lib._ptr(pos, _lib._ip), Nw, _lib._ptr(win, _lib._dp), self._max_shift, self._padding]
        if self._lib.is_hip:
            args += [self._device, _lib.F_DEVICE_FRAMES if on_device else 0]
        self._handle = self._lib.create(*args)
        if not self._handle:
            raise RuntimeError("could not create the native model: %s" % self._lib.error())
        self._ROI = None
        self._set_ROI(ROI)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.destroy(h)

    def update_frames(self, sam_list=None, ref_list=None):
        """Swap in new sample and/or reference stacks of the same shapes without rebuilding the model
        (extension: the reference rebuilds a model per projection, ``umpa_multi.py:149``).  On the GPU
        the untouched stack stays resident."""
        new_sam = self._prepare_frames(sam_list) if sam_list is not None else None
        new_ref = self._prepare_frames(ref_list) if ref_list is not None else None
        for new in (new_sam, new_ref):
            if new is not None and [tuple(x.shape) for x in new] != [tuple(x.shape) for x in self._sam]:
                raise RuntimeError('update_frames needs stacks of the shapes the model was built with.')
        if not self._lib.is_hip or hasattr(self._sam[0], "data_ptr"):
            raise RuntimeError('update_frames needs a model that owns device copies of host frames.')
        fs_s = _lib.FrameSet(new_sam) if new_sam is not None else None
        fs_r = _lib.FrameSet(new_ref) if new_ref is not None else None
        self._lib.check(self._lib.update_frames(self._handle, fs_s.table if fs_s else None,
                                                fs_r.table if fs_r else None), "update_frames")
        if new_sam is not None:
            self._sam, self._sam_list = new_sam, sam_list
        if new_ref is not None:
            self._ref, self._ref_list = new_ref, ref_list
        self._fs = (_lib.FrameSet(self._sam), _lib.FrameSet(self._ref), self._fs[2])

    # -- input handling
    def _check_contiguous(self, a):
        if any(not (x.is_contiguous() if hasattr(x, "is_contiguous") else x.flags.c_contiguous) for x in a):
            raise RuntimeError('The provided image frames are not C-contiguous.')      # model.pyx:311-317

    def _prepare_frames(self, frames):
        frames = list(frames)
        if len(frames) and hasattr(frames[0], "data_ptr"):          # torch tensors already on the GPU
            for x in frames:
                if not x.is_cuda or str(x.dtype) != "torch.float64" or x.dim() != 2:
                    raise RuntimeError('device frames must be 2-D float64 CUDA/HIP tensors.')
            self._check_contiguous(frames)
            return frames
        frames = [np.asarray(x) for x in frames]
        self._check_contiguous(frames)
        out = []
        for x in frames:
            if x.ndim != 2:
                raise RuntimeError('frames must be two-dimensional.')
            out.append(x if x.dtype == NPDOUBLE else np.ascontiguousarray(x, dtype=NPDOUBLE))
        return out

    def _make_window(self, n):                                      # model.pyx:691-696
        window = np.multiply.outer(np.hamming(2 * n + 1), np.hamming(2 * n + 1))
        window /= window.sum()
        self._window = np.ascontiguousarray(window, dtype=NPDOUBLE)
        return self._window

    # -- single-pixel entry points
    def _min(self, i, j, values):
        uv = np.zeros(2)                                            # Model.cpp:316-321
        self._lib.check(self._lib.min(self._handle, int(i), int(j), _lib._ptr(values, _lib._dp),
                                      _lib._ptr(uv, _lib._dp), None, None, None), "min")

    def min(self, x, y):
        raise NotImplementedError

    # -- ROI / extent arithmetic (model.pyx:531-623)
    def _calculate_extent(self):
        padding = self._padding
        pmax = np.max(np.array(self._pos_list) + np.array(self._shape_list), axis=0)
        N0 = 1 + (pmax[0] - 2 * padding - 1)
        N1 = 1 + (pmax[1] - 2 * padding - 1)
        return int(N0), int(N1)

    def _convert_ROI_slice(self, ROI=None, step=None):
        N0, N1 = self._calculate_extent()
        if ROI is not None:
            if step is not None:
                raise RuntimeError('Step and ROI should not be specified simultaneously.')
            s0, s1 = ROI
            if type(s0) is slice:
                s0 = s0.indices(N0)
            if type(s1) is slice:
                s1 = s1.indices(N1)
        else:
            s0, s1 = self._ROI
            if step is not None:
                s0 = slice(s0[0], s0[1], step).indices(N0)
                s1 = slice(s1[0], s1[1], step).indices(N1)
        return tuple(int(v) for v in s0), tuple(int(v) for v in s1)

    def _set_ROI(self, ROI=None):
        N0, N1 = self._calculate_extent()
        if ROI is None:
            self._ROI = ((0, N0, 1), (0, N1, 1))
        else:
            s0, s1 = ROI
            if type(s0) is slice:
                s0 = s0.indices(N0)
            if type(s1) is slice:
                s1 = s1.indices(N1)
            self._ROI = (tuple(int(v) for v in s0), tuple(int(v) for v in s1))

    def set_step(self, step):
        self._set_ROI(ROI=self._convert_ROI_slice(step=step))
        return self._ROI

    def coords(self, ROI=None):
        offset = self.padding
        if ROI is not None:
            s0, s1 = self._convert_ROI_slice(ROI=ROI)
        else:
            s0, s1 = self._ROI
        return offset + np.arange(*s0), offset + np.arange(*s1)

    def test(self):                                                 # ModelBase::test, Model.cpp:226-234
        return float(len(self._sam))

    @staticmethod
    def _counts(s0, s1):
        start0, end0, step0 = s0
        start1, end1, step1 = s1
        if step0 < 1 or step1 < 1:
            raise RuntimeError('ROI steps must be positive.')
        N0 = 1 + (end0 - start0 - 1) // step0                       # model.pyx:414-415
        N1 = 1 + (end1 - start1 - 1) // step1
        return N0, N1

    def _check_range(self, s0, s1, N0, N1):
        E0, E1 = self._calculate_extent()
        if N0 < 1 or N1 < 1:
            raise RuntimeError('Empty ROI %s.' % ((s0, s1),))
        if s0[0] < 0 or s1[0] < 0 or s0[0] + s0[2] * (N0 - 1) >= E0 or s1[0] + s1[2] * (N1 - 1) >= E1:
            raise RuntimeError('ROI %s exceeds the reconstructible extent %s.' % ((s0, s1), (E0, E1)))

    # -- coverage (model.pyx:499-529)
    def _trivial_coverage(self):
        if self._mask is not None:
            return False
        sh0 = tuple(self._shape_list[0])
        return all(tuple(p) == (0, 0) for p in self._pos_list) and all(tuple(s) == sh0 for s in self._shape_list)

    def coverage(self, step=None, ROI=None):
        s0, s1 = self._convert_ROI_slice(ROI, step)
        N0, N1 = self._counts(s0, s1)
        self._check_range(s0, s1, N0, N1)
        cmap = np.zeros((N0, N1), dtype=NPDOUBLE)
        if self._lib.is_hip:
            self._lib.check(self._lib.coverage_region(self._handle, s0[0], s0[2], N0, s1[0], s1[2], N1,
                                                      _lib._ptr(cmap, _lib._dp)), "coverage")
        else:
            off = self._padding
            one = np.zeros(1)
            for xi in range(N0):
                for xj in range(N1):
                    self._lib.coverage(self._handle, _lib._ptr(one, _lib._dp),
                                       off + s0[0] + s0[2] * xi, off + s1[0] + s1[2] * xj)
                    cmap[xi, xj] = one[0]
        return cmap

    # -- the match loop (model.pyx:334-497)
    def _match(self, step=None, input_values=None, dxdy=None, ROI=None,
               num_threads=None, quiet=False):
        if (ROI is not None) and (step is not None):
            print("Warning: 'ROI' and 'step' parameters are set simultaneously. "
                  "'step' parameter is ignored.")
            step = None
        if not quiet:
            if self._lib.is_hip:
                print("Using HIP device %d" % self._device)
        s0, s1 = self._convert_ROI_slice(ROI, step)
        self._set_ROI((s0, s1))
        N0, N1 = self._counts(s0, s1)
        self._check_range(s0, s1, N0, N1)
        sh = (N0, N1)
        shp = (N0, N1, self.Nparam)
        planar = False

        if self._trivial_coverage():
            covermap = None                                         # == Na everywhere: nothing is skipped
            thr = 0.0
        else:
            covermap = self.coverage(ROI=(s0, s1))
            thr = .1 * covermap.max() / len(self._sam)              # model.pyx:431

        if input_values is not None:                                # model.pyx:442-455
            if input_values.shape != shp:
                raise RuntimeError("Input values have the wrong shape: "
                                   "%s, should be %s" % (input_values.shape, shp))
            if input_values.dtype != np.float64:
                raise RuntimeError("Input values have the wrong type: "
                                   "%s, should be %s" % (input_values.dtype, np.float64))
            values = np.ascontiguousarray(input_values)
        elif self._lib.is_hip:
            # one plane per map: the device writes the result maps directly (no host-side de-interleaving)
            planar = True
            shp = (self.Nparam, N0, N1)
            values = self._alloc(shp, NPDOUBLE, covermap is not None)
        else:
            values = np.zeros(shp, dtype=NPDOUBLE)

        uv = None
        if dxdy is not None:                                        # model.pyx:461-465
            uv = np.zeros((N0, N1, 2), dtype=NPDOUBLE)
            uv[:, :, 0] = dxdy[0]
            uv[:, :, 1] = dxdy[1]
        if self._lib.is_hip:                                        # page-locked: downloaded at PCIe rate, chunk by chunk
            err = self._alloc(sh, np.int32, covermap is not None)
        else:
            err = np.zeros(sh, dtype=np.int32)
        result = {}
        dd = da = dn = None
        if self.debug:                                              # debug = "ncalls": only the evaluation counts
            mk = (lambda shape, dt: self._alloc(shape, dt, covermap is not None)) if self._lib.is_hip else \
                 (lambda shape, dt: np.zeros(shape, dtype=dt))
            if self.debug != "ncalls":
                dd = mk(sh + (25,), NPDOUBLE)
                da = mk(sh + (16,), NPDOUBLE)
            dn = mk(sh, np.int32)

        vp = lambda a: a.ctypes.data if a is not None else None
        args = [self._handle, s0[0], s0[2], N0, s1[0], s1[2], N1, vp(values), self.Nparam, vp(uv), vp(err),
                vp(covermap), float(thr), vp(dd), vp(da), vp(dn)]
        if self._lib.is_hip:
            flags = self._match_flags() | (_lib.F_PLANAR if planar else 0)
            if not hasattr(self._sam[0], "data_ptr"):               # frames owned by the library: it knows when the
                flags |= _lib.F_REUSE_REF_MAPS                      # reference stack changed (update_frames, Nw)
            args += [flags, None]
        else:
            if uv is None:
                uv = np.zeros((N0, N1, 2), dtype=NPDOUBLE)
                args[9] = vp(uv)
            args += [int(num_threads) if num_threads else max(1, self._lib.max_threads())]
        self._lib.check(self._lib.match_region(*args), "match_region")

        result['values'] = values
        result['_planar'] = planar
        result['err'] = err
        if self.debug:
            if dd is not None:
                result['debug_d'] = dd
                result['debug_a'] = da
            result['debug_Ncalls'] = dn
        return result

    _force = 0
    _use_staged = False
    _async = False
    _out_alloc = None           # hook (shape, dtype, zero) -> array for the result maps (farm.py: shared-memory result slots)

    def _alloc(self, shape, dtype, zero):
        if self._out_alloc is not None:
            return self._out_alloc(shape, dtype, zero)
        return _lib.pinned_empty(shape, dtype, zero=zero)

    def match_async(self, **kw):
        """``match()`` that returns as soon as the kernels and the downloads are enqueued (HIP only); the maps of the
        returned dictionary are valid after ``wait()``.  Lets a caller upload the next projection meanwhile."""
        self._async = True
        try:
            return self.match(**kw)
        finally:
            self._async = False

    def wait(self):
        self._lib.check(self._lib.wait(self._handle), "wait")

    def _match_flags(self):
        f = self._force
        if self._async:
            f |= _lib.F_ASYNC
        if self._use_staged:                                        # one-shot: the staged stack is adopted by this match
            f |= _lib.F_USE_STAGED
            self._use_staged = False
        return f

    def stage_sample(self, raw_frames, dark=None, flat=None):
        """Upload the NEXT sample stack while the current match is still running (extension; the reference builds a
        new model per projection, ``umpa_multi.py:149``).  ``raw_frames``: host frames of the model's shapes, float64 /
        float32 / uint16 (page-locked arrays -- ``_lib.pinned_empty`` -- make this return at once); ``dark`` / ``flat``:
        device float64 stacks (lists of 2-D HIP tensors) for the fused ``(raw - dark) / flat`` of ``umpa_multi.py:144``.
        The staged stack becomes the sample stack of the next ``match()``."""
        if not self._lib.is_hip or hasattr(self._sam[0], "data_ptr"):
            raise RuntimeError('stage_sample needs a model that owns device copies of host frames.')
        raw = [np.asarray(x) for x in raw_frames]
        if [tuple(x.shape) for x in raw] != [tuple(x.shape) for x in self._sam]:
            raise RuntimeError('stage_sample needs a stack of the shapes the model was built with.')
        code = {np.dtype(np.float64): 0, np.dtype(np.float32): 1, np.dtype(np.uint16): 2}.get(raw[0].dtype)
        if code is None or any(x.dtype != raw[0].dtype for x in raw):
            raise RuntimeError('raw frames must all be float64, float32 or uint16.')
        self._check_contiguous(raw)
        fs = _lib.FrameSet(raw)
        fd = _lib.FrameSet(list(dark)) if dark is not None else None
        ff = _lib.FrameSet(list(flat)) if flat is not None else None
        self._lib.check(self._lib.stage_sample(self._handle, fs.table, code, fd.table if fd else None,
                                               ff.table if ff else None), "stage_sample")
        self._staged_keep = (raw, dark, flat)                       # alive until the upload has been consumed
        self._use_staged = True

    # -- properties (model.pyx:625-755)
    @property
    def extent(self):
        return self._calculate_extent()

    @property
    def ROI(self):
        return self._ROI

    @ROI.setter
    def ROI(self, new_ROI):
        self._set_ROI(new_ROI)

    @property
    def sh(self):
        s0, s1 = self._ROI
        return ((s0[1] - s0[0] - 1) // s0[2] + 1, (s1[1] - s1[0] - 1) // s1[2] + 1)

    @property
    def Na(self):
        return len(self._sam)

    @property
    def sam_list(self):
        return self._sam_list

    @property
    def ref_list(self):
        return self._ref_list

    @property
    def mask_list(self):
        return self._mask_list

    @property
    def shape_list(self):
        return self._shape_list

    @property
    def pos_list(self):
        return self._pos_list

    @property
    def window(self):
        return self._window

    @property
    def Nw(self):
        return self._Nw

    @Nw.setter
    def Nw(self, new_Nw):
        new_Nw = int(new_Nw)
        if new_Nw < 0:
            raise RuntimeError("Nw must be non-negative.")           # Model.cpp:242
        if new_Nw + self._max_shift + self.safe_crop > self._padding:
            raise RuntimeError("Nw=%d does not fit the padding %d fixed at construction "
                               "(the reference would read outside the frames)." % (new_Nw, self._padding))
        old = self._window
        win = self._make_window(new_Nw)
        rc = self._lib.set_window(self._handle, _lib._ptr(win, _lib._dp), new_Nw)
        if rc is not None and rc < 0:
            self._window = old
            raise RuntimeError(self._lib.error())
        self._Nw = new_Nw

    @property
    def max_shift(self):
        return self._max_shift

    @property
    def padding(self):
        return self._padding

    @property
    def assign_coordinates(self):
        return {0: 'sam', 1: 'ref'}[self._refshift]

    @assign_coordinates.setter
    def assign_coordinates(self, new_mode):
        opts = {'sam': 0, 'ref': 1}
        try:
            set_value = opts[new_mode]
        except (KeyError, TypeError):
            print('Option %s is not available, parameter was not changed.' % repr(new_mode))
        else:
            self._refshift = set_value
            self._lib.set_reference_shift(self._handle, set_value)

    @property
    def sub_pixel_mode(self):
        return self._subpx

    @sub_pixel_mode.setter
    def sub_pixel_mode(self, new_mode):
        self._subpx = int(new_mode)
        self._lib.set_subpx(self._handle, self._subpx)

    # -- packing shared by the subclasses (model.pyx:815-822, :881-889)
    def _unpack(self, result, with_df):
        values = result.pop('values')
        if result.pop('_planar', False):
            planes = values                                         # written plane by plane on the device
        else:
            # one pass over the interleaved array instead of one strided copy per map
            planes = np.ascontiguousarray(np.moveaxis(values[:, :, :5 if with_df else 4], 2, 0))
        result['f'], result['T'], result['dx'], result['dy'] = planes[0], planes[1], planes[2], planes[3]
        if with_df:
            result['df'] = planes[4]
        return result


class UMPAModelNoDF(UMPAModelBase):
    """Mirror of ``UMPAModelNoDF`` (``model.pyx:758-822``)."""
    Nparam = 4
    safe_crop = 0
    _kind = KIND_NODF

    def min(self, i, j):
        values = np.zeros((self.Nparam,), dtype=NPDOUBLE)
        self._min(i, j, values)
        return values

    def cost(self, i, j, sx, sy):
        """Cost and transmission at pixel (i, j) for the shift (sx rows, sy columns), rounded."""
        values = np.zeros(2)
        self._lib.check(self._lib.cost(self._handle, int(i), int(j), _cround(sx), _cround(sy),
                                       _lib._ptr(values, _lib._dp)), "cost")
        return (values[0], values[1])

    def match(self, step=None, dxdy=None, ROI=None, num_threads=None, quiet=False):
        result = self._match(step=step, dxdy=dxdy, ROI=ROI, num_threads=num_threads, quiet=quiet)
        return self._unpack(result, False)


class UMPAModelDF(UMPAModelBase):
    """Mirror of ``UMPAModelDF`` (``model.pyx:824-896``)."""
    Nparam = 5
    safe_crop = 0
    _kind = KIND_DF

    def min(self, i, j):
        values = np.zeros((self.Nparam,), dtype=NPDOUBLE)
        self._min(i, j, values)
        return values

    def cost(self, i, j, sx, sy):
        values = np.zeros(3)
        self._lib.check(self._lib.cost(self._handle, int(i), int(j), _cround(sx), _cround(sy),
                                       _lib._ptr(values, _lib._dp)), "cost")
        return (values[0], values[1], values[2])

    def match(self, step=None, dxdy=None, ROI=None, num_threads=None, quiet=False):
        result = self._match(step=step, dxdy=dxdy, ROI=ROI, num_threads=num_threads, quiet=quiet)
        return self._unpack(result, True)

    @property
    def Im(self):
        """Mean ref frame amplitude: never set by the reference either (``Model.h:148``)."""
        return 0.0


class UMPAModelDFKernel(UMPAModelBase):
    """Mirror of ``UMPAModelDFKernel`` (``model.pyx:899-997``): the reference blurred on the fly by a
    per-pixel 17x17 Gaussian ``exp(-a i^2 - b ij - c j^2)``; ``abc`` is an input.  Runs on the general
    direct kernel (289 taps per window pixel, as heavy as in the reference)."""
    Nparam = 7
    safe_crop = 8
    _kind = KIND_DFKERNEL

    def min(self, i, j, a, b, c):
        values = np.zeros((self.Nparam,), dtype=NPDOUBLE)
        values[4], values[5], values[6] = a, b, c
        self._min(i, j, values)
        return values

    def cost(self, i, j, sx, sy, a, b, c):
        values = np.zeros(5)
        values[2], values[3], values[4] = a, b, c
        self._lib.check(self._lib.cost(self._handle, int(i), int(j), _cround(sx), _cround(sy),
                                       _lib._ptr(values, _lib._dp)), "cost")
        return (values[0], values[1])

    def match(self, step=None, abc=None, dxdy=None, ROI=None, num_threads=None, quiet=False):
        s0, s1 = self._convert_ROI_slice(ROI, step)
        self._set_ROI((s0, s1))
        N0, N1 = self._counts(s0, s1)
        sh = (N0, N1)
        if abc is None:
            raise RuntimeError('abc array has to be provided')
        elif abc.shape != sh + (3,):
            raise RuntimeError('Wrong array shape for abc: %s, should be %s' % (abc.shape, sh + (3,)))
        values = np.zeros(sh + (self.Nparam,), dtype=np.float64)
        values[:, :, -3:] = abc
        result = self._match(step=step, input_values=values, dxdy=dxdy, ROI=ROI,
                             num_threads=num_threads, quiet=quiet)
        return self._unpack(result, False)
