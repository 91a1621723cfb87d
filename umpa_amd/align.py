"""
The callers next to the matching path that the reference keeps in ``UMPA/align.py``:

* ``correct_bad_pixels`` (reference ``align.py:661-732``) -- outlier repair of a result map, on the
  device (``umpa_hip_correct_bad_pixels``, include/umpa_hip.h);
* ``UMPA_normal`` / ``UMPA_nobias`` (reference ``align.py:12-117``) -- a dark-field match (and, for
  ``_nobias``, a second match of the reference stack against itself whose ``dx``/``dy`` are
  subtracted) followed by the repair of ``dx`` and ``dy`` with the threshold ``shift``.

Same signatures, defaults and result dictionaries as the reference.  The registration utilities of
``align.py`` (``shift_best``, ``find_shift`` ...) are pre-processing and are not part of this package.
"""
import ctypes

import numpy as np

from . import _lib, model

__all__ = ["correct_bad_pixels", "UMPA_normal", "UMPA_nobias"]


def correct_bad_pixels(img_in, th=None, iterations=1, dims=(-2, -1), p=0.5, device=None):
    """
    Replace outliers by the median of their neighbours (reference ``align.py:661-732``).

    Values outside ``[-th, th]`` -- or, with ``th=None``, outside the ``p`` / ``100-p`` percentiles --
    are "bad"; ``iterations`` times, every bad pixel becomes the median of its two neighbours along each
    of ``dims`` (edges reflect), all taken from the image as it was before the pass.  Returns a new array.

    ``dims`` may name one or two axes (the reference's default ``(-2, -1)`` suits a stack of maps);
    other axes are batch axes.  More than two neighbour axes are not implemented.
    """
    img_in = np.asarray(img_in)
    img = np.array(img_in, dtype=np.float64, copy=True)
    if th is None:
        th = [np.percentile(img, p), np.percentile(img, 100 - p)]     # align.py:703-706
    else:
        th = [-th, th]
    lo, hi = float(min(th)), float(max(th))
    axes = [d % img.ndim for d in dims]
    if len(set(axes)) != len(axes) or not 1 <= len(axes) <= 2:
        raise NotImplementedError("correct_bad_pixels: `dims` must name one or two distinct axes, got %r" % (dims,))
    if img.size == 0:
        return img.astype(img_in.dtype, copy=False)
    # neighbour axes last, batch axes first
    work = np.ascontiguousarray(np.moveaxis(img, axes, list(range(img.ndim - len(axes), img.ndim))))
    W = work.shape[-1]
    H = work.shape[-2] if len(axes) == 2 else 1
    nimg = work.size // (H * W)
    if W < 2 or (len(axes) == 2 and H < 2):
        raise IndexError("correct_bad_pixels: the neighbour axes need at least two entries")
    out = np.empty_like(work)
    lib = _lib.hip()
    dev = model._default_device() if device is None else int(device)
    rc = lib.correct_bad_pixels(work.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p),
                                nimg, H, W, len(axes), lo, hi, int(iterations), dev, 0, None)
    lib.check(rc, "correct_bad_pixels")
    out = np.ascontiguousarray(np.moveaxis(out, list(range(img.ndim - len(axes), img.ndim)), axes))
    return out.astype(img_in.dtype, copy=False)                       # the reference works in the input's dtype


def _model_kwargs(window, shift, pos_list, mask_list):
    kw = dict(window_size=window, max_shift=shift)
    if pos_list is not None:
        kw["pos_list"] = pos_list
    if mask_list is not None:
        kw["mask_list"] = mask_list
    return kw


_ALL = (slice(None, None, None), slice(None, None, None))


def UMPA_normal(sams, refs, window=1, shift=3, pos_list=None, mask_list=None, assign_coordinates='sam',
                num_threads=None, ROI=_ALL):
    """Dark-field match without bias correction, ``dx``/``dy`` repaired with threshold ``shift``
    (reference ``align.py:12-61``)."""
    PM = model.UMPAModelDF(sams, refs, **_model_kwargs(window, shift, pos_list, mask_list))
    PM.assign_coordinates = assign_coordinates
    res = PM.match(num_threads=num_threads, ROI=ROI, quiet=True)
    res['dx'] = correct_bad_pixels(res['dx'], shift)
    res['dy'] = correct_bad_pixels(res['dy'], shift)
    return res


def UMPA_nobias(sams, refs, window=1, shift=3, pos_list=None, mask_list=None, assign_coordinates='sam',
                num_threads=None, ROI=_ALL):
    """Dark-field match minus the match of the reference stack against itself, ``dx``/``dy`` repaired with
    threshold ``shift`` (reference ``align.py:63-117``; the bias model keeps the default coordinates, ``:114``)."""
    kw = _model_kwargs(window, shift, pos_list, mask_list)
    # The reference builds two models (align.py:98-113).  Both matches share the reference stack: one model keeps it
    # (and the reference-side maps of the tiled path) resident on the GPU, matches it against itself first and then
    # gets the sample stack swapped in -- same numbers, half the uploads and allocations.  That needs frames the
    # library owns (host arrays) of matching shapes; device tensors (borrowed, not copied) and anything the model
    # constructor would reject go the reference's way, two models, so that its error messages apply.
    sams_l, refs_l = list(sams), list(refs)
    shared = (len(sams_l) == len(refs_l) and len(refs_l) > 0 and not hasattr(refs_l[0], "data_ptr")
              and not hasattr(sams_l[0], "data_ptr")
              and all(tuple(np.shape(a)) == tuple(np.shape(b)) for a, b in zip(sams_l, refs_l)))
    if shared:
        PM = model.UMPAModelDF(refs, refs, **kw)
        res_b = PM.match(num_threads=num_threads, ROI=ROI, quiet=True)
        bdx, bdy = np.array(res_b['dx']), np.array(res_b['dy'])
        PM.update_frames(sam_list=sams)
    else:
        PM = model.UMPAModelDF(sams, refs, **kw)                      # raises the reference's messages on bad input
        PMb = model.UMPAModelDF(refs, refs, **kw)
        res_b = PMb.match(num_threads=num_threads, ROI=ROI, quiet=True)
        bdx, bdy = np.array(res_b['dx']), np.array(res_b['dy'])
    PM.assign_coordinates = assign_coordinates
    res = PM.match(num_threads=num_threads, ROI=ROI, quiet=True)
    res['dx'] = correct_bad_pixels(res['dx'] - bdx, shift)
    res['dy'] = correct_bad_pixels(res['dy'] - bdy, shift)
    return res
