"""
Convenience API: ``match`` and ``match_unbiased`` with the reference's signatures and
behaviour (reference ``UMPA/speckle_matching.py:12-75``), running on the HIP models.
"""
from . import model

__all__ = ["match", "match_unbiased"]


def _contiguous(frames, what):
    def ok(x):
        return x.is_contiguous() if hasattr(x, "is_contiguous") else x.flags.c_contiguous
    if any(not ok(x) for x in frames):
        print('Warning: provided list of %s frames are not c contiguous - working with a copy.' % what)
        frames = [x.contiguous() if hasattr(x, "contiguous") else x.copy() for x in frames]
    return frames


def match(Isample, Iref, Nw, mask=None, step=1, max_shift=4, df=True):
    """
    Speckle matching with the UMPA algorithm (reference ``speckle_matching.py:12-48``).

    ``max_shift`` is accepted and ignored, exactly as in the reference (``:23``, ``:44-46``): the
    models are built with their default ``max_shift=4``.

    Returns the result dictionary of ``UMPAModelDF.match`` / ``UMPAModelNoDF.match``:
    'T', 'dx', 'dy', 'df' (dark-field model only), 'f', 'err' and the ``debug_*`` arrays.
    """
    Isample = _contiguous(Isample, 'sample')
    Iref = _contiguous(Iref, 'reference')
    cls = model.UMPAModelDF if df else model.UMPAModelNoDF
    PM = cls(sam_list=Isample, ref_list=Iref, mask_list=mask, window_size=Nw)
    return PM.match(step=step)


def match_unbiased(Isample, Iref, Nw, mask=None, step=1, max_shift=4, df=True, bias=True):
    """
    Speckle matching including bias correction (reference ``speckle_matching.py:51-75``):
    ``bias=True`` matches the reference stack against itself and subtracts the resulting
    ``dx``/``dy``; ``bias=False`` subtracts nothing; a ``(dx, dy)`` pair is used as given.
    """
    if bias is True:
        cls = model.UMPAModelDF if df else model.UMPAModelNoDF
        Isample = _contiguous(Isample, 'sample')
        Iref = _contiguous(Iref, 'reference')
        PMref = cls(sam_list=Iref, ref_list=Iref, mask_list=mask, window_size=Nw)
        bias_result = PMref.match(step=step)
        dx = bias_result['dx']
        dy = bias_result['dy']
        if PMref._lib.is_hip and not hasattr(PMref._sam[0], "data_ptr"):
            # both matches share the reference stack: keep it resident on the GPU and swap only the sample stack
            # (same numbers as building a second model, which is what the reference does)
            PMref.update_frames(sam_list=Isample)
            PMref.ROI = None
            result = PMref.match(step=step)
            result['dx'] -= dx
            result['dy'] -= dy
            return result
    elif bias is False:
        dx = 0.
        dy = 0.
    else:
        dx, dy = bias
    result = match(Isample=Isample, Iref=Iref, Nw=Nw, mask=mask, step=step, max_shift=max_shift, df=df)
    result['dx'] -= dx
    result['dy'] -= dy
    return result
