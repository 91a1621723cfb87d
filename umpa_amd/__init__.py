"""
umpa_amd -- MI355X (gfx950) implementation of the UMPA per-pixel matching path.

Drop-in for the hot path of optimato/UMPA: ``match`` / ``match_unbiased`` and the
``UMPAModelNoDF`` / ``UMPAModelDF`` model classes keep the reference's API
(reference ``UMPA/__init__.py:8``, ``UMPA/speckle_matching.py``, ``UMPA/model.pyx``);
the numerics run in ``libumpa_hip.so`` (C ABI: ``include/umpa_hip.h``).  ``align`` holds the three
callers of ``UMPA/align.py`` that wrap the match (``UMPA_normal``, ``UMPA_nobias``, ``correct_bad_pixels``).
"""
from . import model
from . import align
from .model import UMPAModelNoDF, UMPAModelDF, UMPAModelDFKernel
from .speckle_matching import match, match_unbiased

__all__ = ["model", "align", "match", "match_unbiased", "UMPAModelNoDF", "UMPAModelDF", "UMPAModelDFKernel"]
