"""
``UMPA`` -- import-name alias of :mod:`umpa_amd`, so that code written against the reference package
(``import UMPA``, ``from UMPA import model``, ``UMPA.match(...)``: reference ``UMPA/__init__.py:8``,
``speckle_matching.py:9``, ``umpa_multi.py:149``) runs without editing its import lines.  Nothing lives here.
"""
from umpa_amd import *                                                # noqa: F401,F403
from umpa_amd import align, model, speckle_matching                   # noqa: F401
from umpa_amd import __all__ as _all

__all__ = list(_all)
