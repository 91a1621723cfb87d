"""``UMPA.speckle_matching`` -> :mod:`umpa_amd.speckle_matching` (``match``, ``match_unbiased``)."""
from umpa_amd.speckle_matching import *                               # noqa: F401,F403
from umpa_amd.speckle_matching import match, match_unbiased           # noqa: F401
