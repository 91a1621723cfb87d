"""``UMPA.align`` -> :mod:`umpa_amd.align` (``correct_bad_pixels``, ``UMPA_normal``, ``UMPA_nobias``)."""
from umpa_amd.align import *                                          # noqa: F401,F403
