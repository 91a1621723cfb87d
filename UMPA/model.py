"""``UMPA.model`` -> :mod:`umpa_amd.model` (``from UMPA.model import UMPAModelDF`` as in the reference)."""
from umpa_amd.model import *                                          # noqa: F401,F403
from umpa_amd.model import UMPAModelBase, UMPAModelDF, UMPAModelDFKernel, UMPAModelNoDF, spm, spmq   # noqa: F401
