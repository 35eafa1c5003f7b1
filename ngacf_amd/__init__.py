"""ngacf_amd -- MI355X-native graph-attention propagation + BPR scoring behind the module
surface of cleverer123/NGACF (graphattention/SPUIGACF.py, BPRLoss.py).

Importing the package loads ngacf_amd/lib/libngacf_hip.so through ctypes and fails loudly
when it is missing; there is no CPU fallback."""
from . import _lib                                             # noqa: F401  (loads the HIP library)
from ._lib import version                                      # noqa: F401
from .BPRLoss import BPRLoss                                   # noqa: F401
from .plan import GraphPlan, build_plan                        # noqa: F401
from .SPUIGACF import (SPUIGACF, SPUIGAGPCF, SPUIMultiGACF, SPUIStackedGACF, GPLayer, SpUIGAT, SPUIMultiGAT,   # noqa: F401
                       SpUIGraphAttentionLayer, set_compute_dtype)
from . import SPGA                                             # noqa: F401,E402  (SPGA.SpGraphAttentionLayer, SPGA.SpGAT)
from . import modules                                          # noqa: F401,E402  (dense multi-head attention: MultiHeadAttention, ATTLayer, GACFV1, GACFMask)
from . import optim                                            # noqa: F401,E402  (optim.Adam: torch.optim.Adam's arithmetic and state, one launch per step)
