"""pivlfn -- MI355X-native PIV-LiteFlowNet inference path (drop-in for the reference's
`src/models.py` + `src/correlation.py` + `inference.estimate`).

The compute is in `libpivlfn.so` (hand-written HIP for gfx950, C ABI in include/pivlfn.h); this package is
the thin Python host side that mirrors the reference's names and argument meaning.  There is no CPU
fallback: importing works anywhere, but every op raises if the library or a GPU is missing.
"""
from .correlation import FunctionCorrelation, ModuleCorrelation          # noqa: F401
from .models import LiteFlowNet, LiteFlowNet2, Network, backwarp, hui_liteflownet, piv_liteflownet  # noqa: F401
from .inference import Inference, estimate                               # noqa: F401

__all__ = ["FunctionCorrelation", "ModuleCorrelation", "LiteFlowNet", "LiteFlowNet2", "Network", "backwarp",
           "hui_liteflownet", "piv_liteflownet", "estimate", "Inference"]
