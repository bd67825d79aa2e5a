"""Alias of pivlfn.models under the reference's import path (src/models.py)."""
from pivlfn.models import LiteFlowNet, LiteFlowNet2, Network, backwarp, hui_liteflownet, piv_liteflownet  # noqa: F401

__all__ = ['hui_liteflownet', 'piv_liteflownet']  # same export list as the reference (src/models.py:8)
