"""Alias of pivlfn.correlation under the reference's import path (src/correlation.py)."""
from pivlfn.correlation import FunctionCorrelation, ModuleCorrelation  # noqa: F401
