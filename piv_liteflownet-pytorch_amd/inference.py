"""Alias of pivlfn.inference under the reference's import path (inference.py: `from inference import estimate`)."""
from pivlfn.inference import estimate  # noqa: F401
