"""Alias of pivlfn.inference under the reference's import path (run.py:16: `from inference import Inference, estimate`)."""
from pivlfn.inference import Inference, estimate  # noqa: F401
