"""MaskConv2 (reference lic360_operator/MaskConstrain.py) -> extras.py"""
from .extras import MaskConv2  # noqa: F401
