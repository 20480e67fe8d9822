"""SSIM (reference lic360_operator/pytorch_ssim.py) -> extras.py"""
from .extras import SSIM  # noqa: F401
