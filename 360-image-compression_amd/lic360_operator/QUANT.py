"""Import-path alias: the reference keeps QUANT in lic360_operator/QUANT.py."""
from .quantize import QUANT  # noqa: F401
