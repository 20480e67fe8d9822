"""Import-path alias: the reference keeps Imp2mask in lic360_operator/Imp2mask.py."""
from .quantize import Imp2mask  # noqa: F401
