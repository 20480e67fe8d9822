"""Import-path alias: the reference keeps Dquant in lic360_operator/Dquant.py."""
from .quantize import Dquant  # noqa: F401
