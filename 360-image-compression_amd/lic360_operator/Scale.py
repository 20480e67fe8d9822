"""Import-path alias: the reference keeps Scale in lic360_operator/Scale.py."""
from .quantize import Scale  # noqa: F401
