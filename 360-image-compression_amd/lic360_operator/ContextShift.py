"""Import-path alias: the reference keeps ContextShift in lic360_operator/ContextShift.py."""
from .quantize import ContextShift  # noqa: F401
