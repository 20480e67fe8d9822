"""Import-path alias: the reference keeps ImpMap in lic360_operator/ImpMap.py."""
from .quantize import ImpMap  # noqa: F401
