"""Import-path alias: the reference keeps Dtow in lic360_operator/Dtow.py."""
from .quantize import Dtow  # noqa: F401
