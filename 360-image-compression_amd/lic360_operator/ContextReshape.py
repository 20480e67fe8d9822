"""Import-path alias: the reference keeps ContextReshape in lic360_operator/ContextReshape.py."""
from .quantize import ContextReshape  # noqa: F401
