"""Import-path alias: the reference keeps SpherePad in lic360_operator/SpherePad.py."""
from .sphere import SpherePad  # noqa: F401
