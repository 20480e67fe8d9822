"""Import-path alias: the reference keeps SphereTrim in lic360_operator/SphereTrim.py."""
from .sphere import SphereTrim  # noqa: F401
