"""Import-path alias: the reference keeps EntropyGmm in lic360_operator/EntropyGmm.py."""
from .tables import EntropyGmm  # noqa: F401
