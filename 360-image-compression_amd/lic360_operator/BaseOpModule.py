"""Import-path alias: the reference keeps BaseOpModule in lic360_operator/BaseOpModule.py."""
from .base import BaseOpModule  # noqa: F401
