"""Import-path alias: the reference keeps CodeContex in lic360_operator/CodeContex.py."""
from .planes import CodeContex  # noqa: F401
