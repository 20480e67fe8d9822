"""Import-path alias: the reference keeps CconvEc, CconvEcBatch in lic360_operator/CconvEc.py."""
from .conv import CconvEc, CconvEcBatch  # noqa: F401
