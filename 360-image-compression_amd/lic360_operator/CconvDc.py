"""Import-path alias: the reference keeps CconvDc, CconvDcBatch in lic360_operator/CconvDc.py."""
from .conv import CconvDc, CconvDcBatch  # noqa: F401
