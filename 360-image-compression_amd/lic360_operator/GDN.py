"""GDN (reference lic360_operator/GDN.py) -> extras.py"""
from .extras import GDN  # noqa: F401
