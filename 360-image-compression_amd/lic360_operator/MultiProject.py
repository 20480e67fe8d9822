"""MultiProject (reference lic360_operator/MultiProject.py) -> extras.py"""
from .extras import MultiProject  # noqa: F401
