"""Logger (reference lic360_operator/Logger.py) -> extras.py"""
from .extras import Logger  # noqa: F401
