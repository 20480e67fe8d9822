"""DropGrad (reference lic360_operator/DropGrad.py) -> extras.py"""
from .extras import DropGrad  # noqa: F401
