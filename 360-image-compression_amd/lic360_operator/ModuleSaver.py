"""ModuleSaver (reference lic360_operator/ModuleSaver.py) -> extras.py"""
from .extras import ModuleSaver  # noqa: F401
