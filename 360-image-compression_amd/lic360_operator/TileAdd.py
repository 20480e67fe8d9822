"""Import-path alias: the reference keeps TileAdd in lic360_operator/TileAdd.py."""
from .planes import TileAdd  # noqa: F401
