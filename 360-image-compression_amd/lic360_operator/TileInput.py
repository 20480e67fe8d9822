"""Import-path alias: the reference keeps TileInput in lic360_operator/TileInput.py."""
from .planes import TileInput  # noqa: F401
