"""Import-path alias: the reference keeps TileExtract, TileExtractBatch in lic360_operator/TileExtract.py."""
from .planes import TileExtract, TileExtractBatch  # noqa: F401
