"""Import-path alias: the reference keeps SphereCutEdge in lic360_operator/SphereCutEdge.py."""
from .sphere import SphereCutEdge  # noqa: F401
