"""Import-path alias: the reference keeps EntropyTable in lic360_operator/EntropyTable.py."""
from .tables import EntropyTable  # noqa: F401
