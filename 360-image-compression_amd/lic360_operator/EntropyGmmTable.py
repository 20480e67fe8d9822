"""Import-path alias: the reference keeps EntropyGmmTable, EntropyBatchGmmTable in lic360_operator/EntropyGmmTable.py."""
from .tables import EntropyGmmTable, EntropyBatchGmmTable  # noqa: F401
