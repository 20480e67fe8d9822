"""Import-path alias: the reference keeps SphereLatScaleNet, ScaleResidualBlock in lic360_operator/SphereLatScaleNet.py."""
from .sphere import SphereLatScaleNet, ScaleResidualBlock  # noqa: F401
