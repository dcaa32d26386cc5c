"""deformationpyramid_amd -- MI355X-native NDP registration hot path.

Drop-in for the reference's Registration.register() / Deformation_Pyramid /
compute_truncated_chamfer_distance surface.
"""
from .layout import LayerDesc  # noqa: F401
from .nets import Deformation_Pyramid, NDPLevel  # noqa: F401

__version__ = "0.1.0"
