"""deformationpyramid_amd -- MI355X-native NDP registration hot path.

Drop-in for the reference's Registration.register() / Deformation_Pyramid /
compute_truncated_chamfer_distance surface.
"""
import os as _os

# The batched path keeps four or five HIP streams busy (two engines, the pair producer, the final warps).  ROCm maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two engine streams sharing one queue serialise and the
# overlap between the engines is lost (measured: 682 pairs/s instead of 751).  The variable is read when the HIP
# runtime initialises, i.e. at the first device call after this import; an explicit user setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .layout import LayerDesc  # noqa: F401,E402
from .nets import Deformation_Pyramid, NDPLevel  # noqa: F401,E402

__version__ = "0.1.0"
