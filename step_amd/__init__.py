"""step_amd -- MI355X (gfx950) implementation of the STEP hot path: I3D backbone, two-branch head,
ROIAlign / ROIPool / NMS.  Drop-in for the reference's `models` package and
`external.maskrcnn_benchmark.roi_layers` package."""
from .backbone import BaseNet, I3D, I3D_head, build_base_i3d, weights_init  # noqa: F401
from .heads import ContextNet, ROINet, TwoBranchNet  # noqa: F401
from . import dist  # noqa: F401
from .optim import FlatAdam, LossScaler  # noqa: F401

__all__ = ["BaseNet", "ROINet", "TwoBranchNet", "ContextNet", "I3D", "I3D_head"]
__version__ = "0.1.0"
