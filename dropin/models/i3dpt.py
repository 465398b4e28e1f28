# Drop-in for /root/reference/models/i3dpt.py:165-262 (networks.py:13 / two_branch.py:12 import these names)
from step_amd.backbone import I3D, I3D_head, Unit3D as Unit3Dpy, MaxPoolTF as MaxPool3dTFPadding, Mixed  # noqa: F401
