# Drop-in replacement for /root/reference/models/__init__.py:6-7 (Option A)
from step_amd import BaseNet, ROINet, TwoBranchNet, ContextNet  # noqa: F401
