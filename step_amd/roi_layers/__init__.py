"""step_amd.roi_layers -- drop-in for `external.maskrcnn_benchmark.roi_layers`
(/root/reference/external/maskrcnn_benchmark/roi_layers/__init__.py:29-35): the same five names with
the same call signatures, backed by the gfx950 kernels of libstep_amd.so instead of the `_C`
pybind extension (csrc/vision.cpp:30-36)."""
from .nms import nms, nms_batched
from .roi_align import ROIAlign, roi_align
from .roi_pool import ROIPool, roi_pool

__all__ = ["nms", "roi_align", "ROIAlign", "roi_pool", "ROIPool", "nms_batched"]
