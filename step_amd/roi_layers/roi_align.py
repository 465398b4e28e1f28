"""ROIAlign operator API (reference: roi_layers/roi_align.py:45-100).

`roi_align(input, rois, output_size, spatial_scale, sampling_ratio)` and the `ROIAlign` module keep
the reference signatures.  input is logical [B,C,H,W]; a channels-last input (our own backbone's
output) takes the coalesced NHWC kernel and yields a channels-last output, a torch-contiguous one
takes the NCHW kernel.  fp16/bf16 inputs are computed with fp32 arithmetic inside the kernel (the
reference forces fp32 through apex `float_function`, roi_align.py:89)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import ops


class _ROIAlignFn(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, deterministic=True):
        ctx.save_for_backward(roi)
        ctx.deterministic = bool(deterministic)
        ctx.geom = (_pair(output_size), float(spatial_scale), int(sampling_ratio), tuple(input.shape))
        ph, pw = ctx.geom[0]
        return ops.roi_align_forward(input, roi, ph, pw, spatial_scale, sampling_ratio)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (roi,) = ctx.saved_tensors
        (ph, pw), scale, sr, (B, C, H, W) = ctx.geom
        grad_input = ops.roi_align_backward(grad_output, roi, ph, pw, scale, sr, B, C, H, W, deterministic=ctx.deterministic)
        return grad_input, None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, deterministic=True):
    """The reference's five arguments (roi_layers/roi_align.py:73) plus `deterministic`: which backward THIS call's autograd node
    runs -- the fixed-order gather (default, bit-reproducible) or the reference's fp32-atomics scatter.  Chosen per call and
    carried by the node, never read from process state."""
    return _ROIAlignFn.apply(input, rois, output_size, spatial_scale, sampling_ratio, deterministic)


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, deterministic=True):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.deterministic = deterministic          # backward mode of this module's calls (see roi_align)

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.deterministic)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
