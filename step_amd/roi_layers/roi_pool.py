"""ROIPool operator API (reference: roi_layers/roi_pool.py:45-98)."""
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import ops


class _ROIPoolFn(Function):
    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale):
        ph, pw = _pair(output_size)
        output, argmax = ops.roi_pool_forward(input, roi, ph, pw, spatial_scale)
        ctx.save_for_backward(roi, argmax)
        ctx.geom = ((ph, pw), tuple(input.shape))
        ctx.mark_non_differentiable(argmax)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        roi, argmax = ctx.saved_tensors
        (ph, pw), (B, C, H, W) = ctx.geom
        return ops.roi_pool_backward(grad_output, argmax, roi, ph, pw, B, C, H, W), None, None, None


roi_pool = _ROIPoolFn.apply


class ROIPool(nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = output_size
        self.spatial_scale = spatial_scale

    def forward(self, input, rois):
        return roi_pool(input, rois, self.output_size, self.spatial_scale)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s)" % (self.__class__.__name__, self.output_size, self.spatial_scale)
