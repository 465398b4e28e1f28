"""NMS operator API (reference: roi_layers/nms.py:38, csrc/nms.h:34-52).

`nms(dets[k,4], scores[k], thr)` returns the kept ORIGINAL indices in ascending order as an int64
CPU tensor -- what the reference returns on every path (cpu/nms_cpu.cpp:88; the CUDA op also returns
a CPU tensor, nms.cu:151-154) -- with the reference CPU op's semantics (IoU >= thr suppresses),
bit-exact.  Inputs may live on the GPU (no `.cpu()` needed first) or on the CPU (the reference's
callers move them there, test.py:158-160: they are uploaded, the work still happens on the GPU).

`nms_batched` is the tube-batched form ("nms_3d"): every (clip, class) group in one launch, result
left on the device as a keep mask.
"""
import torch

from .. import ops


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("step_amd: no ROCm device available and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


_COUNTS = {}      # (device, k) -> the one-group `counts` tensor of a k-box call (the reference's callers make 180 x B such calls per batch)


def _counts(dev, k):
    c = _COUNTS.get((dev, k))
    if c is None:
        if len(_COUNTS) > 4096:
            _COUNTS.clear()
        c = _COUNTS[(dev, k)] = torch.tensor([k], dtype=torch.int32, device=dev)
    return c


def nms(dets, scores, threshold):
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device="cpu")
    # the reference dispatches on the box dtype (AT_DISPATCH_FLOATING_TYPES, cpu/nms_cpu.cpp:95): double stays double (bit-exact
    # against nms_cpu_kernel<double>); 16-bit inputs are up-cast as apex's float_function does (roi_layers/nms.py:36-43)
    dt = torch.float64 if dets.dtype == torch.float64 else torch.float32
    k = dets.shape[0]
    if dets.is_cuda:
        dev = dets.device
        b = dets.detach().to(dtype=dt).reshape(1, k, 4)
        s = scores.detach().to(device=dev, dtype=dt).reshape(1, k)
    else:
        # CPU tensors (test.py:158-160 moves them there before every call): boxes and scores travel in ONE staging buffer
        # [4k | k] -- one upload, one launch, one download of the k-byte keep mask per call
        dev = _device()
        d = torch.cat([dets.detach().to(dt).reshape(-1), scores.detach().to(device="cpu", dtype=dt).reshape(-1)]).to(dev)
        b, s = d[:4 * k].view(1, k, 4), d[4 * k:].view(1, k)
    keep = ops.nms_batched(b, s, _counts(dev, k), threshold)
    # the kept ORIGINAL indices, ascending, on the host: the mask comes down in one copy and is enumerated there (a device-side
    # nonzero costs a second kernel and a second synchronisation for the same k bytes)
    return torch.nonzero(keep[0].cpu()).squeeze(1)


def nms_batched(boxes, scores, counts, threshold):
    """boxes [G,kmax,4], scores [G,kmax], counts [G] on one device -> uint8 keep mask [G,kmax]."""
    return ops.nms_batched(boxes, scores, counts, threshold)
