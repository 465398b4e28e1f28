"""step_amd/tube_math.py -- box / tube parameterisation used inside the hot path
(reference: utils/tube_utils.py:127-189, called from models/two_branch.py:306,320 and
utils/utils.py:68-79).  Tiny element-wise torch math on [n,4] boxes (x1,y1,x2,y2), "+1" pixel sizes."""
import numpy as np
import torch


def get_center_size(boxes):
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    return boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h, w, h


def encode_coef(gt_tubes, tubes):
    """regression target of gt w.r.t. proposal: ((gx-x)/w, (gy-y)/h, log(gw/w), log(gh/h))"""
    gx, gy, gw, gh = get_center_size(gt_tubes)
    x, y, w, h = get_center_size(tubes)
    return torch.stack(((gx - x) / w, (gy - y) / h, torch.log(gw / w), torch.log(gh / h)), dim=1)


def decode_coef(anchors, deltas):
    """inverse of encode_coef; the max corner gets -1 (tube_utils.py:186-187)"""
    x, y, w, h = get_center_size(anchors)
    px = w * deltas[:, 0] + x
    py = h * deltas[:, 1] + y
    pw = w * torch.exp(deltas[:, 2])
    ph = h * torch.exp(deltas[:, 3])
    return torch.stack((px - 0.5 * pw, py - 0.5 * ph, px + 0.5 * pw - 1, py + 0.5 * ph - 1), dim=1)


def flatten_tubes(tubes, batch_idx=False):
    """list of [n_i,T,dim] arrays -> ([sum n_i, T, dim(+1)], [n_i]); col 0 = frame index b*T+t
    (tube_utils.py:214-246)."""
    T = tubes[0].shape[1]
    flat, nums = [], []
    for i, t in enumerate(tubes):
        nums.append(t.shape[0])
        if t.shape[0] == 0:
            continue
        if batch_idx:
            idx = np.broadcast_to((np.arange(T) + i * T).reshape(1, T, 1), (t.shape[0], T, 1)).astype(t.dtype)
            flat.append(np.concatenate((idx, t), axis=2))
        else:
            flat.append(t.copy())
    return np.concatenate(flat, axis=0), nums


def valid_tubes(tubes, width=400, height=400):
    """clamp to the image; boxes that are not at least 3 px in each direction become the whole
    image (tube_utils.py:59-92), vectorised.  Works on numpy arrays or torch tensors (returns a copy)."""
    if isinstance(tubes, np.ndarray):
        b = tubes.reshape(-1, 4).copy()
        b[:, 0] = np.maximum(0, b[:, 0])
        b[:, 1] = np.maximum(0, b[:, 1])
        b[:, 2] = np.minimum(width, b[:, 2])
        b[:, 3] = np.minimum(height, b[:, 3])
        bad = ~((b[:, 0] < b[:, 2] - 2) & (b[:, 1] < b[:, 3] - 2))
        b[bad] = np.asarray([0, 0, width, height], dtype=b.dtype)
        return b.reshape(tubes.shape)
    b = tubes.reshape(-1, 4)
    x1, y1 = b[:, 0].clamp(min=0), b[:, 1].clamp(min=0)
    x2, y2 = b[:, 2].clamp(max=width), b[:, 3].clamp(max=height)
    bad = ~((x1 < x2 - 2) & (y1 < y2 - 2))
    # fill kernels only (no host->device scalar copies): the whole step loop is hipGraph-capturable
    out = torch.stack((torch.where(bad, torch.zeros_like(x1), x1), torch.where(bad, torch.zeros_like(y1), y1),
                       torch.where(bad, torch.full_like(x2, float(width)), x2),
                       torch.where(bad, torch.full_like(y2, float(height)), y2)), 1)
    return out.reshape(tubes.shape)


def extrapolate_tubes(tubes, T=6, height=400, width=400):
    """[n, Tl, 4] -> [n, Tl + 2T, 4]: T frames of linear extrapolation on either side, frame by frame from the two frames T
    apart ((T/(T-1)) a - (1/(T-1)) b), then clamped to [0, width-1] x [0, height-1] (tube_utils.py:159-176; the reference calls
    it with the default 400 x 400 whatever the image size).  numpy in -> numpy out (fp32), torch in -> torch out (same device,
    element-wise kernels only: no host synchronisation)."""
    is_np = isinstance(tubes, np.ndarray)
    n, Tl = tubes.shape[0], tubes.shape[1]
    if is_np:
        new = np.zeros((n, Tl + 2 * T, tubes.shape[2]), dtype=np.float32)
    else:
        new = torch.zeros((n, Tl + 2 * T, tubes.shape[2]), dtype=torch.float32, device=tubes.device)
    new[:, T:T + Tl] = tubes
    L = Tl + 2 * T
    for i in range(T):
        new[:, L - T + i] = (T / (T - 1)) * new[:, L - T + i - 1] - (1 / (T - 1)) * new[:, L - T + i - T]
        new[:, T - i - 1] = (T / (T - 1)) * new[:, T - i] - (1 / (T - 1)) * new[:, T - i + T - 1]
    if is_np:
        new[:, :, 0] = np.maximum(0, new[:, :, 0])
        new[:, :, 1] = np.maximum(0, new[:, :, 1])
        new[:, :, 2] = np.minimum(width - 1, new[:, :, 2])
        new[:, :, 3] = np.minimum(height - 1, new[:, :, 3])
        return new
    lo = torch.stack((new[..., 0].clamp(min=0), new[..., 1].clamp(min=0), new[..., 2].clamp(max=width - 1), new[..., 3].clamp(max=height - 1)), -1)
    return lo


def generate_anchors(scales=(4.0 / 3.0, 2.0), overlaps=(5.0 / 6.0, 3.0 / 4.0)):
    """The initial tube grid of the default anchor mode "1": for each (scale, overlap) a regular grid of square boxes
    of side 1/scale with stride side*(1-overlap), normalised [x1, y1, x2, y2] -- 9 + 25 = 34 boxes
    (data/data_utils.py:19-45)."""
    out = []
    for scale, overlap in zip(scales, overlaps):
        size = 1.0 / scale
        stride = size * (1 - overlap)
        i = 0
        while i + size <= 1:
            j = 0
            while j + size <= 1:
                out.append([i, j, i + size, j + size])
                j += stride
            i += stride
    return np.asarray(out, dtype=np.float32)


ANCHOR_MODES = {"1": ((4 / 3, 2), (5 / 6, 3 / 4)), "2": ((4 / 3, 2, 3), (5 / 6, 3 / 4, 1 / 2)),
                "3": ((4 / 3, 2, 3, 4), (5 / 6, 3 / 4, 1 / 2, 1 / 4)), "4": ((4 / 3, 2, 3, 4, 5), (5 / 6, 3 / 4, 1 / 2, 1 / 4, 0))}


def anchor_tubes(anchor_mode="1", T=3):
    """[n, T, 4] normalised initial tubes of `--anchor_mode` 1 | 2 | 3 | 4 (34 / 59 / 84 / 109 tubes; config.py:93,
    data/ava.py:342-354): the anchor grid repeated over the T frames; any other mode is the reference's single void tube."""
    if anchor_mode not in ANCHOR_MODES:
        return np.zeros([1, T, 4])
    a = generate_anchors(*ANCHOR_MODES[anchor_mode])
    return np.tile(np.expand_dims(a, axis=1), (1, T, 1))
