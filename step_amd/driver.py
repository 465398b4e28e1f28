"""step_amd/driver.py -- the multi-step inference driver over the HIP modules.

Counterpart of utils/utils.py:15-131 (`inference`): per step, ROI-pool the current tubes out of the
backbone feature, run that step's TwoBranchNet, decode the regressed boxes, and (between steps 2 and
3 of the default schedule) extend the tubes in time with the predicted neighbours.  Same `history`
contract: a list with one dict per step holding pred_prob [N,Tl,classes], pred_loc [N,Tl,4],
pred_first_loc / pred_last_loc [N,T,4] and tubes_nums.

Differences from the reference's host glue (they do not change results):
  * everything stays on the device between steps: the tube bookkeeping is tensor ops, there is no
    per-tube `.item()` (utils.py:57), no numpy round trip (utils.py:107-125), no per-box Python loop
    (tube_utils.py:84-88);
  * the ROI-pooled features are produced channels-last by the NHWC ROIAlign kernel straight from the
    channels-last backbone feature (no `.contiguous()` transpose of the slice, utils.py:48).
All three temporal modes of the reference are implemented (utils.py:102-120): "predict" (every shipped script: the head's
neighbour regressions extend the tube), "extrapolate" (linear, tube_utils.py:159-176) and "mean".
"""
import numpy as np
import torch

from .tube_math import extrapolate_tubes, decode_coef, valid_tubes


def _flat_tubes(tubes_list, device, dtype=torch.float32):
    """list (one per clip) of [n_i, T, 4] tensors/arrays -> flat [sum n_i, T, 5] with frame index, nums"""
    rows, nums = [], []
    for b, t in enumerate(tubes_list):
        t = torch.as_tensor(t, dtype=dtype, device=device)
        nums.append(int(t.shape[0]))
        if t.shape[0] == 0:
            continue
        T = t.shape[1]
        idx = (torch.arange(T, device=device, dtype=dtype) + b * T).view(1, T, 1).expand(t.shape[0], T, 1)
        rows.append(torch.cat([idx, t], dim=2))
    return torch.cat(rows, dim=0), nums


def inference(args, conv_feat, context_feat, nets, exec_iter, tubes):
    """args: Namespace with T, NUM_CHUNKS, max_iter, num_classes, image_size, no_context, temporal_mode
    conv_feat [B,T_all,C,H,W] (BaseNet output), context_feat [B,1024,T_all,1,1] or None,
    nets: {'roi_net': ROINet, 'det_net0': TwoBranchNet, ...}, tubes: list of [n_i,T,4] per clip.
    Returns (history, trajectory) like the reference."""
    if getattr(args, "temporal_mode", "predict") not in ("predict", "extrapolate", "mean"):
        raise NotImplementedError("step_amd.driver.inference: temporal_mode %r" % (args.temporal_mode,))
    dev = conv_feat.device
    flat, nums = _flat_tubes(tubes, dev)
    clip_of = torch.as_tensor(np.repeat(np.arange(len(nums)), nums), device=dev)
    return inference_flat(args, conv_feat, context_feat, nets, exec_iter, flat, nums, clip_of)


def inference_flat(args, conv_feat, context_feat, nets, exec_iter, flat, nums, clip_of):
    """The device-resident core of `inference`: flat [N,T,5] tubes (col 0 = frame index), nums = tubes per
    clip, clip_of [N] = clip index of every tube.  Pure tensor ops with static shapes and no host
    synchronisation, so the whole multi-step pipeline can be captured in a hipGraph (GraphedInference)."""
    dev = conv_feat.device
    history, trajectory = [], []
    for i in range(1, exec_iter + 1):
        chunks = args.NUM_CHUNKS[i]
        T_start = int((args.NUM_CHUNKS[args.max_iter] - chunks) / 2) * args.T
        T_length = chunks * args.T
        chunk_idx = [j * args.T + int(args.T / 2) for j in range(chunks)]
        half_T = int(args.T / 2)

        pooled = nets["roi_net"](conv_feat[:, T_start:T_start + T_length], flat)       # [N*Tl, C, 7, 7]
        pooled = pooled.reshape(-1, T_length, *pooled.shape[1:])
        ctx = None
        if not args.no_context:
            ctx = context_feat[clip_of][:, :, T_start:T_start + T_length]               # utils.py:55-57, batched
        prob, local_loc, first_loc, last_loc, _, _, _ = nets["det_net%d" % (i - 1)](pooled, context_feat=ctx)

        pred_prob = prob.view(-1, 1, args.num_classes).expand(-1, T_length, -1)
        flat = flat.to(local_loc)
        pred_loc = decode_coef(flat.reshape(-1, 5)[:, 1:], local_loc.reshape(-1, 4)).view(local_loc.shape)
        lo, hi = chunk_idx[0] - half_T, chunk_idx[0] + half_T + 1
        lo2, hi2 = chunk_idx[-1] - half_T, chunk_idx[-1] + half_T + 1
        pred_first = decode_coef(flat[:, lo:hi].reshape(-1, 5)[:, 1:], first_loc.reshape(-1, 4)).view(first_loc.shape)
        pred_last = decode_coef(flat[:, lo2:hi2].reshape(-1, 5)[:, 1:], last_loc.reshape(-1, 4)).view(last_loc.shape)
        history.append({"pred_prob": pred_prob, "pred_loc": pred_loc, "pred_first_loc": pred_first,
                        "pred_last_loc": pred_last, "tubes_nums": list(nums)})

        # next step's proposals (utils.py:91-129), all clips at once
        if i < args.max_iter and args.NUM_CHUNKS[i + 1] == args.NUM_CHUNKS[i] + 2:
            mode = getattr(args, "temporal_mode", "predict")
            if mode == "predict":
                prop = torch.cat([pred_first, pred_loc, pred_last], dim=1)
            elif mode == "extrapolate":
                prop = extrapolate_tubes(pred_loc, args.T)
            else:                                               # the tube's mean box on both sides (utils.py:116-120)
                m = pred_loc.mean(dim=1, keepdim=True).expand(-1, args.T, -1)
                prop = torch.cat([m, pred_loc, m], dim=1)
        else:
            prop = pred_loc
        prop = valid_tubes(prop, width=args.image_size[0], height=args.image_size[1])
        trajectory.append((prop, torch.argmax(prob, dim=-1)))
        Tn = prop.shape[1]
        idx = (clip_of.to(prop.dtype) * Tn).view(-1, 1, 1) + torch.arange(Tn, device=dev, dtype=prop.dtype).view(1, Tn, 1)
        flat = torch.cat([idx, prop], dim=2)
    return history, trajectory


def postprocess(args, history, conf_thresh=0.01, nms_thresh=0.4, topk=300):
    """Per-(clip, class) NMS on the middle-frame boxes of the final step -- the batched form of the
    reference's Python loop over 60 classes (test.py:157-218).  One nms launch for all groups.
    Returns a list (per clip) of (boxes [m,4] normalised, scores [m], labels [m])."""
    from .roi_layers import nms_batched

    h = history[-1]
    nums = h["tubes_nums"]
    Tl = h["pred_loc"].shape[1]
    mid = Tl // 2
    boxes = valid_tubes(h["pred_loc"][:, mid:mid + 1].float(), args.image_size[0], args.image_size[1])[:, 0]   # [N,4]
    scores = h["pred_prob"][:, mid].float()                                                                   # [N,classes]
    dev = boxes.device
    B, kmax, NC = len(nums), max(nums), scores.shape[1]
    gb = torch.zeros((B, NC, kmax, 4), device=dev)
    gs = torch.full((B, NC, kmax), -1.0, device=dev)
    start = 0
    for b, n in enumerate(nums):
        gb[b, :, :n] = boxes[start:start + n].unsqueeze(0)
        gs[b, :, :n] = scores[start:start + n].t()
        start += n
    # reference masks score > conf_thresh BEFORE nms (test.py:180-186): push the others past `counts`
    valid = gs > conf_thresh
    order = torch.argsort((~valid).to(torch.int8), dim=2, stable=True)             # valid boxes first, original order kept
    gb = torch.gather(gb, 2, order.unsqueeze(-1).expand(-1, -1, -1, 4))
    gs = torch.gather(gs, 2, order)
    counts = valid.sum(2).to(torch.int32)
    keep = nms_batched(gb.view(B * NC, kmax, 4), gs.view(B * NC, kmax), counts.view(-1), nms_thresh).view(B, NC, kmax).bool()
    out = []
    W, H = float(args.image_size[0]), float(args.image_size[1])
    for b in range(B):
        kb = keep[b]
        cls = torch.nonzero(kb)[:, 0]
        bx = gb[b][kb] / torch.tensor([W, H, W, H], device=dev)
        sc = gs[b][kb]
        if topk > 0 and sc.numel() > topk:
            sc, sel = torch.topk(sc, topk)
            bx, cls = bx[sel], cls[sel]
        out.append((bx, sc, cls))
    return out


class GraphedInference:
    """BaseNet -> ContextNet -> multi-step inference for a FIXED batch size / tubes-per-clip, captured once
    in a hipGraph and replayed: the ~200 small launches of the three refinement steps stop being bound by
    Python / launch latency.  `__call__(images, tubes)` copies the inputs into the captured buffers, replays
    and returns (history, conv_feat, context_feat) -- static tensors that the next call overwrites."""

    def __init__(self, args, base_net, context_net, nets, images, tubes):
        self.args, self.base, self.ctx, self.nets = args, base_net, context_net, nets
        dev = images.device
        self.images = images.clone()
        flat, self.nums = _flat_tubes(tubes, dev)
        self.flat0 = flat.clone()
        self.clip_of = torch.as_tensor(np.repeat(np.arange(len(self.nums)), self.nums), device=dev)
        with torch.no_grad():
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                for _ in range(2):                        # warm-up: packs weights, fills caches and the allocator
                    self._run()
            torch.cuda.current_stream(dev).wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._run()

    def _run(self):
        cf = self.base(self.images)
        cx = self.ctx(cf) if not self.args.no_context else None
        hist, _ = inference_flat(self.args, cf, cx, self.nets, self.args.max_iter, self.flat0, self.nums, self.clip_of)
        return hist, cf, cx

    def __call__(self, images, tubes=None):
        self.images.copy_(images)
        if tubes is not None:
            flat, nums = _flat_tubes(tubes, images.device)
            assert nums == self.nums, "GraphedInference was captured for %s tubes per clip" % (self.nums,)
            self.flat0.copy_(flat)
        self.graph.replay()
        return self.out
