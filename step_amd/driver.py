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

from . import ops
from .tube_math import extrapolate_tubes, decode_coef, valid_tubes

TUBE_KERNEL = True             # per-step tube bookkeeping as one HIP launch (False: the tensor-op restatement below)
COMPACT_KERNEL = True          # postprocess: the detection rows of all iterations and clips compacted by one HIP launch (False: nonzero / gather / bincount)


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
    Returns (history, trajectory) in the reference's form: history[i] = {pred_prob [N,Tl,classes], pred_loc [N,Tl,4],
    pred_first_loc / pred_last_loc [N,T,4] (None unless temporal_mode == "predict"), tubes_nums}, detached;
    trajectory[i] = per clip (numpy proposals for the next step, class index [n,Tl])."""
    if getattr(args, "temporal_mode", "predict") not in ("predict", "extrapolate", "mean"):
        raise NotImplementedError("step_amd.driver.inference: temporal_mode %r" % (args.temporal_mode,))
    dev = conv_feat.device
    flat, nums = _flat_tubes(tubes, dev)
    clip_of = torch.as_tensor(np.repeat(np.arange(len(nums)), nums), device=dev)
    history, traj = inference_flat(args, conv_feat, context_feat, nets, exec_iter, flat, nums, clip_of)
    # the reference's trajectory (utils.py:87-124): per step, per clip (numpy proposals [n,T',4], class index [n,Tl])
    trajectory = []
    for (prop, cls), h in zip(traj, history):
        Tl = h["pred_loc"].shape[1]
        props = np.split(prop.detach().cpu().numpy(), np.cumsum(nums)[:-1], axis=0)
        classes = torch.split(cls.view(-1, 1).expand(-1, Tl), nums, dim=0)
        trajectory.append(list(zip(props, classes)))
    return history, trajectory


def inference_flat(args, conv_feat, context_feat, nets, exec_iter, flat, nums, clip_of, want_classes=True):
    """The device-resident core of `inference`: flat [N,T,5] tubes (col 0 = frame index), nums = tubes per
    clip, clip_of [N] = clip index of every tube.  Pure tensor ops with static shapes and no host
    synchronisation, so the whole multi-step pipeline can be captured in a hipGraph (GraphedInference).
    want_classes=False: the trajectory's class index (an argmax per step that only `inference` hands on) is None."""
    dev = conv_feat.device
    history, trajectory = [], []
    clip32 = None
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
        mode = getattr(args, "temporal_mode", "predict")
        extend = i < args.max_iter and args.NUM_CHUNKS[i + 1] == args.NUM_CHUNKS[i] + 2
        if mode == "predict" and TUBE_KERNEL and first_loc is not None and last_loc is not None and flat.shape[1] == T_length:
            # decode x3 -> cat -> valid_tubes -> frame-index column: one launch (step_tube_update) instead of ~60 element-wise ones
            if clip32 is None:
                clip32 = clip_of.to(torch.int32)
            pred_loc, pred_first, pred_last, flat = ops.tube_update(
                flat, local_loc, first_loc, last_loc, clip32, chunk_idx[0] - half_T, chunk_idx[-1] - half_T, extend,
                args.image_size[0], args.image_size[1])
            history.append({"pred_prob": pred_prob.detach(), "pred_loc": pred_loc, "pred_first_loc": pred_first,
                            "pred_last_loc": pred_last, "tubes_nums": list(nums)})
            trajectory.append((flat[:, :, 1:], torch.argmax(prob, dim=-1) if want_classes else None))
            continue
        flat = flat.to(local_loc)
        pred_loc = decode_coef(flat.reshape(-1, 5)[:, 1:], local_loc.reshape(-1, 4)).view(local_loc.shape)
        lo, hi = chunk_idx[0] - half_T, chunk_idx[0] + half_T + 1
        lo2, hi2 = chunk_idx[-1] - half_T, chunk_idx[-1] + half_T + 1
        pred_first = decode_coef(flat[:, lo:hi].reshape(-1, 5)[:, 1:], first_loc.reshape(-1, 4)).view(first_loc.shape)
        pred_last = decode_coef(flat[:, lo2:hi2].reshape(-1, 5)[:, 1:], last_loc.reshape(-1, 4)).view(last_loc.shape)
        history.append({"pred_prob": pred_prob.detach(), "pred_loc": pred_loc.detach(),            # utils.py:81-85 (.data)
                        "pred_first_loc": pred_first.detach() if mode == "predict" else None,
                        "pred_last_loc": pred_last.detach() if mode == "predict" else None, "tubes_nums": list(nums)})

        # next step's proposals (utils.py:91-129), all clips at once
        if extend:
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
        trajectory.append((prop, torch.argmax(prob, dim=-1) if want_classes else None))
        Tn = prop.shape[1]
        idx = (clip_of.to(prop.dtype) * Tn).view(-1, 1, 1) + torch.arange(Tn, device=dev, dtype=prop.dtype).view(1, Tn, 1)
        flat = torch.cat([idx, prop], dim=2)
    return history, trajectory


def _clip_groups(nums, device):
    """Index tables of the ragged (clip, tube) layout: gather index [B,kmax] into the flat tube axis (clamped) and the
    validity mask [B,kmax]; built from host ints without touching the device data."""
    B, kmax = len(nums), max(max(nums), 1)
    n = torch.as_tensor(nums, device=device, dtype=torch.int64)
    start = torch.cumsum(n, 0) - n
    j = torch.arange(kmax, device=device, dtype=torch.int64)
    valid = j.view(1, kmax) < n.view(B, 1)
    idx = (start.view(B, 1) + j.view(1, kmax)).clamp_(max=max(int(sum(nums)) - 1, 0))
    return idx, valid


_POST_CONST = {}


def postprocess(args, history, conf_thresh=None, nms_thresh=None, evaluate_topk=None, topk=None, iterations=None):
    """The evaluation loop of test.py:157-210 (the same code is inlined in train.py:512-573 and demo.py:123-174) as batched
    tensor operations: for EVERY refinement iteration and every clip, per class: mask the middle-frame scores with
    `score > conf_thresh`, valid_tubes() the middle-frame boxes (its 400 x 400 default, as the reference calls it), greedy
    NMS, normalise by the frame size; then the reference's row order -- classes ascending, kept tubes in ascending original
    order -- or, with evaluate_topk > 0, its stable ascending score sort, reversed, cut at `[:args.topk]` (including what that
    slice does for topk = -1).  The score mask, valid_tubes and the NMS of all (clip, class) groups of an iteration are ONE launch
    (step_detect_nms; the reference makes 60 x B serial CPU calls), there is no Python loop over classes or boxes, and ONE host
    synchronisation for all iterations fetches the row counts for the per-clip split.

    Thresholds default to args.conf_thresh / nms_thresh / evaluate_topk / topk (config.py:62-65).
    Returns a list over iterations of lists over clips of dicts {boxes [m,4] fp32 normalised, scores [m], labels [m] (class
    index), tubes [m] (index of the tube inside its clip)} in the order the reference writes its CSV rows."""
    from . import ops

    conf = float(getattr(args, "conf_thresh", 0.01) if conf_thresh is None else conf_thresh)
    thr = float(getattr(args, "nms_thresh", 0.4) if nms_thresh is None else nms_thresh)
    etopk = int(getattr(args, "evaluate_topk", -1) if evaluate_topk is None else evaluate_topk)
    topk = int(getattr(args, "topk", -1) if topk is None else topk)
    W, H = float(args.image_size[0]), float(args.image_size[1])
    todo = [(it, h) for it, h in enumerate(history) if iterations is None or it in iterations]
    out = [None] * len(todo)
    # iterations that share a clip layout (they all do in inference()) go through ONE pass: a fused mask + clamp + NMS launch each
    # (ops.detect_nms), then one nonzero over all of them and ONE host synchronisation for the row counts
    fast, slow = [], []
    for oi, (it, h) in enumerate(todo):
        nums = [int(v) for v in h["tubes_nums"]]
        if len(nums) == 0 or sum(nums) == 0:
            e = torch.zeros(0, device=h["pred_loc"].device)
            out[oi] = [{"boxes": e.view(0, 4), "scores": e, "labels": e.long(), "tubes": e.long()} for _ in nums]
        elif max(nums) <= 64:
            fast.append((oi, h, nums))
        else:
            slow.append((oi, h, nums))
    groups = {}
    for oi, h, nums in fast:
        groups.setdefault((tuple(nums), h["pred_prob"].shape[-1], h["pred_loc"].device), []).append((oi, h))
    for (nums, NC, dev), members in groups.items():
        B, kmax, I = len(nums), max(nums), len(members)
        # per-layout constants live on the device (a host -> device copy from pageable memory per call stalls the host; the layout of a
        # serving loop never changes)
        ck = (nums, dev, W, H)
        hit = _POST_CONST.get(ck)
        if hit is None:
            n = torch.as_tensor(nums, device=dev, dtype=torch.int32)
            hit = _POST_CONST[ck] = (n, (torch.cumsum(n, 0) - n).to(torch.int32), torch.tensor([W, H, W, H], device=dev))
            if len(_POST_CONST) > 64:
                _POST_CONST.pop(next(iter(_POST_CONST)))
        n, start, whwh = hit
        keep = torch.empty((I, B, NC, kmax), dtype=torch.uint8, device=dev)
        sc_all, bx_all = [], []
        for k, (oi, h) in enumerate(members):
            prob, loc = h["pred_prob"], h["pred_loc"]
            scores = prob[:, int(prob.shape[1] / 2)].float()                                          # [N,NC] middle frame (test.py:159)
            _, boxes = ops.detect_nms(scores, loc[:, int(loc.shape[1] / 2)].float(), start, n, kmax, conf, thr, 400.0, 400.0, keep[k])
            sc_all.append(scores)                                                                     # (valid_tubes' 400 x 400 default, as the reference calls it: test.py:161,191)
            bx_all.append(boxes)
        # rows in the reference's order: iteration, clip, class ascending, kept tube ascending == row-major order of `keep`
        if COMPACT_KERNEL and I <= 8:
            # one launch (step_detect_compact: a ballot prefix per (iteration, clip) into fixed-capacity segments) and one small copy of
            # the row counts -- before: nonzero + index + cat + gather + bincount, ~25 launches (~100 us of the GPU per step at 4 x 34
            # tubes) and two host synchronisations
            rb_, rs_, kc_, kj_, cnt = ops.detect_compact(keep, bx_all, sc_all, start, W, H)
            per = cnt.tolist()                                                                        # the one host sync
            cap = NC * kmax
            pieces = [(rb_[g * cap:g * cap + per[g]], rs_[g * cap:g * cap + per[g]], kc_[g * cap:g * cap + per[g]], kj_[g * cap:g * cap + per[g]])
                      for g in range(I * B)]
        else:
            ki, kb, kc, kj = torch.nonzero(keep, as_tuple=True)
            tube = start.long()[kb] + kj + ki * sum(nums)
            rb = torch.cat(bx_all)[tube] / whwh                                                       # test.py:197-198
            rs = torch.cat(sc_all)[tube, kc]
            per = torch.bincount(ki * B + kb, minlength=I * B).tolist()                               # the one host sync
            pieces = list(zip(rb.split(per), rs.split(per), kc.split(per), kj.split(per)))
        for k, (oi, h) in enumerate(members):
            clips = []
            for bx, sc, cl, tb in pieces[k * B:(k + 1) * B]:
                if etopk > 0:                                                                         # test.py:205-208
                    # list.sort(key=score) is stable and ascending, then reversed: descending with ties in REVERSED row order
                    sel = torch.flip(torch.argsort(sc, stable=True), dims=(0,))[:topk]
                    bx, sc, cl, tb = bx[sel], sc[sel], cl[sel], tb[sel]
                clips.append({"boxes": bx, "scores": sc, "labels": cl, "tubes": tb})
            out[oi] = clips
    for oi, h, nums in slow:
        out[oi] = _postprocess_general(h, nums, conf, thr, etopk, topk, W, H)
    return out


def _postprocess_general(h, nums, conf, thr, etopk, topk, W, H):
    """One iteration with tensor operations + step_nms_batched: more than 64 tubes per clip (anchor modes 3 / 4)."""
    from .roi_layers import nms_batched
    prob, loc = h["pred_prob"], h["pred_loc"]
    dev = loc.device
    B, NC = len(nums), prob.shape[-1]
    boxes = valid_tubes(loc[:, int(loc.shape[1] / 2)].float().unsqueeze(1))[:, 0]                     # [N,4] (test.py:161,191)
    scores = prob[:, int(prob.shape[1] / 2)].float()                                                  # [N,NC] (:159)
    idx, valid = _clip_groups(nums, dev)
    kmax = idx.shape[1]
    gs = scores[idx].permute(0, 2, 1)                                                                 # [B,NC,kmax]
    mask = (gs > conf) & valid.view(B, 1, kmax)                                                       # :180
    # the reference compacts the masked boxes before nms (order kept): move them to the front of each group
    order = torch.argsort((~mask).to(torch.int8), dim=2, stable=True)
    gs = torch.gather(gs, 2, order)
    gb = boxes[idx].view(B, 1, kmax, 4).expand(B, NC, kmax, 4)
    gb = torch.gather(gb, 2, order.unsqueeze(-1).expand(B, NC, kmax, 4)).contiguous()
    counts = mask.sum(2).to(torch.int32)
    keep = nms_batched(gb.view(B * NC, kmax, 4), gs.reshape(B * NC, kmax), counts.view(-1), thr).view(B, NC, kmax).bool()
    kb, kc, kj = torch.nonzero(keep, as_tuple=True)
    rb = gb[kb, kc, kj] / torch.tensor([W, H, W, H], device=dev)                                      # :197-198
    rs = gs[kb, kc, kj]
    rt = order[kb, kc, kj]
    per_clip = torch.bincount(kb, minlength=B).tolist()
    clips = []
    for bx, sc, cl, tb in zip(rb.split(per_clip), rs.split(per_clip), kc.split(per_clip), rt.split(per_clip)):
        if etopk > 0:
            sel = torch.flip(torch.argsort(sc, stable=True), dims=(0,))[:topk]
            bx, sc, cl, tb = bx[sel], sc[sel], cl[sel], tb[sel]
        clips.append({"boxes": bx, "scores": sc, "labels": cl, "tubes": tb})
    return clips


def detections_csv(dets, infos, label_dict=None):
    """The text test.py:210-218 writes for one iteration: `dets` = one entry of postprocess()'s result, infos = per clip
    {'video_name', 'fid'}; label_dict maps class index -> label id (identity + 1 when None)."""
    lines = []
    for d, info in zip(dets, infos):
        bx, sc, cl = d["boxes"].cpu().numpy(), d["scores"].cpu().numpy(), d["labels"].cpu().numpy()
        for k in range(len(sc)):
            lab = int(cl[k]) + 1 if label_dict is None else label_dict[int(cl[k])]
            lines.append("{0},{1:04},{2:.4},{3:.4},{4:.4},{5:.4},{6},{7:.4}\n".format(
                info["video_name"], info["fid"], bx[k, 0], bx[k, 1], bx[k, 2], bx[k, 3], lab, sc[k]))
    return lines


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
        hist, _ = inference_flat(self.args, cf, cx, self.nets, self.args.max_iter, self.flat0, self.nums, self.clip_of, want_classes=False)
        return hist, cf, cx

    def __call__(self, images=None, tubes=None):
        # images: a new clip batch, copied into the captured input buffer -- or None / `self.images` itself when the caller (a decoder, the
        # uint8 -> 16-bit conversion, bench.py's resident-input loop) has written the batch there already: at C3 size the copy is 138 MB
        # read + 138 MB written per step, ~1 % of it
        if images is not None and images.data_ptr() != self.images.data_ptr():
            self.images.copy_(images)
        if tubes is not None:
            flat, nums = _flat_tubes(tubes, self.images.device)
            assert nums == self.nums, "GraphedInference was captured for %s tubes per clip" % (self.nums,)
            self.flat0.copy_(flat)
        self.graph.replay()
        return self.out
