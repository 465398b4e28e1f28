"""step_amd/selection.py -- training sample selection between the steps of the progressive head.

Host-side counterpart of the reference's `train_select` / `select_proposals` (utils/utils.py:135-423) and
`compute_tube_iou` / `compute_box_iou` (utils/tube_utils.py:269-351).  The reference walks ground truths x proposals
x frames in interpreted loops; here the IoU table and the target assembly are array operations.  Two things are kept
exactly, because they decide WHICH tubes are trained on:

* the fp32 arithmetic of the IoU (same operations in the same order; the per-tube mean is taken in double and rounded to
  fp32 once, as the reference's Python-float accumulation does), and
* the order of the draws from the global `random` / `numpy.random` streams (one optional `random.shuffle`, then at most
  two `np.random.choice(..., replace=False)`), so that with the same seeds the same proposals are selected
  (tests/golden/selection_golden.npz was recorded from the reference).

All three `temporal_mode`s ("predict": every shipped script; "extrapolate"; "mean") are supported, like step_amd/driver.py.
"""
import random

import numpy as np

from .tube_math import extrapolate_tubes, valid_tubes


def box_iou(a, b):
    """IoU table [len(a), len(b)] of boxes [x1,y1,x2,y2] without the +1 pixel convention (tube_utils.py:269-306):
    zero unless both overlap extents are positive; degenerate pairs divide by zero like the reference (nan / inf)."""
    a = np.asarray(a, np.float32).reshape(-1, 4)[:, None, :]
    b = np.asarray(b, np.float32).reshape(-1, 4)[None, :, :]
    iw = np.maximum(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]), np.float32(0))
    ih = np.maximum(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]), np.float32(0))
    inter = np.where((iw > 0) & (ih > 0), iw * ih, np.float32(0))
    union = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(np.float32)


def tube_iou(t1, t2):
    """[n1, n2] mean over the T frames of the per-frame box IoU; a pair contributes zeros when either whole tube sums
    to zero (the padding convention of tube_utils.py:308-351)."""
    t1 = np.asarray(t1, np.float32)
    t2 = np.asarray(t2, np.float32)
    if t1.ndim < 3:
        t1 = t1.reshape(1, -1, 4)
    if t2.ndim < 3:
        t2 = t2.reshape(1, -1, 4)
    if t1.shape[1] != t2.shape[1]:
        raise AssertionError("Tube with different length!")
    T = t1.shape[1]
    live = np.array([bool(np.sum(t)) for t in t1])[:, None] & np.array([bool(np.sum(t)) for t in t2])[None, :]
    acc = np.zeros((t1.shape[0], t2.shape[0]), np.float64)
    for t in range(T):                                              # frame by frame: the double sum runs in the same order
        acc += np.where(live, box_iou(t1[:, t], t2[:, t]).astype(np.float64), 0.0)
    if T > 0:
        acc /= T
    return acc.astype(np.float32)


def select_proposals(gt_tubes, anchors, scores=None, cls_thresh=0.2, max_pos_num=5, sampling="random", neg_ratio=2, ious=None):
    """-> (positives [(gt, proposal)], negatives [(gt, proposal)], iou table) -- utils/utils.py:341-423.
    Positives: the best free proposal of every ground truth (highest-IoU ground truth first), then random ones among the
    proposals above cls_thresh; negatives: drawn from the rest, uniformly / by score / by softmax(score)."""
    if ious is None:                                                # (the device front end hands the table in: step_select_prepare)
        ious = tube_iou(np.asarray(gt_tubes)[:, :, :4], anchors)
    G, A = ious.shape
    if scores is None:
        scores = ious.max(axis=0)
    taken = set()
    pos = []
    left = ious.copy()
    for _ in range(G):
        g = int(np.argmax(left.max(axis=1)))
        for a in np.argsort(ious[g, :])[::-1]:
            if int(a) not in taken:
                taken.add(int(a))
                pos.append((g, int(a)))
                left[g, :] = -1
                break
    if len(pos) > max_pos_num:
        random.shuffle(pos)
        pos = pos[:max_pos_num]
    above = [int(a) for a in np.where(np.sum(ious > cls_thresh, axis=0))[0] if int(a) not in taken]
    if above and len(pos) < max_pos_num:
        owner = np.argmax(ious[:, above], axis=0)
        draw = np.random.choice(len(above), min(len(above), max_pos_num - len(pos)), p=np.ones((len(above),)) / len(above),
                                replace=False)
        for d in draw:
            taken.add(above[d])
            pos.append((int(owner[d]), above[d]))
            if len(pos) == max_pos_num:
                break
    pos = pos[:max_pos_num]
    taken.update(above)                                             # never a negative: they overlap some ground truth
    rest = [a for a in range(A) if a not in taken]
    neg = []
    if rest:
        w = np.asarray(scores)[rest]
        if sampling == "uniform":
            w = (w + 1e-6) / np.sum(w + 1e-6)
        elif sampling == "random":
            w = np.ones((len(rest),)) / len(rest)
        elif sampling == "softmax":
            w = np.exp(w) / np.sum(np.exp(w))
        else:
            raise NotImplementedError(sampling)
        for d in np.random.choice(len(rest), min(len(pos) * neg_ratio, len(rest)), p=w, replace=False):
            neg.append((int(np.argmax(ious[:, rest[d]])), rest[d]))
    if neg_ratio > 0:
        pos = pos[:max(max_pos_num, int(len(neg) / neg_ratio))]
    return pos, neg, ious


def _top_candidates(prob, topk, num_classes):
    """Rows of the clip's predictions to keep, best first, with their scores: per class the best 2*topk/num_classes tubes
    (all when topk <= 0), merged by score, one entry per tube (utils/utils.py:179-214)."""
    per_cls_idx, per_cls_score = [], []
    keep = int(topk / num_classes) * 2 if topk > 0 else None
    for c in range(num_classes):
        s = prob[:, c].reshape(-1)
        order = np.argsort(s)[::-1]
        per_cls_idx.append(order[:keep])
        per_cls_score.append(s[order][:keep])
    flat_s = np.concatenate(per_cls_score)
    flat_i = np.concatenate(per_cls_idx)
    # ascending stable sort, reversed: what list.sort(key=score)[::-1] yields, ties included
    seen, rows, sc = set(), [], []
    for k in np.argsort(flat_s, kind="stable")[::-1]:
        i = int(flat_i[k])
        if i not in seen:
            seen.add(i)
            rows.append(i)
            sc.append(flat_s[k])
    if topk > 0:
        rows, sc = rows[:topk], sc[:topk]
    return np.asarray(rows, np.int64), np.asarray(sc)


def train_select(step, history, targets, tubes, args, device=None):
    """-> (selected_tubes, target_tubes), one array per clip -- utils/utils.py:135-339.

    step 1 trains on the initial proposals `tubes[b]`; later steps on the best-scoring refined tubes of the previous step
    (`history`: pred_prob [N,T,C], pred_loc [N,T,4], pred_first_loc / pred_last_loc [N,T,4], tubes_nums).  Every selected tube
    comes with one target row per loss frame [first neighbour, centre, last neighbour], each
    [x1,y1,x2,y2, cls flag, reg flag, class labels...].

    device=True (default when the history lives on a ROCm device): the per-tube arithmetic -- class scores averaged over the frames,
    valid_tubes of the three predicted tubes, the IoU table against the clip's ground truths -- is ONE launch (step_select_prepare) and
    ONE device-to-host copy of its small results instead of four copies of the raw predictions and numpy passes over them; the
    sorts and the draws from the random streams, which decide WHICH tubes are trained on, stay here.  Same selections, bit for bit."""
    if args.temporal_mode not in ("predict", "extrapolate", "mean"):
        raise NotImplementedError("temporal_mode %r" % (args.temporal_mode,))
    chunks, max_chunks = args.NUM_CHUNKS[step], args.NUM_CHUNKS[args.max_iter]
    T = args.T
    t_start = int((max_chunks - chunks) / 2) * T
    t_len = chunks * T
    mid = int(max_chunks / 2)
    cls_thresh, reg_thresh = args.cls_thresh[step - 1], args.reg_thresh[step - 1]
    W, H = args.image_size[0], args.image_size[1]
    nc = args.num_classes
    predict = args.temporal_mode == "predict"
    grow = (step - 1) in args.NUM_CHUNKS and args.NUM_CHUNKS[step] == args.NUM_CHUNKS[step - 1] + 2
    grows_next = predict and step < args.max_iter and args.NUM_CHUNKS[step + 1] == args.NUM_CHUNKS[step] + 2

    def host(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)

    dev_path = None
    if step > 1:
        bounds = np.concatenate(([0], np.cumsum(history["tubes_nums"]))).astype(np.int64)
        on_dev = hasattr(history["pred_prob"], "is_cuda") and history["pred_prob"].is_cuda
        if device if device is not None else on_dev:
            dev_path = _prepare_on_device(history, targets, mid, predict, W, H)
        else:
            prob, loc = host(history["pred_prob"]), host(history["pred_loc"])
            first = host(history["pred_first_loc"]) if predict else None
            last = host(history["pred_last_loc"]) if predict else None

    selected, wanted = [], []
    for b in range(len(targets)):
        gt = np.asarray(targets[b])
        if step == 1:
            cand, cand_score, cand_first, cand_last = np.asarray(tubes[b]), None, None, None
        else:
            lo, hi = bounds[b], bounds[b + 1]
            if dev_path is not None:
                mean_prob, vloc, vfirst, vlast, iou_all = dev_path
                rows, cand_score = _top_candidates(mean_prob[lo:hi], args.topk, nc)
                cand = vloc[lo:hi][rows]
                cand_first = vfirst[lo:hi][rows] if predict else None
                cand_last = vlast[lo:hi][rows] if predict else None
                table = np.ascontiguousarray(iou_all[lo:hi][rows][:, :gt.shape[0]].T)            # [G, candidates]
            else:
                rows, cand_score = _top_candidates(prob[lo:hi].mean(axis=1), args.topk, nc)
                cand = valid_tubes(loc[lo:hi][rows], W, H)
                cand_first = valid_tubes(first[lo:hi][rows], W, H) if predict else None
                cand_last = valid_tubes(last[lo:hi][rows], W, H) if predict else None
        pos, neg, ious = select_proposals(gt[:, mid].reshape(gt.shape[0], 1, -1), cand[:, int(cand.shape[1] / 2)].reshape(cand.shape[0], 1, -1),
                                          cand_score, cls_thresh, args.max_pos_num, args.selection_sampling, args.neg_ratio,
                                          ious=table if (step > 1 and dev_path is not None) else None)
        pg = np.asarray([g for g, _ in pos], np.int64)
        pa = np.asarray([a for _, a in pos], np.int64)
        ng = np.asarray([g for g, _ in neg], np.int64)
        na = np.asarray([a for _, a in neg], np.int64)
        rows_a = np.concatenate((pa, na))
        R, P = len(rows_a), len(pa)
        sel = cand[rows_a].astype(np.float32).reshape(R, cand.shape[1], 4)
        centre = np.zeros((R, 1, 6 + nc), np.float32)
        centre[:P, 0, :4] = gt[pg, mid, :4]
        centre[:P, 0, 6:] = gt[pg, mid, 4:]
        centre[:P, 0, 4:6] = 1                                      # positives: classification and regression
        reg_only = ious[ng, na] >= reg_thresh if len(ng) else np.zeros((0,), bool)
        centre[P:, 0, :4][reg_only] = gt[ng[reg_only], mid, :4]     # negatives close enough to a ground truth still regress
        centre[P:, 0, 6:][reg_only] = gt[ng[reg_only], mid, 4:]
        centre[P:, 0, 5][reg_only] = 1
        if grow:
            if predict:
                sel = np.concatenate((cand_first[rows_a].astype(np.float32).reshape(R, T, 4), sel,
                                      cand_last[rows_a].astype(np.float32).reshape(R, T, 4)), axis=1)
            elif args.temporal_mode == "extrapolate":
                sel = extrapolate_tubes(sel, T)
            else:
                m = np.tile(np.mean(sel, axis=1, keepdims=True), (1, T, 1))
                sel = np.concatenate((m, sel, m), axis=1)
        before = np.zeros((R, 1, 6 + nc), np.float32)
        after = np.zeros((R, 1, 6 + nc), np.float32)
        if grows_next and P:
            for dst, frame in ((before, int((t_start - T) / T)), (after, int((t_start + t_len) / T))):
                dst[:P, 0, :4] = gt[pg, frame, :4]
                dst[:P, 0, 5] = dst[:P, 0, :4].sum(axis=1) > 0       # an all-zero box is padding: no regression target
                dst[:P, 0, 6:] = gt[pg, frame, 4:]
        selected.append(sel)
        wanted.append(np.concatenate((before, centre, after), axis=1))
    return selected, wanted


def _prepare_on_device(history, targets, mid, predict, W, H):
    """step_select_prepare over a step's predictions (device tensors) -> host arrays (mean_prob [N,NC], vloc [N,T,4], vfirst / vlast
    [N,Tw,4] | None, iou [N,Gmax]) through one packed device-to-host copy."""
    import torch

    from . import ops
    prob = history["pred_prob"]
    dev = prob.device
    nums = [int(v) for v in history["tubes_nums"]]
    B = len(nums)
    Gmax = max([np.asarray(t).shape[0] for t in targets] + [1])
    gt = np.zeros((B, Gmax, 4), np.float32)
    cnt = np.zeros((B,), np.int32)
    for b, t in enumerate(targets):
        t = np.asarray(t)
        cnt[b] = t.shape[0]
        gt[b, :t.shape[0]] = t[:, mid, :4]
    clip_of = torch.as_tensor(np.repeat(np.arange(B), nums).astype(np.int32), device=dev)
    outs = ops.select_prepare(prob, history["pred_loc"], history["pred_first_loc"] if predict else None,
                              history["pred_last_loc"] if predict else None, clip_of, torch.from_numpy(gt).to(dev), torch.from_numpy(cnt).to(dev),
                              float(W), float(H))
    sizes = [o.numel() for o in outs if o is not None]
    packed = torch.cat([o.reshape(-1) for o in outs if o is not None]).cpu().numpy()           # the one device-to-host copy
    res, k, off = [], 0, 0
    for o in outs:
        if o is None:
            res.append(None)
            continue
        res.append(packed[off:off + sizes[k]].reshape(tuple(o.shape)))
        off += sizes[k]
        k += 1
    return res
