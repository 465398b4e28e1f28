"""oracle/make_golden.py -- generate tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs only where /root/reference exists (the build container).  It imports the reference's Python
(models/, utils/) from where it lies, with
  * empty stand-in modules for `torchvision` and `cv2` (imported by models/networks.py:11 and
    data/data_utils.py:9 but never used on this path), and
  * `external.maskrcnn_benchmark.roi_layers._C` = oracle/_ref/_C.so, the reference's own C++ CPU
    operators compiled from its sources (make -C oracle ref),
fills every parameter/buffer with the closed-form filler of oracle/i3d_ref.py (a pure function
of the state_dict key and the flat index -- no RNG stream, no weights shipped), runs the reference
and stores inputs that cannot be regenerated plus the expected outputs.

The fixtures are data only (inputs and outputs); no reference source text is stored.

    python -m oracle.make_golden            # from the repo root
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("STEP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    from oracle import load_reference_C

    sys.path.insert(0, REF)
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.modules["external.maskrcnn_benchmark.roi_layers._C"] = load_reference_C()
    import models  # noqa: F401  (reference)
    import utils.utils as ref_utils  # reference

    # On CPU tensors `pred_loc.cpu().numpy()` (utils/utils.py:107-120) ALIASES the tensor stored in
    # `history`, so the in-place clamping of valid_tubes (utils/tube_utils.py:73-92) would overwrite the
    # recorded predictions -- an artefact of running the reference on CPU; on its real (GPU) path
    # `.cpu()` copies.  Give valid_tubes a copy so that the fixtures record what the GPU path records.
    _vt = ref_utils.valid_tubes
    ref_utils.valid_tubes = lambda tubes, *a, **kw: _vt(tubes.copy(), *a, **kw)
    from external.maskrcnn_benchmark.roi_layers import nms, ROIAlign  # reference python API
    return models, ref_utils, nms, ROIAlign


def cfg(**kw):
    base = dict(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False, T=3,
                num_classes=60, fc_dim=256, dropout=0.0, pool_size=7, no_context=False, max_iter=3,
                NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3}, temporal_mode="predict", image_size=(400, 400),
                pool_mode="align")
    base.update(kw)
    return types.SimpleNamespace(**base)


def fill_module(mod, tag=""):
    from oracle.i3d_ref import fill_state_dict

    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict(fill_state_dict(shapes, tag))
    return shapes


def stage_digest(t):
    """Small digest of a big activation: stats + a strided 256-element sample."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // 256)
    return {"mean": float(f.double().mean()), "absmax": float(f.abs().max()), "l2": float(f.double().norm()),
            "step": step, "sample": f[::step][:256].numpy().copy()}


def main():
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    models, ref_utils, ref_nms, RefROIAlign = import_reference()
    from oracle import i3d_ref as R
    C = sys.modules["external.maskrcnn_benchmark.roi_layers._C"]
    keyinfo = {}

    # ---------------------------------------------------------------- 1. ROIAlign fwd + NMS (reference C++)
    g = {}
    feat = R.fill_tensor("golden.roi.feat", (3, 6, 25, 25), "image")
    rois = np.array([
        [0, 0, 0, 400, 400],            # whole frame -> 4x4 samples per bin
        [1, 100.3, 57.9, 101.0, 58.2],  # < 1 px -> forced 1x1
        [2, 310.5, 200.25, 400, 399.5], # touches x=400
        [0, 13.7, 91.2, 250.9, 333.3],  # fractional
        [1.0, -40, -30, 90, 120],       # partly outside (negative)
        [2.0, 350, 350, 470, 520],      # partly outside (beyond)
        [0, -200, -200, -100, -100],    # fully outside: every sample void
        [1, 200, 100, 100, 50],         # malformed (x2<x1): max(size,1)
        [2, 0, 0, 15.99, 15.99],        # first cell only
        [0, 384, 384, 400, 400],        # last cell, clamps y_low>=H-1
    ], dtype=np.float32)
    g["align_rois"] = rois
    for tag, pooled, sr in (("p7s0", (7, 7), 0), ("p7s2", (7, 7), 2), ("p3x5s0", (3, 5), 0)):
        out = C.roi_align_forward(feat, torch.from_numpy(rois), 1.0 / 16.0, pooled[0], pooled[1], sr)
        g["align_out_" + tag] = out.numpy()
    out0 = C.roi_align_forward(feat, torch.zeros((0, 5)), 1.0 / 16.0, 7, 7, 0)
    assert out0.shape == (0, 6, 7, 7)
    # the python-level reference API on a tube-shaped call (networks.py:44)
    rs = np.random.RandomState(7)
    conv = R.fill_tensor("golden.roi.conv", (2, 3, 16, 25, 25), "feat")
    tubes = []
    for b in range(2):
        for _ in range(4):
            x1, y1 = rs.uniform(0, 300, 2)
            w, h = rs.uniform(20, 99, 2)
            for t in range(3):
                dx, dy = rs.uniform(-6, 6, 2)
                tubes.append([b * 3 + t, x1 + dx, y1 + dy, x1 + dx + w, y1 + dy + h])
    tubes = np.asarray(tubes, np.float32).reshape(8, 3, 5)
    g["tube_rois"] = tubes
    g["tube_out"] = RefROIAlign((7, 7), 1.0 / 16.0, 0)(conv.view(-1, 16, 25, 25), torch.from_numpy(tubes).view(-1, 5)).numpy()

    def rand_boxes(n, rs, span=400.0):
        xy = rs.uniform(0, span * 0.7, (n, 2))
        wh = rs.uniform(8, span * 0.5, (n, 2))
        return np.concatenate([xy, np.minimum(xy + wh, span)], 1).astype(np.float32)

    nms_cases = []
    for n, thr in ((34, 0.4), (11, 0.4), (200, 0.5), (1, 0.4), (64, 0.3), (65, 0.7), (129, 0.4)):
        b = rand_boxes(n, rs)
        s = rs.permutation(n).astype(np.float32) / n + 0.001      # tie-free
        nms_cases.append((b, s, thr))
    # clustered boxes (heavy suppression)
    base = rand_boxes(6, rs)
    b = np.repeat(base, 10, 0) + rs.uniform(-6, 6, (60, 4)).astype(np.float32)
    nms_cases.append((b.astype(np.float32), (rs.permutation(60).astype(np.float32) + 1) / 61, 0.4))
    # threshold equality: IoU of these two boxes is exactly 0.5 (areas 100 and 50, inter 50):  >= suppresses
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [50, 50, 59, 59]], np.float32)
    nms_cases.append((b, np.array([0.9, 0.8, 0.7], np.float32), 0.5))
    # identical boxes, distinct scores
    b = np.tile(np.array([[10, 10, 50, 60]], np.float32), (5, 1))
    nms_cases.append((b, np.array([0.1, 0.5, 0.3, 0.9, 0.2], np.float32), 0.4))
    for i, (b, s, thr) in enumerate(nms_cases):
        keep = ref_nms(torch.from_numpy(b), torch.from_numpy(s), thr)
        assert keep.dtype == torch.int64
        g["nms%d_boxes" % i], g["nms%d_scores" % i] = b, s
        g["nms%d_thr" % i] = np.float32(thr)
        g["nms%d_keep" % i] = keep.numpy()
    g["nms_count"] = np.int64(len(nms_cases))
    assert ref_nms(torch.zeros((0, 4)), torch.zeros((0,)), 0.4).numel() == 0
    np.savez_compressed(os.path.join(OUT, "roi_nms_golden.npz"), **g)
    print("roi_nms_golden: %d arrays" % len(g))

    # ---------------------------------------------------------------- 2. C1: BaseNet on [1,8,3,112,112]
    base = models.BaseNet(cfg())
    keyinfo["BaseNet"] = {k: list(v) for k, v in fill_module(base).items()}
    keyinfo["BaseNet_trainable"] = [k for k, p in base.named_parameters() if p.requires_grad]
    base.eval()
    g = {}
    x = R.fill_tensor("golden.c1.images", (1, 8, 3, 112, 112), "image")
    stages = []
    hooks = [m.register_forward_hook(lambda _m, _i, o: stages.append(o)) for m in base.base_model]
    with torch.no_grad():
        y = base(x)
    for h in hooks:
        h.remove()
    assert tuple(y.shape) == (1, 2, 832, 7, 7)
    g["conv_feat"] = y.contiguous().numpy()
    for i, s in enumerate(stages):
        d = stage_digest(s)
        g["stage%d_shape" % i] = np.asarray(s.shape)
        g["stage%d_stats" % i] = np.asarray([d["mean"], d["absmax"], d["l2"], d["step"]], np.float64)
        g["stage%d_sample" % i] = d["sample"]
    np.savez_compressed(os.path.join(OUT, "i3d_c1_golden.npz"), **g)
    print("i3d_c1_golden ok, absmax out %.4f" % float(y.abs().max()))

    # ---------------------------------------------------------------- 3. single ops: pools, one Mixed, one unit
    from models.i3dpt import MaxPool3dTFPadding, Mixed, Unit3Dpy
    g = {}
    xin = R.fill_tensor("golden.pool.in", (2, 5, 6, 9, 11), "image")      # has negative values
    for tag, k, s in (("k133s122", (1, 3, 3), (1, 2, 2)), ("k333s222", (3, 3, 3), (2, 2, 2)),
                      ("k333s111", (3, 3, 3), (1, 1, 1)), ("k222s222", (2, 2, 2), (2, 2, 2))):
        g["pool_" + tag] = MaxPool3dTFPadding(k, s)(xin).numpy()
    g["pool_allneg"] = MaxPool3dTFPadding((3, 3, 3), (2, 2, 2))(-torch.ones(1, 1, 4, 5, 5)).numpy()
    mx = Mixed(24, [8, 12, 16, 4, 8, 8])
    keyinfo["Mixed_small"] = {k: list(v) for k, v in fill_module(mx, "golden.mixed.").items()}
    mx.eval()
    xm = R.fill_tensor("golden.mixed.in", (2, 24, 3, 9, 7), "feat")
    with torch.no_grad():
        g["mixed_out"] = mx(xm).numpy()
    for tag, ci, co, k, s, shp in (("stem", 3, 16, (7, 7, 7), (2, 2, 2), (1, 3, 9, 21, 19)),
                                   ("k3", 20, 24, (3, 3, 3), (1, 1, 1), (2, 20, 3, 6, 7)),
                                   ("k1", 20, 12, (1, 1, 1), (1, 1, 1), (2, 20, 3, 6, 7))):
        u = Unit3Dpy(ci, co, kernel_size=k, stride=s)
        fill_module(u, "golden.unit." + tag + ".")
        u.eval()
        with torch.no_grad():
            g["unit_%s_out" % tag] = u(R.fill_tensor("golden.unit.%s.in" % tag, shp, "image")).numpy()
    np.savez_compressed(os.path.join(OUT, "ops_golden.npz"), **g)
    print("ops_golden ok")

    # ---------------------------------------------------------------- 4. heads: ContextNet, TwoBranchNet (+losses)
    g = {}
    ctx = models.ContextNet(cfg())
    keyinfo["ContextNet"] = {k: list(v) for k, v in fill_module(ctx).items()}
    ctx.eval()
    cf = R.fill_tensor("golden.ctx.feat", (1, 3, 832, 25, 25), "feat")
    with torch.no_grad():
        g["context_out"] = ctx(cf).numpy()
    det = models.TwoBranchNet(cfg())
    keyinfo["TwoBranchNet"] = {k: list(v) for k, v in fill_module(det, "det0.").items()}
    keyinfo["TwoBranchNet_trainable"] = [k for k, p in det.named_parameters() if p.requires_grad]
    det_cls = models.TwoBranchNet(cfg(), cls_only=True)
    keyinfo["TwoBranchNet_cls_only"] = {k: list(v.shape) for k, v in det_cls.state_dict().items()}
    det.set_device("cpu")
    det.eval()
    for tl in (3, 9):
        pf = R.fill_tensor("golden.det.pooled%d" % tl, (2, tl, 832, 7, 7), "feat")
        cx = R.fill_tensor("golden.det.ctx%d" % tl, (2, 1024, tl, 1, 1), "feat")
        with torch.no_grad():
            o = det(pf, context_feat=cx)
        for nme, t in zip(("prob", "loc", "first", "last"), o[:4]):
            g["det_T%d_%s" % (tl, nme)] = t.numpy()
    # losses, Tl = 3
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    tubes = np.zeros((2, 3, 5), np.float32)
    tubes[0, :, 1:] = [60, 80, 220, 300]
    tubes[1, :, 1:] = [150, 40, 330, 280]
    tubes[:, :, 0] = np.arange(3)[None]
    targets = np.zeros((2, 3, 66), np.float32)
    targets[0, :, :4] = [[70, 85, 215, 310], [72, 86, 230, 305], [75, 90, 226, 300]]
    targets[1, :, :4] = [[140, 50, 320, 270], [150, 45, 335, 290], [149, 38, 340, 284]]
    targets[:, :, 4] = [[1, 1, 1], [0, 0, 0]]      # cls indicator (second tube = background)
    targets[:, :, 5] = [[1, 1, 0], [1, 1, 1]]      # regression indicator
    targets[0, :, 6 + 3] = 1
    targets[0, :, 6 + 17] = 1
    targets[1, :, 6 + 40] = 1
    g["loss_tubes"], g["loss_targets"] = tubes, targets
    with torch.no_grad():
        o = det(pf, context_feat=cx, tubes=torch.from_numpy(tubes), targets=torch.from_numpy(targets))
    g["loss_cls"], g["loss_loc"], g["loss_nbr"] = o[4].numpy(), o[5].numpy(), o[6].numpy()
    np.savez_compressed(os.path.join(OUT, "head_golden.npz"), **g)
    print("head_golden ok; losses", o[4].mean().item(), o[5].item(), o[6].item())

    # ---------------------------------------------------------------- 5. reference inference() on a synthetic conv_feat
    g = {}
    args = cfg()
    nets = {"roi_net": models.ROINet("align", 7)}
    for i in range(3):
        d = models.TwoBranchNet(args)
        fill_module(d, "det%d." % i)
        d.set_device("cpu")
        d.eval()
        nets["det_net%d" % i] = d
    anchors = R.anchors()
    assert anchors.shape == (34, 4)
    keyinfo["anchors34"] = anchors.tolist()
    for ntubes in (11, 34):
        conv_feat = R.fill_tensor("golden.inf.feat", (2, 9, 832, 25, 25), "feat")
        with torch.no_grad():
            context = ctx(conv_feat)
        tl = [np.tile((anchors[:ntubes] * 400.0)[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
        # make clip 1 differ from clip 0
        tl[1] = tl[1][::-1].copy()
        with torch.no_grad():
            history, _ = ref_utils.inference(args, conv_feat, context, nets, 3, [t.copy() for t in tl])
        for i, h in enumerate(history):
            for k in ("pred_prob", "pred_loc", "pred_first_loc", "pred_last_loc"):
                a = h[k].numpy()
                if k == "pred_prob":
                    a = a[:, 0]                       # identical over T (expand)
                g["n%d_step%d_%s" % (ntubes, i, k)] = a
            g["n%d_step%d_nums" % (ntubes, i)] = np.asarray(h["tubes_nums"])
        g["n%d_context" % ntubes] = context.numpy()
    np.savez_compressed(os.path.join(OUT, "inference_golden.npz"), **g)
    print("inference_golden ok")

    # ---------------------------------------------------------------- 6. end to end, AVA-shaped clip (C3, B=1, 11 tubes)
    g = {}
    x = R.fill_tensor("golden.c3.images", (1, 36, 3, 400, 400), "image")
    with torch.no_grad():
        conv_feat = base(x)
        context = ctx(conv_feat)
        tl = [np.tile((anchors[:11] * 400.0)[:, None, :], (1, 3, 1)).astype(np.float32)]
        history, _ = ref_utils.inference(args, conv_feat, context, nets, 3, tl)
    assert tuple(conv_feat.shape) == (1, 9, 832, 25, 25)
    d = stage_digest(conv_feat.contiguous())
    g["conv_feat_stats"] = np.asarray([d["mean"], d["absmax"], d["l2"], d["step"]], np.float64)
    g["conv_feat_sample"] = d["sample"]
    g["conv_feat_slice"] = conv_feat[0, :, ::13, ::4, ::4].contiguous().numpy()
    g["context"] = context.numpy()
    for i, h in enumerate(history):
        g["step%d_pred_prob" % i] = h["pred_prob"].numpy()[:, 0]
        for k in ("pred_loc", "pred_first_loc", "pred_last_loc"):
            g["step%d_%s" % (i, k)] = h[k].numpy()
    np.savez_compressed(os.path.join(OUT, "e2e_c3_golden.npz"), **g)
    print("e2e_c3_golden ok")

    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keyinfo, f, indent=0, sort_keys=True)
    print("done ->", OUT)


def selection_cases():
    """Seeded synthetic inputs of the training sample selection (utils/utils.py:135-423): per case the ground truths of
    two clips [G, max_chunks, 4 + classes], the initial proposals, and a fake `history` of the previous step."""
    import random as pyrandom

    nc = 60
    anchors = None
    from oracle import i3d_ref as R
    anchors = (R.anchors() * 400.0).astype(np.float32)
    cases = []
    for ci, (seed, sampling, topk, neg_ratio, max_pos, mode) in enumerate((
            (1, "softmax", -1, 2, 5, "predict"), (2, "random", 300, 2, 5, "predict"), (3, "uniform", 120, 1, 2, "predict"),
            (4, "softmax", -1, 2, 5, "mean"), (5, "softmax", 60, 3, 1, "predict"), (6, "random", -1, 2, 4, "extrapolate"))):
        rs = np.random.RandomState(1000 + seed)
        targets = []
        for b in range(2):
            G = int(rs.randint(1, 5)) if ci != 4 else 7                  # (case 4: more ground truths than max_pos_num)
            t = np.zeros((G, 3, 4 + nc), np.float32)
            for gidx in range(G):
                a = anchors[rs.randint(0, 34)] + rs.uniform(-25, 25, 4).astype(np.float32)
                for c in range(3):
                    if c == 1 or rs.rand() < 0.7:                       # neighbour chunks may be padding (all zero)
                        t[gidx, c, :4] = a + rs.uniform(-6, 6, 4).astype(np.float32) * (c != 1)
                        t[gidx, c, 4 + rs.randint(0, nc, 3)] = 1
            targets.append(t)
        tubes = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
        nums = [34, 20]
        n = sum(nums)
        hist = {}
        prob = rs.rand(n, 1, nc).astype(np.float32) ** 4
        prob[3] = prob[4]                                               # exact score ties between tubes
        hist["pred_prob"] = np.tile(prob, (1, 3, 1))
        base = np.concatenate([anchors[:34], anchors[:20]], 0)[:, None, :] + rs.uniform(-30, 30, (n, 3, 4)).astype(np.float32)
        hist["pred_loc"] = base.astype(np.float32)
        hist["pred_first_loc"] = (base + rs.uniform(-10, 10, (n, 3, 4))).astype(np.float32)
        hist["pred_last_loc"] = (base + rs.uniform(-10, 10, (n, 3, 4))).astype(np.float32)
        hist["pred_loc"][5] = -50.0                                     # clamps to an invalid box -> whole image
        hist["tubes_nums"] = nums
        a = cfg(cls_thresh=[0.2, 0.35, 0.5], reg_thresh=[0.2, 0.35, 0.5], max_pos_num=max_pos, neg_ratio=neg_ratio,
                selection_sampling=sampling, topk=topk, temporal_mode=mode)
        cases.append((seed, a, targets, tubes, hist))
    return cases, pyrandom


def selection_main():
    """tests/golden/selection_golden.npz: what the reference's train_select / select_proposals / compute_tube_iou return for
    the seeded cases above, with `random.seed(s); np.random.seed(s)` set right before every call."""
    _, ref_utils, _, _ = import_reference()
    import utils.tube_utils as ref_tu  # reference

    cases, pyrandom = selection_cases()
    g = {}
    for ci, (seed, a, targets, tubes, hist) in enumerate(cases):
        th = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in hist.items()}
        for k in ("neg_ratio", "max_pos_num", "topk"):
            g["c%d_%s" % (ci, k)] = np.asarray(getattr(a, k))
        g["c%d_sampling" % ci] = np.asarray(a.selection_sampling)
        g["c%d_mode" % ci] = np.asarray(a.temporal_mode)
        g["c%d_seed" % ci] = np.asarray(seed)
        for b, t in enumerate(targets):
            g["c%d_targets%d" % (ci, b)] = t
        for k in ("pred_prob", "pred_loc", "pred_first_loc", "pred_last_loc"):
            g["c%d_hist_%s" % (ci, k)] = hist[k][:, :1] if k == "pred_prob" else hist[k]
        for step in (1, 2, 3):
            pyrandom.seed(seed * 10 + step)
            np.random.seed(seed * 10 + step)
            sel, tgt = ref_utils.train_select(step, th if step > 1 else None, [t.copy() for t in targets],
                                              [t.copy() for t in tubes], a)
            for b in range(2):
                g["c%d_s%d_sel%d" % (ci, step, b)] = sel[b]
                g["c%d_s%d_tgt%d" % (ci, step, b)] = tgt[b]
        g["c%d_iou" % ci] = ref_tu.compute_tube_iou(targets[0][:, :, :4], hist["pred_loc"][:9])
    # IoU edge cases: padding tubes, touching and nested boxes
    t1 = np.array([[[0, 0, 10, 10], [0, 0, 0, 0]], [[0, 0, 0, 0], [0, 0, 0, 0]], [[5, 5, 20, 30], [1, 2, 3, 4]]], np.float32)
    t2 = np.array([[[10, 0, 20, 10], [0, 0, 5, 5]], [[2, 2, 8, 8], [1, 2, 3, 4]], [[0, 0, 0, 0], [0, 0, 0, 0]]], np.float32)
    g["edge_t1"], g["edge_t2"] = t1, t2
    with np.errstate(all="ignore"):
        g["edge_iou"] = ref_tu.compute_tube_iou(t1, t2)
    # select_proposals on its own: more ground truths than proposals, duplicate proposals, nothing above the threshold,
    # no negatives wanted, scores given / derived from the IoUs, every sampling mode
    rs = np.random.RandomState(77)
    edge = []
    for ei, (G, A, thr, max_pos, sampling, neg_ratio, with_scores) in enumerate((
            (7, 3, 0.2, 5, "random", 2, False), (2, 12, 0.2, 5, "softmax", 2, True), (3, 10, 0.99, 2, "uniform", 1, True),
            (1, 6, 0.1, 5, "random", 0, False), (4, 40, 0.35, 3, "softmax", 3, False), (5, 9, 0.5, 1, "uniform", 2, True))):
        gt = np.zeros((G, 1, 4), np.float32)
        xy = rs.uniform(0, 250, (G, 2)); wh = rs.uniform(40, 150, (G, 2))
        gt[:, 0] = np.concatenate([xy, xy + wh], 1)
        an = np.zeros((A, 1, 4), np.float32)
        for a_ in range(A):
            src = gt[rs.randint(0, G), 0] if rs.rand() < 0.6 else np.concatenate([rs.uniform(0, 300, 2), rs.uniform(300, 400, 2)])
            an[a_, 0] = src + rs.uniform(-20, 20, 4)
        if ei == 1:
            an[3] = an[2]                                               # duplicates: ties in every ranking
            an[4] = an[2]
        sc = rs.rand(A).astype(np.float32) if with_scores else None
        pyrandom.seed(500 + ei)
        np.random.seed(500 + ei)
        with np.errstate(all="ignore"):
            pos, neg, ious = ref_utils.select_proposals(gt.copy(), an.copy(), None if sc is None else sc.copy(), thr, max_pos, sampling, neg_ratio)
        g["e%d_gt" % ei], g["e%d_an" % ei] = gt, an
        if sc is not None:
            g["e%d_scores" % ei] = sc
        g["e%d_cfg" % ei] = np.asarray([thr, max_pos, neg_ratio], np.float64)
        g["e%d_sampling" % ei] = np.asarray(sampling)
        g["e%d_pos" % ei] = np.asarray([(int(a_), int(b_)) for a_, b_ in pos], np.int64).reshape(-1, 2)
        g["e%d_neg" % ei] = np.asarray([(int(a_), int(b_)) for a_, b_ in neg], np.int64).reshape(-1, 2)
        g["e%d_ious" % ei] = ious
    g["n_edge"] = np.asarray(6)
    g["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(OUT, "selection_golden.npz"), **g)
    print("selection_golden ok", len(g), "arrays")


def modes_main():
    """tests/golden/inference_modes_golden.npz: the reference's inference() (utils/utils.py:15-131) with temporal_mode
    "extrapolate" and "mean" (the tube extension between steps 2 and 3 without the head's neighbour regressions), same
    synthetic conv_feat / heads / 11 tubes as inference_golden.npz."""
    models, ref_utils, _, _ = import_reference()
    from oracle import i3d_ref as R

    args = cfg()
    ctx = models.ContextNet(args)
    fill_module(ctx)
    ctx.set_device("cpu")
    ctx.eval()
    nets = {"roi_net": models.ROINet("align", 7)}
    for i in range(3):
        d = models.TwoBranchNet(args)
        fill_module(d, "det%d." % i)
        d.set_device("cpu")
        d.eval()
        nets["det_net%d" % i] = d
    anchors = R.anchors()
    conv_feat = R.fill_tensor("golden.inf.feat", (2, 9, 832, 25, 25), "feat")
    with torch.no_grad():
        context = ctx(conv_feat)
    g = {}
    for mode in ("extrapolate", "mean"):
        a = cfg(temporal_mode=mode)
        tl = [np.tile((anchors[:11] * 400.0)[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
        tl[1] = tl[1][::-1].copy()
        with torch.no_grad():
            history, traj = ref_utils.inference(a, conv_feat, context, nets, 3, [t.copy() for t in tl])
        for i, h in enumerate(history):
            g["%s_step%d_pred_prob" % (mode, i)] = h["pred_prob"].numpy()[:, 0]
            g["%s_step%d_pred_loc" % (mode, i)] = h["pred_loc"].numpy()
        g["%s_step1_proposals" % mode] = np.concatenate([t[0] for t in traj[1]], 0)      # the extended tubes fed to step 3
    # extrapolate_tubes on its own
    rs = np.random.RandomState(5)
    import utils.tube_utils as ref_tu  # reference
    t = (rs.uniform(0, 400, (7, 3, 4))).astype(np.float32)
    t[0, :, 0] = 1.0; t[0, 1, 0] = 30.0                                                    # extrapolates below 0 / beyond 399
    g["ext_in"] = t
    g["ext_out_T3"] = ref_tu.extrapolate_tubes(t.copy(), 3)
    g["ext_out_T6"] = ref_tu.extrapolate_tubes(np.tile(t, (1, 2, 1)).copy(), 6)
    np.savez_compressed(os.path.join(OUT, "inference_modes_golden.npz"), **g)
    print("inference_modes_golden ok", len(g), "arrays")


def variants_main():
    """tests/golden/head_variants_golden.npz: the reference's TwoBranchNet in its two other configurations -- `cls_only=True`
    (two_branch.py:166-202: no regressors) and `no_context=True` (no ContextNet feature) -- on the head_golden inputs."""
    models, _, _, _ = import_reference()
    from oracle import i3d_ref as R

    g = {}
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    for tag, net, kw in (("cls_only", models.TwoBranchNet(cfg(), cls_only=True), dict(context_feat=cx)),
                         ("no_context", models.TwoBranchNet(cfg(no_context=True)), dict())):
        fill_module(net, "det0.")
        net.set_device("cpu")
        net.eval()
        with torch.no_grad():
            o = net(pf, **kw)
        for nme, t in zip(("prob", "loc", "first", "last"), o[:4]):
            if t is not None and torch.is_tensor(t):
                g["%s_%s" % (tag, nme)] = t.numpy()
        g["%s_present" % tag] = np.asarray([int(t is not None and torch.is_tensor(t)) for t in o[:4]])
    np.savez_compressed(os.path.join(OUT, "head_variants_golden.npz"), **g)
    print("head_variants_golden ok", sorted(g))


def tube_math_main():
    """tests/golden/tube_math_golden.npz: the reference's box / tube helpers on random inputs (utils/tube_utils.py:59-92,
    127-189, 214-246): valid_tubes (numpy and torch), get_center_size, encode_coef, decode_coef, flatten_tubes."""
    import_reference()
    import utils.tube_utils as tu  # reference

    rs = np.random.RandomState(9)
    g = {}
    t = rs.uniform(-60, 460, (9, 3, 4)).astype(np.float32)
    t[..., 2:] = t[..., :2] + rs.uniform(-4, 200, (9, 3, 2)).astype(np.float32)       # some boxes thinner than 3 px / inverted
    t[0, 0] = [10, 10, 12, 50]                                                          # x2 - 2 == x1: not valid (strict <)
    t[0, 1] = [10, 10, 12.5, 13.5]
    g["vt_in"] = t
    g["vt_np"] = tu.valid_tubes(t.copy(), 400, 400)
    g["vt_np_320x240"] = tu.valid_tubes(t.copy(), 320, 240)
    g["vt_torch"] = tu.valid_tubes(torch.from_numpy(t.copy()), 400, 400).numpy()
    a = rs.uniform(0, 300, (50, 2)); wh = rs.uniform(1, 150, (50, 2))
    boxes = np.concatenate([a, a + wh], 1).astype(np.float32)
    gt = (boxes + rs.uniform(-15, 15, boxes.shape)).astype(np.float32)
    deltas = (rs.randn(50, 4) * 0.3).astype(np.float32)
    g["boxes"], g["gt"], g["deltas"] = boxes, gt, deltas
    g["center_size"] = np.stack([v.numpy() for v in tu.get_center_size(torch.from_numpy(boxes))])
    g["encode"] = tu.encode_coef(torch.from_numpy(gt), torch.from_numpy(boxes)).numpy()
    g["decode"] = tu.decode_coef(torch.from_numpy(boxes), torch.from_numpy(deltas)).numpy()
    tl = [rs.uniform(0, 400, (n, 3, 4)).astype(np.float32) for n in (4, 0, 2)]
    for flag in (False, True):
        flat, nums = tu.flatten_tubes([x.copy() for x in tl], batch_idx=flag)
        g["flat_%d" % flag], g["flat_nums_%d" % flag] = np.asarray(flat), np.asarray(nums)
    for i, x in enumerate(tl):
        g["flat_in%d" % i] = x
    # the initial tube grids of the four anchor modes (data/ava.py:342-354 -> data/data_utils.py:19-45)
    from data.data_utils import generate_anchors as ref_anchors  # reference
    for mode, (sc, ov) in {"1": ([4 / 3, 2], [5 / 6, 3 / 4]), "2": ([4 / 3, 2, 3], [5 / 6, 3 / 4, 1 / 2]),
                           "3": ([4 / 3, 2, 3, 4], [5 / 6, 3 / 4, 1 / 2, 1 / 4]),
                           "4": ([4 / 3, 2, 3, 4, 5], [5 / 6, 3 / 4, 1 / 2, 1 / 4, 0])}.items():
        g["anchors_mode%s" % mode] = ref_anchors(sc, ov)
    np.savez_compressed(os.path.join(OUT, "tube_math_golden.npz"), **g)
    print("tube_math_golden ok", len(g), "arrays")


def postprocess_history(seed, nums, num_classes=60):
    """A seeded synthetic `history` (utils/utils.py:81-85) that exercises every branch of the evaluation loop: scores below
    conf_thresh, exact score ties inside and across classes, whole classes without a detection, boxes that valid_tubes
    clamps, and degenerate boxes it replaces by the whole image."""
    rs = np.random.RandomState(seed)
    N = int(sum(nums))
    hist = []
    for Tl in (3, 3, 9):
        xy = rs.uniform(-30, 330, (N, Tl, 2))
        wh = rs.uniform(-5, 160, (N, Tl, 2))                          # some negative / < 2 px sizes -> whole-image boxes
        loc = np.concatenate([xy, xy + wh], 2).astype(np.float32)
        loc[rs.rand(N) < 0.3] += rs.uniform(0, 120, (1, 1, 4)).astype(np.float32)   # clusters of overlapping boxes
        prob = rs.rand(N, num_classes).astype(np.float32) ** 6        # most scores tiny
        prob[rs.rand(N, num_classes) < 0.25] = 0.005                  # below conf_thresh
        prob = np.round(prob * 16) / 16 * (rs.rand(N, num_classes) < 0.5) + prob * 0.02   # ties: multiples of 1/16 + small
        prob[:, 7] = 0.0                                              # a class with no detection at all
        prob[:, 9] = 0.5                                              # a class where every box ties
        prob = prob.astype(np.float32)
        hist.append({"pred_prob": np.repeat(prob[:, None, :], Tl, axis=1), "pred_loc": loc, "tubes_nums": list(nums)})
    return hist


def postprocess_main():
    """Run the evaluation loop of the reference's test.py (:157-210 -- it is inline code of main(), not a function) on the
    seeded history: the loop's source is read from the reference where it lies, compiled and executed here with the
    reference's own nms / valid_tubes; the fixture records, for every row the loop writes, (iteration, clip, class, box, score)
    in the order written, plus the text lines themselves."""
    import io
    import textwrap
    models, ref_utils, ref_nms, _ = import_reference()
    from utils.tube_utils import valid_tubes as ref_valid_tubes   # reference
    src = open(os.path.join(REF, "test.py")).read().split("\n")
    beg = next(k for k, l in enumerate(src) if l.strip() == "# loop for each  iteration")
    end = next(k for k, l in enumerate(src) if k > beg and l.strip() == "fout.close()")
    body = textwrap.dedent("\n".join(src[beg:end]))
    code = compile(body, os.path.join(REF, "test.py") + ":eval-loop", "exec")
    g = {}
    nums = [11, 7, 9]
    cases = {"all": dict(evaluate_topk=-1, topk=-1), "top20": dict(evaluate_topk=1, topk=20), "topm1": dict(evaluate_topk=5, topk=-1)}
    hist_np = postprocess_history(2024, nums)
    for i, h in enumerate(hist_np):
        g["hist%d_prob" % i] = h["pred_prob"][:, 0].copy()
        g["hist%d_loc" % i] = h["pred_loc"]
    g["nums"] = np.asarray(nums)
    for tag, kw in cases.items():
        args = types.SimpleNamespace(num_classes=60, conf_thresh=0.01, nms_thresh=0.4, **kw)
        history = [{"pred_prob": torch.from_numpy(h["pred_prob"].copy()), "pred_loc": torch.from_numpy(h["pred_loc"].copy()),
                    "tubes_nums": list(nums)} for h in hist_np]
        rows, lines = [], []

        class Sink:
            def __init__(self, ns):
                self.ns = ns

            def write(self, text):
                ns = self.ns                                     # the loop's own variables at the moment it writes a row
                rows.append([ns["i"], ns["b"], ns["cl_ind"]] + [float(v) for v in ns["box"]] + [float(ns["s"])])
                lines.append(text)

        ns = {"np": np, "nms": ref_nms, "valid_tubes": ref_valid_tubes, "args": args, "history": history,
              "infos": [{"video_name": "vid%d" % b, "fid": 900 + b} for b in range(len(nums))],
              "label_dict": {c: c + 1 for c in range(60)}, "width": 400, "height": 400}
        ns["fouts"] = [Sink(ns) for _ in history]
        exec(code, ns)
        r = np.asarray(rows, np.float64)
        g[tag + "_meta"] = r[:, :3].astype(np.int32)
        g[tag + "_box"] = r[:, 3:7].astype(np.float32)
        g[tag + "_score"] = r[:, 7].astype(np.float32)
        g[tag + "_lines"] = np.asarray(lines)
        print("postprocess", tag, len(rows), "rows")
    np.savez_compressed(os.path.join(OUT, "postprocess_golden.npz"), **g)


def i3d_main():
    """The full Kinetics classifier I3D (models/i3dpt.py:175-262; never called by STEP's scripts): the
    reference's module with closed-form weights on the smallest clip its (2,7,7) average pool accepts."""
    torch.set_num_threads(8)
    import_reference()
    from models.i3dpt import I3D           # reference
    from oracle import i3d_ref as R
    net = I3D(num_classes=24, dropout_prob=0.5)
    shapes = fill_module(net, "i3dcls.")
    net.eval()
    x = R.fill_tensor("golden.i3dcls.clip", (1, 3, 16, 224, 224), "image")
    with torch.no_grad():
        prob, logits = net(x)
        ours = R.i3d_forward(x, R.fill_state_dict(shapes, "i3dcls."))
    assert float((ours[1] - logits).abs().max()) < 1e-5 * float(logits.abs().max())
    keys = sorted(shapes)
    np.savez_compressed(os.path.join(OUT, "i3d_classifier_golden.npz"), prob=prob.numpy(), logits=logits.numpy(),
                        keys=np.asarray(keys), shapes=np.asarray([str(tuple(shapes[k])) for k in keys]))
    print("i3d classifier", tuple(logits.shape), float(logits.abs().max()))


def head_grad_margin_main():
    """As head_grad_main, with the MARGIN weights of oracle.i3d_ref.fill_state_dict_margin: every ReLU pre-activation of the
    reference's TwoBranchNet is at least `margin` (recorded, relative to the layer's rms) away from zero, so a reimplementation's
    gradients must agree to fp32 noise -- no ReLU mask can flip -- and the test asserts 1e-3."""
    from oracle import i3d_ref as R
    models, _, _, _ = import_reference()
    hg = np.load(os.path.join(OUT, "head_golden.npz"))
    det = models.TwoBranchNet(cfg())
    shapes = {k: tuple(v.shape) for k, v in det.state_dict().items()}
    det.load_state_dict(R.fill_state_dict_margin(shapes, "det0."))
    det.set_device("cpu")
    det.train()
    margins = []

    def pre(_m, inp):
        x = inp[0].detach()
        margins.append(float(x.abs().min() / x.pow(2).mean().sqrt()))

    def post_bn(_m, _i, out):
        x = out.detach()
        margins.append(float(x.abs().min() / x.pow(2).mean().sqrt()))

    for m in det.modules():
        if isinstance(m, torch.nn.ReLU):
            m.register_forward_pre_hook(pre)
        if isinstance(m, torch.nn.BatchNorm3d):
            m.register_forward_hook(post_bn)
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    o = det(pf, context_feat=cx, tubes=torch.from_numpy(hg["loss_tubes"]), targets=torch.from_numpy(hg["loss_targets"]))
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    assert len(margins) == 12 + 9, len(margins)                 # 12 units of mixed_5b / 5c, 3 ReLUs in each of the 3 bottlenecks
    assert min(margins) >= 1e-3, margins
    g = {"loss": np.float64(loss.item()), "margin": np.float64(min(margins)), "outputs": np.concatenate([t.detach().reshape(-1).numpy() for t in o[:4]])}
    names = []
    for k, p in det.named_parameters():
        if not p.requires_grad:
            continue
        f = p.grad.detach().reshape(-1)
        step = max(1, f.numel() // 512)
        names.append(k)
        g["norm." + k] = np.float64(f.double().norm().item())
        g["step." + k] = np.int64(step)
        g["sample." + k] = f[::step][:512].numpy().copy()
    g["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "head_grad_margin_golden.npz"), **g)
    print("head_grad_margin_golden ok: %d tensors, loss %.6f, least |pre-activation| / rms over the %d ReLU sites %.3e" % (
        len(names), loss.item(), len(margins), min(margins)))


def head_grad_main():
    """Gradients of the reference's OWN TwoBranchNet under its own autograd (two_branch.py:215-333, train mode, frozen BN,
    dropout 0): the loss of train.py:318-331 (cls + 5 reg + neighbour) on the head_golden inputs, back-propagated to every
    trainable parameter.  Stored per parameter as its L2 norm and a strided sample (the tensors themselves are ~80 MB)."""
    from oracle import i3d_ref as R
    models, _, _, _ = import_reference()
    hg = np.load(os.path.join(OUT, "head_golden.npz"))
    det = models.TwoBranchNet(cfg())
    fill_module(det, "det0.")
    det.set_device("cpu")
    det.train()
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    o = det(pf, context_feat=cx, tubes=torch.from_numpy(hg["loss_tubes"]), targets=torch.from_numpy(hg["loss_targets"]))
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    g = {"loss": np.float64(loss.item())}
    names = []
    for k, p in det.named_parameters():
        if not p.requires_grad:
            continue
        f = p.grad.detach().reshape(-1)
        step = max(1, f.numel() // 512)
        names.append(k)
        g["norm." + k] = np.float64(f.double().norm().item())
        g["step." + k] = np.int64(step)
        g["sample." + k] = f[::step][:512].numpy().copy()
    g["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "head_grad_golden.npz"), **g)
    print("head_grad_golden ok: %d tensors, loss %.6f" % (len(names), loss.item()))


def base_grad_main():
    """Gradients of the reference's OWN BaseNet (models/networks.py:55-81 over models/i3dpt.py) under its own autograd, training
    mode with frozen BN: a scalar (every output element times a signed weight) back-propagated to all 45 trainable tensors, at
    the two sizes tests/module_cases.case_basenet_backward_matches_oracle_autograd uses (same seeded clip, same weight tensor).
    Stored per parameter as L2 norm + strided 512-element sample."""
    from oracle import i3d_ref as R
    models, _, _, _ = import_reference()
    g = {}
    for tag, shape in (("gpu", (1, 8, 3, 112, 112)), ("emul", (1, 4, 3, 32, 32))):
        net = models.BaseNet(cfg())
        fill_module(net)
        net.train()
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(10)) * 2 - 1
        y = net(x)
        wgt = R.fill_tensor("golden.bwd.base.w", tuple(y.shape), "image")
        (y * wgt).sum().backward()
        names = []
        for k, p in net.named_parameters():
            if not p.requires_grad:
                continue
            f = p.grad.detach().reshape(-1)
            step = max(1, f.numel() // 512)
            names.append(k)
            g["%s.norm.%s" % (tag, k)] = np.float64(f.double().norm().item())
            g["%s.step.%s" % (tag, k)] = np.int64(step)
            g["%s.sample.%s" % (tag, k)] = f[::step][:512].numpy().copy()
        g[tag + ".names"] = np.array(names)
        g[tag + ".out_l2"] = np.float64(y.detach().double().norm().item())
        print("base_grad_golden %s: %d tensors, |y| %.6f" % (tag, len(names), g[tag + ".out_l2"]))
    np.savez_compressed(os.path.join(OUT, "base_grad_golden.npz"), **g)


def bn_train_main():
    """--freeze_stats False: the reference's OWN BaseNet with its BatchNorm layers in TRAINING mode (models/networks.py:85-99 only forces
    eval when freeze_stats; models/i3dpt.py:95-110) and trainable BN affine (--freeze_affine False), two clips per batch so that the batch
    statistics span clips.  One forward + backward under the reference's autograd: output (digest + sample), every BN layer's running
    statistics after the step, gradients of all 135 trainable tensors (45 conv weights + 45 x (gamma, beta)) as L2 norm + strided sample.
    Three sizes: 'emul' [2,4,3,48,48] (the interpreter's), 'gpu' [2,8,3,48,48] and 'c1' [2,8,3,112,112].  Conditioning, measured by
    running the restatement in fp32 and in fp64 (worst relative L2 over the 135 gradients): 48x48 clips 4e-5; a 32x32 clip 1.5e-2 (its
    last maps are 1x2x2: a 3x3x3 pool makes them constant per clip and the batch variance of the following conv a difference of nearly
    equal numbers); the C1-sized clip 6e-3 (the signed loss weights make the BN bias gradients sums with heavy cancellation, and a few
    ReLU masks flip between fp32 and fp64) -- so gradients are pinned at 1e-3 on the 48x48 clips and output / running statistics also
    at C1 size."""
    from oracle import i3d_ref as R
    models, _, _, _ = import_reference()
    g = {}
    for tag, shape in (("gpu", (2, 8, 3, 48, 48)), ("emul", (2, 4, 3, 48, 48)), ("c1", (2, 8, 3, 112, 112))):
        net = models.BaseNet(cfg(freeze_stats=False, freeze_affine=False))
        fill_module(net)
        net.train()
        assert all(m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm3d))
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(11)) * 2 - 1
        y = net(x)
        wgt = R.fill_tensor("golden.bn_train.w", tuple(y.shape), "image")
        (y * wgt).sum().backward()
        f = y.detach().reshape(-1)
        step = max(1, f.numel() // 4096)
        g[tag + ".out_l2"] = np.float64(f.double().norm().item())
        g[tag + ".out_step"] = np.int64(step)
        g[tag + ".out_sample"] = f[::step][:4096].numpy().copy()
        sd = net.state_dict()
        rkeys = [k for k in sd if k.endswith("running_mean") or k.endswith("running_var")]
        g[tag + ".running_keys"] = np.array(rkeys)
        g[tag + ".running"] = np.concatenate([sd[k].numpy().reshape(-1) for k in rkeys]).astype(np.float32)
        g[tag + ".tracked"] = np.array([int(sd[k]) for k in sd if k.endswith("num_batches_tracked")], np.int64)
        names = []
        for k, p in net.named_parameters():
            assert p.requires_grad and p.grad is not None, k
            fg = p.grad.detach().reshape(-1)
            st = max(1, fg.numel() // 512)
            names.append(k)
            g["%s.norm.%s" % (tag, k)] = np.float64(fg.double().norm().item())
            g["%s.step.%s" % (tag, k)] = np.int64(st)
            g["%s.sample.%s" % (tag, k)] = fg[::st][:512].numpy().copy()
        g[tag + ".names"] = np.array(names)
        print("bn_train_golden %s: %d tensors, |y| %.6f, %d running-stat values" % (tag, len(names), g[tag + ".out_l2"], g[tag + ".running"].size))
    np.savez_compressed(os.path.join(OUT, "bn_train_golden.npz"), **g)


FULL_SIZE_CLIPS = {"c2": ("golden.c2.images", (1, 32, 3, 224, 224), (1, 8, 832, 14, 14)),     # BASELINE C2: one clip of the batch of 8
                   "c5": ("golden.c5.images", (1, 64, 3, 400, 400), (1, 16, 832, 25, 25))}    # BASELINE C5: one long clip


def full_size_main():
    """The imported reference's BaseNet (models/networks.py:69-83) on ONE clip of the C2 and of the C5 shape: per-stage digests (statistics +
    a strided 256-element sample, as for C1) and a 4096-element strided sample of conv_feat -- a few KB that pin the restatement
    (oracle/i3d_ref.basenet_forward) at the shapes the bench number is quoted on; tests/module_cases.py then holds the HIP path to the
    restatement's FULL tensor at that size."""
    torch.set_num_threads(8)
    models, _, _, _ = import_reference()
    from oracle import i3d_ref as R
    base = models.BaseNet(cfg())
    fill_module(base)
    base.eval()
    g = {}
    for tag, (name, shape, oshape) in FULL_SIZE_CLIPS.items():
        x = R.fill_tensor(name, shape, "image")
        stages = []
        hooks = [m.register_forward_hook(lambda _m, _i, o: stages.append(stage_digest(o) | {"shape": tuple(o.shape)})) for m in base.base_model]
        with torch.no_grad():
            y = base(x)
        for h in hooks:
            h.remove()
        assert tuple(y.shape) == oshape, y.shape
        f = y.contiguous().reshape(-1)
        st = max(1, f.numel() // 4096)
        g[tag + ".out_shape"] = np.asarray(y.shape)
        g[tag + ".out_stats"] = np.asarray([float(f.double().mean()), float(f.abs().max()), float(f.double().norm()), st], np.float64)
        g[tag + ".out_sample"] = f[::st][:4096].numpy().copy()
        for i, d in enumerate(stages):
            g["%s.stage%d_shape" % (tag, i)] = np.asarray(d["shape"])
            g["%s.stage%d_stats" % (tag, i)] = np.asarray([d["mean"], d["absmax"], d["l2"], d["step"]], np.float64)
            g["%s.stage%d_sample" % (tag, i)] = d["sample"]
        print("full_size_golden %s: out %s absmax %.4f l2 %.4f" % (tag, tuple(y.shape), g[tag + ".out_stats"][1], g[tag + ".out_stats"][2]))
    np.savez_compressed(os.path.join(OUT, "full_size_golden.npz"), **g)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "full_size":
        full_size_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "bn_train":
        bn_train_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "i3d":
        i3d_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "head_grad_margin":
        head_grad_margin_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "head_grad":
        head_grad_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "base_grad":
        base_grad_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "postprocess":
        postprocess_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "tube_math":
        tube_math_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "variants":
        variants_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "selection":
        selection_main()
    elif len(sys.argv) > 1 and sys.argv[1] == "modes":
        modes_main()
    else:
        main()
        selection_main()
        modes_main()
        variants_main()
        tube_math_main()
        postprocess_main()
        i3d_main()
        bn_train_main()
        head_grad_main()
        base_grad_main()
        full_size_main()
