"""The oracle (oracle/step_oracle.c, oracle/i3d_ref.py) against the golden vectors produced by the
reference itself (oracle/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import i3d_ref as R

from tests.conftest import GOLDEN


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_roi_align_forward_bit_exact(golden):
    g = golden("roi_nms_golden")
    feat = R.fill_tensor("golden.roi.feat", (3, 6, 25, 25), "image").numpy()
    for tag, pooled, sr in (("p7s0", (7, 7), 0), ("p7s2", (7, 7), 2), ("p3x5s0", (3, 5), 0)):
        out = oracle.roi_align_forward(feat, g["align_rois"], pooled, 1.0 / 16.0, sr)
        assert np.array_equal(out, g["align_out_" + tag]), tag
    assert oracle.roi_align_forward(feat, np.zeros((0, 5), np.float32), (7, 7), 1 / 16., 0).shape == (0, 6, 7, 7)


def test_roi_align_tube_call(golden):
    g = golden("roi_nms_golden")
    conv = R.fill_tensor("golden.roi.conv", (2, 3, 16, 25, 25), "feat")
    out = R.roinet_forward(conv, torch.from_numpy(g["tube_rois"]))
    assert np.array_equal(out.numpy(), g["tube_out"])


def test_nms_bit_exact(golden):
    g = golden("roi_nms_golden")
    for i in range(int(g["nms_count"])):
        keep = oracle.nms(g["nms%d_boxes" % i], g["nms%d_scores" % i], float(g["nms%d_thr" % i]))
        assert keep.dtype == np.int64
        assert np.array_equal(keep, g["nms%d_keep" % i]), i
    assert oracle.nms(np.zeros((0, 4)), np.zeros((0,)), 0.4).size == 0


def test_nms_threshold_equality_uses_ge(golden):
    g = golden("roi_nms_golden")
    # case 8 of make_golden: IoU == thr == 0.5 exactly; the CPU reference suppresses (>=)
    i = 8
    assert float(g["nms%d_thr" % i]) == 0.5
    assert list(g["nms%d_keep" % i]) == [0, 2]


def test_nms_batched_matches_single():
    rs = np.random.RandomState(3)
    G, kmax = 7, 40
    boxes = np.zeros((G, kmax, 4), np.float32)
    scores = np.zeros((G, kmax), np.float32)
    counts = rs.randint(0, kmax + 1, G).astype(np.int32)
    counts[0], counts[1] = 0, kmax
    for gi in range(G):
        xy = rs.uniform(0, 300, (kmax, 2))
        wh = rs.uniform(10, 120, (kmax, 2))
        boxes[gi] = np.concatenate([xy, xy + wh], 1)
        scores[gi] = rs.permutation(kmax) / kmax
    mask = oracle.nms_batched(boxes, scores, counts, 0.4)
    for gi in range(G):
        k = oracle.nms(boxes[gi, :counts[gi]], scores[gi, :counts[gi]], 0.4)
        exp = np.zeros(kmax, np.uint8)
        exp[k] = 1
        assert np.array_equal(mask[gi], exp)


def test_roi_align_backward_is_adjoint_of_forward():
    rs = np.random.RandomState(0)
    B, C, H, W = 3, 4, 25, 25
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois = np.array([[0, 0, 0, 400, 400], [1, 33.3, 50.1, 180.7, 222.2], [2, -20, 300, 90, 450],
                     [1, 100, 100, 100.5, 100.5], [0, 384, 384, 400, 400]], np.float32)
    for sr in (0, 2):
        y = oracle.roi_align_forward(x, rois, (7, 7), 1 / 16., sr)
        g = rs.randn(*y.shape).astype(np.float32)
        gx = oracle.roi_align_backward(g, rois, (7, 7), 1 / 16., sr, x.shape)
        lhs = float((y.astype(np.float64) * g).sum())
        rhs = float((x.astype(np.float64) * gx).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_roi_pool_hand_cases():
    # 1 channel 4x4 ramp, scale 1: ROI (0,0,3,3) pooled 2x2 -> max of each 2x2 quadrant
    x = np.arange(16, dtype=np.float32).reshape(1, 1, 4, 4)
    out, arg = oracle.roi_pool_forward(x, np.array([[0, 0, 0, 3, 3]], np.float32), (2, 2), 1.0)
    assert out.reshape(-1).tolist() == [5, 7, 13, 15]
    assert arg.reshape(-1).tolist() == [5, 7, 13, 15]
    # ROI entirely outside -> empty bins: value 0, argmax -1
    out, arg = oracle.roi_pool_forward(x, np.array([[0, 10, 10, 12, 12]], np.float32), (2, 2), 1.0)
    assert out.reshape(-1).tolist() == [0, 0, 0, 0] and arg.reshape(-1).tolist() == [-1] * 4
    # round(): 0.5 rounds away from zero -> start 1
    out, _ = oracle.roi_pool_forward(x, np.array([[0, 0.5, 0.5, 1.4, 1.4]], np.float32), (1, 1), 1.0)
    assert out.reshape(-1).tolist() == [5]
    g = np.ones((1, 1, 2, 2), np.float32)
    _, arg = oracle.roi_pool_forward(x, np.array([[0, 0, 0, 3, 3]], np.float32), (2, 2), 1.0)
    gi = oracle.roi_pool_backward(g, arg, np.array([[0, 0, 0, 3, 3]], np.float32), (2, 2), x.shape)
    assert gi.sum() == 4 and gi.reshape(-1)[[5, 7, 13, 15]].tolist() == [1, 1, 1, 1]


def test_shape_tables_match_reference_state_dicts():
    info = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    for name, table in (("BaseNet", R.backbone_shapes()), ("ContextNet", R.context_shapes()),
                        ("TwoBranchNet", R.twobranch_shapes()), ("TwoBranchNet_cls_only", R.twobranch_shapes(cls_only=True))):
        ref = {k: tuple(v) for k, v in info[name].items()}
        assert set(ref) == set(table), name
        for k in ref:
            assert tuple(table[k]) == ref[k], (name, k)
    assert len(info["BaseNet"]) == 270 and len(info["ContextNet"]) == 72 and len(info["TwoBranchNet"]) == 94
    assert np.allclose(R.anchors(), np.asarray(info["anchors34"], np.float32))


def test_backbone_c1(golden):
    g = golden("i3d_c1_golden")
    sd = R.fill_state_dict(R.backbone_shapes())
    x = R.fill_tensor("golden.c1.images", (1, 8, 3, 112, 112), "image")
    with torch.no_grad():
        y, stages = R.basenet_forward(x, sd, return_stages=True)
    assert tuple(y.shape) == (1, 2, 832, 7, 7)
    assert rel_err(y.contiguous().numpy(), g["conv_feat"]) < 1e-5
    for i, s in enumerate(stages):
        assert list(s.shape) == list(g["stage%d_shape" % i])
        step = int(g["stage%d_stats" % i][3])
        assert rel_err(s.reshape(-1)[::step][:256].numpy(), g["stage%d_sample" % i]) < 1e-5, i


FULL_SIZE_CLIPS = {"c2": ("golden.c2.images", (1, 32, 3, 224, 224)), "c5": ("golden.c5.images", (1, 64, 3, 400, 400))}


@pytest.mark.parametrize("tag", ["c2", "c5"])
def test_backbone_full_size(golden, tag):
    """The restatement at the shapes the bench numbers are quoted on (BASELINE C2: T=32, 224^2; C5: T=64, 400^2), one clip each, against
    digests of the imported reference's BaseNet at that shape (oracle/make_golden.py full_size; models/networks.py:69-83)."""
    g = golden("full_size_golden")
    name, shape = FULL_SIZE_CLIPS[tag]
    sd = R.fill_state_dict(R.backbone_shapes())
    x = R.fill_tensor(name, shape, "image")
    with torch.no_grad():
        y, stages = R.basenet_forward(x, sd, return_stages=True)
    assert list(y.shape) == list(g[tag + ".out_shape"])
    f = y.contiguous().reshape(-1)
    st = int(g[tag + ".out_stats"][3])
    assert rel_err(f[::st][:4096].numpy(), g[tag + ".out_sample"]) < 1e-5
    assert abs(float(f.double().norm()) - g[tag + ".out_stats"][2]) < 1e-5 * g[tag + ".out_stats"][2]
    for i, s in enumerate(stages):
        assert list(s.shape) == list(g["%s.stage%d_shape" % (tag, i)])
        step = int(g["%s.stage%d_stats" % (tag, i)][3])
        assert rel_err(s.reshape(-1)[::step][:256].numpy(), g["%s.stage%d_sample" % (tag, i)]) < 1e-5, i
        assert abs(float(s.double().norm()) - g["%s.stage%d_stats" % (tag, i)][2]) < 1e-5 * g["%s.stage%d_stats" % (tag, i)][2], i


def test_single_ops(golden):
    g = golden("ops_golden")
    xin = R.fill_tensor("golden.pool.in", (2, 5, 6, 9, 11), "image")
    for tag, k, s in (("k133s122", (1, 3, 3), (1, 2, 2)), ("k333s222", (3, 3, 3), (2, 2, 2)),
                      ("k333s111", (3, 3, 3), (1, 1, 1)), ("k222s222", (2, 2, 2), (2, 2, 2))):
        assert np.array_equal(R.maxpool_tf(xin, k, s).numpy(), g["pool_" + tag]), tag
    # zero-valued (not -inf) padding: an all -1 input gives 0 where the window overhangs
    out = R.maxpool_tf(-torch.ones(1, 1, 4, 5, 5), (3, 3, 3), (2, 2, 2)).numpy()
    assert np.array_equal(out, g["pool_allneg"]) and out.max() == 0.0 and out.min() == -1.0
    info = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    sd = R.fill_state_dict({k: tuple(v) for k, v in info["Mixed_small"].items()}, "golden.mixed.")
    sd = {"m." + k: v for k, v in sd.items()}
    xm = R.fill_tensor("golden.mixed.in", (2, 24, 3, 9, 7), "feat")
    with torch.no_grad():
        assert rel_err(R.mixed(xm, sd, "m").numpy(), g["mixed_out"]) < 1e-5
    for tag, ci, co, k, s, shp in (("stem", 3, 16, (7, 7, 7), (2, 2, 2), (1, 3, 9, 21, 19)),
                                   ("k3", 20, 24, (3, 3, 3), (1, 1, 1), (2, 20, 3, 6, 7)),
                                   ("k1", 20, 12, (1, 1, 1), (1, 1, 1), (2, 20, 3, 6, 7))):
        shapes = {"conv3d.weight": (co, ci) + k}
        for nme in ("weight", "bias", "running_mean", "running_var"):
            shapes["batch3d." + nme] = (co,)
        sd = {"u." + kk: v for kk, v in R.fill_state_dict(shapes, "golden.unit." + tag + ".").items()}
        with torch.no_grad():
            y = R.unit3d(R.fill_tensor("golden.unit.%s.in" % tag, shp, "image"), sd, "u", stride=s)
        assert rel_err(y.numpy(), g["unit_%s_out" % tag]) < 1e-5, tag


def test_heads(golden):
    g = golden("head_golden")
    sdc = R.fill_state_dict(R.context_shapes())
    cf = R.fill_tensor("golden.ctx.feat", (1, 3, 832, 25, 25), "feat")
    with torch.no_grad():
        assert rel_err(R.contextnet_forward(cf, sdc).numpy(), g["context_out"]) < 1e-5
    sd = R.fill_state_dict(R.twobranch_shapes(), "det0.")
    for tl in (3, 9):
        pf = R.fill_tensor("golden.det.pooled%d" % tl, (2, tl, 832, 7, 7), "feat")
        cx = R.fill_tensor("golden.det.ctx%d" % tl, (2, 1024, tl, 1, 1), "feat")
        with torch.no_grad():
            o = R.twobranch_forward(pf, cx, sd)
        for nme, t in zip(("prob", "loc", "first", "last"), o[:4]):
            assert rel_err(t.numpy(), g["det_T%d_%s" % (tl, nme)]) < 1e-4, (tl, nme)
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    with torch.no_grad():
        o = R.twobranch_forward(pf, cx, sd, tubes=torch.from_numpy(g["loss_tubes"]),
                                targets=torch.from_numpy(g["loss_targets"]))
    assert rel_err(o[4].numpy(), g["loss_cls"]) < 1e-4
    assert rel_err(o[5].numpy(), g["loss_loc"]) < 1e-4
    assert rel_err(o[6].numpy(), g["loss_nbr"]) < 1e-4


def test_head_gradients_of_the_restatement_match_the_reference_autograd(golden):
    """a-19: the training step's parity anchor.  tests/golden/head_grad_golden.npz holds the gradients the REFERENCE's own
    TwoBranchNet produced under its own autograd for the loss of train.py:318-331 (`python -m oracle.make_golden head_grad`);
    the restatement (oracle/i3d_ref.twobranch_forward + torch autograd) -- which the HIP path's gradients are compared with on the
    GPU -- reproduces them."""
    hg, g = golden("head_golden"), golden("head_grad_golden")
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "batch3d" not in k and "running" not in k)
          for k, v in R.fill_state_dict(R.twobranch_shapes(), "det0.").items()}
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    o = R.twobranch_forward(pf, cx, sd, tubes=torch.from_numpy(hg["loss_tubes"]), targets=torch.from_numpy(hg["loss_targets"]))
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    names = [str(n) for n in g["names"]]
    assert len(names) == 34
    for k in names:
        gr = sd[k].grad.reshape(-1)
        step = int(g["step." + k])
        assert abs(float(gr.double().norm()) - float(g["norm." + k])) <= 2e-4 * float(g["norm." + k]), k
        a, b = gr[::step][:512].numpy().astype(np.float64), g["sample." + k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-4 * max(np.linalg.norm(b), 1e-30), k


def test_head_gradients_with_relu_margins_match_the_reference_autograd(golden):
    """The same anchor on the MARGIN weights (oracle.i3d_ref.fill_state_dict_margin; head_grad_margin_golden.npz from the reference's
    own autograd, every ReLU pre-activation >= 0.35 rms away from zero): no mask can flip, so the restatement must agree to fp32
    noise -- 1e-5 here, and the HIP path is held to 1e-3 against the same file (module_cases.case_training_step_matches_torch_autograd)."""
    hg, g = golden("head_golden"), golden("head_grad_margin_golden")
    assert float(g["margin"]) >= 1e-3
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "batch3d" not in k and "running" not in k)
          for k, v in R.fill_state_dict_margin(R.twobranch_shapes(), "det0.").items()}
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat")
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat")
    o = R.twobranch_forward(pf, cx, sd, tubes=torch.from_numpy(hg["loss_tubes"]), targets=torch.from_numpy(hg["loss_targets"]))
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for k in (str(n) for n in g["names"]):
        gr = sd[k].grad.reshape(-1)
        step = int(g["step." + k])
        assert abs(float(gr.double().norm()) - float(g["norm." + k])) <= 1e-5 * float(g["norm." + k]), k
        a, b = gr[::step][:512].numpy().astype(np.float64), g["sample." + k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(b), 1e-30), k


def test_backbone_gradients_of_the_restatement_match_the_reference_autograd(golden):
    """a-19, backbone leg: base_grad_golden.npz = the gradients of the reference's own BaseNet under its own autograd
    (`python -m oracle.make_golden base_grad`); the restatement's autograd reproduces them (the 32x32 clip of the interpreter run)."""
    g = golden("base_grad_golden")
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "batch3d" not in k) for k, v in R.fill_state_dict(R.backbone_shapes()).items()}
    x = torch.rand(1, 4, 3, 32, 32, generator=torch.Generator().manual_seed(10)) * 2 - 1
    y = R.basenet_forward(x, sd)
    assert abs(float(y.detach().double().norm()) - float(g["emul.out_l2"])) < 1e-5 * float(g["emul.out_l2"])
    wgt = R.fill_tensor("golden.bwd.base.w", tuple(y.shape), "image")
    (y * wgt).sum().backward()
    for k in (str(n) for n in g["emul.names"]):
        gr = sd[k].grad.reshape(-1)
        nr = float(g["emul.norm." + k])
        assert abs(float(gr.double().norm()) - nr) <= 2e-4 * nr, k
        a, b = gr[::int(g["emul.step." + k])][:512].numpy().astype(np.float64), g["emul.sample." + k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-4 * max(np.linalg.norm(b), 1e-30), k


def test_batch_statistics_bn_restatement_matches_the_reference(golden):
    """--freeze_stats False: bn_train_golden.npz = the reference's OWN BaseNet with its BatchNorm layers in training mode and trainable BN
    affine, forward + backward on a two-clip batch (`python -m oracle.make_golden bn_train`).  The restatement with train_bn=True
    reproduces the output, every running statistic after the step and all 135 gradients (the interpreter-sized clip)."""
    g = golden("bn_train_golden")
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in R.fill_state_dict(R.backbone_shapes()).items()}
    x = torch.rand(2, 4, 3, 48, 48, generator=torch.Generator().manual_seed(11)) * 2 - 1
    y = R.basenet_forward(x, sd, train_bn=True)
    assert abs(float(y.detach().double().norm()) - float(g["emul.out_l2"])) < 1e-5 * float(g["emul.out_l2"])
    f = y.detach().reshape(-1)[::int(g["emul.out_step"])][:4096].numpy()
    assert rel_err(f, g["emul.out_sample"]) < 1e-5
    run = np.concatenate([sd[str(k)].detach().numpy().reshape(-1) for k in g["emul.running_keys"]])
    assert rel_err(run, g["emul.running"]) < 1e-5
    assert all(int(sd[k]) == 1 for k in sd if k.endswith("num_batches_tracked")) and bool((g["emul.tracked"] == 1).all())
    wgt = R.fill_tensor("golden.bn_train.w", tuple(y.shape), "image")
    (y * wgt).sum().backward()
    names = [str(n) for n in g["emul.names"]]
    assert len(names) == 135
    for k in names:
        gr = sd[k].grad.reshape(-1)
        nr = float(g["emul.norm." + k])
        assert abs(float(gr.double().norm()) - nr) <= 2e-4 * nr + 1e-12, k
        a, b = gr[::int(g["emul.step." + k])][:512].numpy().astype(np.float64), g["emul.sample." + k].astype(np.float64)
        assert np.linalg.norm(a - b) <= 2e-4 * max(np.linalg.norm(b), 1e-30) + 1e-12, k


@pytest.mark.parametrize("ntubes", [11, 34])
def test_inference_history(golden, ntubes):
    g = golden("inference_golden")
    conv_feat = R.fill_tensor("golden.inf.feat", (2, 9, 832, 25, 25), "feat")
    with torch.no_grad():
        context = R.contextnet_forward(conv_feat, R.fill_state_dict(R.context_shapes()))
    assert rel_err(context.numpy(), g["n%d_context" % ntubes]) < 1e-5
    nets = {"det_net%d" % i: R.fill_state_dict(R.twobranch_shapes(), "det%d." % i) for i in range(3)}
    a = R.anchors()[:ntubes] * 400.0
    tl = [np.tile(a[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
    tl[1] = tl[1][::-1].copy()
    with torch.no_grad():
        hist = R.inference(conv_feat, context, nets, tl)
    for i, h in enumerate(hist):
        assert list(h["tubes_nums"]) == list(g["n%d_step%d_nums" % (ntubes, i)])
        assert rel_err(h["pred_prob"][:, 0].numpy(), g["n%d_step%d_pred_prob" % (ntubes, i)]) < 1e-4
        for k in ("pred_loc", "pred_first_loc", "pred_last_loc"):
            assert rel_err(h[k].numpy(), g["n%d_step%d_%s" % (ntubes, i, k)]) < 1e-4, (i, k)


POST_CASES = {"all": dict(evaluate_topk=-1, topk=-1), "top20": dict(evaluate_topk=1, topk=20), "topm1": dict(evaluate_topk=5, topk=-1)}


def postprocess_fixture_history(g):
    nums = [int(v) for v in g["nums"]]
    hist = []
    for i in range(3):
        loc = g["hist%d_loc" % i]
        hist.append({"pred_prob": np.repeat(g["hist%d_prob" % i][:, None, :], loc.shape[1], axis=1), "pred_loc": loc, "tubes_nums": nums})
    return hist


@pytest.mark.parametrize("tag", sorted(POST_CASES))
def test_postprocess_restatement_matches_the_reference_loop(golden, tag):
    """oracle/postprocess_ref.py against what the reference's own evaluation loop (test.py:157-210, executed by
    oracle/make_golden.py on a seeded history) wrote: same rows, same order, bit-identical boxes and scores, and the same
    CSV text."""
    from oracle import postprocess_ref as P
    g = golden("postprocess_golden")
    rows = P.postprocess(postprocess_fixture_history(g), **POST_CASES[tag])
    meta, box, score = g[tag + "_meta"], g[tag + "_box"], g[tag + "_score"]
    assert len(rows) == len(meta)
    got_meta = np.asarray([[r[0], r[1], r[2]] for r in rows], np.int32)
    assert np.array_equal(got_meta, meta)
    assert np.array_equal(np.asarray([r[3:7] for r in rows], np.float32), box)
    assert np.array_equal(np.asarray([r[7] for r in rows], np.float32), score)
    lines = [P.csv_line("vid%d" % r[1], 900 + r[1], r[3:7], r[2] + 1, r[7]) for r in rows]
    assert lines == [str(x) for x in g[tag + "_lines"]]
    if tag == "topm1":                                   # the reference's `[:args.topk]` with topk = -1 drops the last row
        full = P.postprocess(postprocess_fixture_history(g), evaluate_topk=-1, topk=-1)
        assert len(rows) == len(full) - 9


def test_i3d_classifier_restatement_matches_the_reference_golden(golden):
    """oracle/i3d_ref.i3d_forward (models/i3dpt.py:236-262) against what the reference's own I3D module returned for the
    closed-form weights and clip (tests/golden/i3d_classifier_golden.npz, `python -m oracle.make_golden i3d`)."""
    import torch

    from oracle import i3d_ref as R

    g = golden("i3d_classifier_golden")
    shapes = R.i3d_shapes(24)
    assert sorted(shapes) == [str(k) for k in g["keys"]]
    assert [str(tuple(shapes[k])) for k in sorted(shapes)] == [str(v) for v in g["shapes"]]
    x = R.fill_tensor("golden.i3dcls.clip", (1, 3, 16, 224, 224), "image")
    torch.set_num_threads(8)
    with torch.no_grad():
        prob, logits = R.i3d_forward(x, R.fill_state_dict(shapes, "i3dcls."))
    assert float(np.abs(logits.numpy() - g["logits"]).max()) < 1e-4 * float(np.abs(g["logits"]).max())
    assert float(np.abs(prob.numpy() - g["prob"]).max()) < 1e-5


def test_roi_pool_restatement_against_an_independent_implementation():
    """The reference has no CPU ROIPool (ROIPool.h:47,68 raise on CPU tensors), so the C restatement of
    cuda/ROIPool_cuda.cu:40-132 cannot be pinned by reference outputs.  Second-best pin: the algorithm it restates (Fast
    R-CNN RoI max pooling: bin p covers [floor(p*s), ceil((p+1)*s)) of the RoI, s = roi_size / pooled_size) is what torch's
    adaptive_max_pool2d computes on the cropped RoI (an independent implementation); for in-bounds RoIs with integer
    corners at spatial_scale 1 -- where the float bin arithmetic has no rounding ambiguity (sizes that divide or are
    dyadic multiples of 7 excluded from nothing: both sides use floor / ceil of the same rational) -- values AND argmax
    positions must agree; the backward is then checked as the exact adjoint (scatter of the argmax)."""
    import torch
    import torch.nn.functional as F

    import oracle

    rs = np.random.RandomState(17)
    B, C, H, W = 2, 5, 23, 31
    x = rs.randn(B, C, H, W).astype(np.float32)
    rois, crops = [], []
    for _ in range(40):
        b = rs.randint(0, B)
        x1, y1 = rs.randint(0, W - 8), rs.randint(0, H - 8)
        x2, y2 = rs.randint(x1 + 6, W), rs.randint(y1 + 6, H)          # at least 7 cells a side: no empty bins
        rois.append([b, x1, y1, x2, y2])
    rois = np.asarray(rois, np.float32)
    out, arg = oracle.roi_pool_forward(x, rois, (7, 7), 1.0)
    for k, (b, x1, y1, x2, y2) in enumerate(rois.astype(int)):
        crop = torch.from_numpy(x[b:b + 1, :, y1:y2 + 1, x1:x2 + 1])
        ref, idx = F.adaptive_max_pool2d(crop, (7, 7), return_indices=True)
        assert np.array_equal(out[k], ref[0].numpy()), k
        cw = x2 - x1 + 1
        iy, ix = idx[0].numpy() // cw + y1, idx[0].numpy() % cw + x1      # crop index -> map position
        assert np.array_equal(arg[k], (iy * W + ix).astype(np.int32)), k
    g = rs.randn(*out.shape).astype(np.float32)
    gin = oracle.roi_pool_backward(g, arg, rois, (7, 7), (B, C, H, W))
    ref = np.zeros((B, C, H, W), np.float64)
    for k in range(len(rois)):
        b = int(rois[k, 0])
        for c in range(C):
            np.add.at(ref[b, c].reshape(-1), arg[k, c].reshape(-1), g[k, c].reshape(-1).astype(np.float64))
    assert np.abs(gin - ref).max() < 1e-5


def test_roi_pool_restatement_matches_the_hand_derived_vectors():
    """ROIPool has no CPU implementation in the reference (ROIPool.h:47,68); the C restatement is pinned by 12 cases worked out by
    hand from ROIPool_cuda.cu:40-132 (tests/golden/make_roipool_hand_vectors.py: bin tables with the deciding source line each) --
    forward value, argmax and the backward scatter, bit for bit."""
    import json
    vec = json.load(open(os.path.join(GOLDEN, "roipool_hand_vectors.json")))
    assert len(vec) >= 12
    for v in vec:
        x = np.array(v["feature"], np.float32)
        rois = np.array(v["rois"], np.float32)
        out, arg = oracle.roi_pool_forward(x, rois, tuple(v["pooled"]), v["spatial_scale"])
        assert np.array_equal(out, np.array(v["out"], np.float32)), (v["name"], v["decided_by"])
        assert np.array_equal(arg, np.array(v["argmax"], np.int32)), (v["name"], v["decided_by"])
        gin = oracle.roi_pool_backward(np.array(v["grad_out"], np.float32), arg, rois, tuple(v["pooled"]), x.shape)
        assert np.array_equal(gin, np.array(v["grad_in"], np.float32)), v["name"]
