"""Host-side helpers of the product layer against the oracle's restatement (no GPU)."""
import os

import numpy as np
import pytest

from oracle import i3d_ref as R
from step_amd.tube_math import generate_anchors


def test_generate_anchors_matches_oracle():
    a = generate_anchors()
    assert a.shape == (34, 4)                          # 9 + 25 boxes of anchor mode "1" (data/data_utils.py:19-45)
    assert np.array_equal(a, R.anchors())
    assert (a >= 0).all() and (a <= 1 + 1e-6).all()


# ---------------------------------------------------------------- training sample selection
def _selection_args(g, ci):
    from types import SimpleNamespace as NS
    return NS(T=3, NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3}, max_iter=3, num_classes=60, image_size=(400, 400),
              cls_thresh=[0.2, 0.35, 0.5], reg_thresh=[0.2, 0.35, 0.5], max_pos_num=int(g["c%d_max_pos_num" % ci]),
              neg_ratio=int(g["c%d_neg_ratio" % ci]), topk=int(g["c%d_topk" % ci]),
              selection_sampling=str(g["c%d_sampling" % ci]), temporal_mode=str(g["c%d_mode" % ci]))


@pytest.mark.parametrize("ci", range(6))
def test_train_select_matches_reference(ci):
    """step_amd.selection.train_select against what the reference's train_select (utils/utils.py:135-339) returned for the
    same inputs and the same `random` / `numpy.random` seeds (oracle/make_golden.py selection): same tubes selected, in the
    same order, bit for bit, with the same target rows -- all three steps (initial proposals; refined tubes; extended tubes
    with neighbour targets), softmax / random / uniform sampling, top-k, score ties, an invalid box; temporal modes predict,
    mean and extrapolate."""
    import random

    from step_amd import selection as S
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "selection_golden.npz"))
    args = _selection_args(g, ci)
    seed = int(g["c%d_seed" % ci])
    targets = [g["c%d_targets%d" % (ci, b)] for b in range(2)]
    anchors = (generate_anchors() * 400.0).astype(np.float32)
    tubes = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
    hist = {k: g["c%d_hist_%s" % (ci, k)] for k in ("pred_loc", "pred_first_loc", "pred_last_loc")}
    hist["pred_prob"] = np.tile(g["c%d_hist_pred_prob" % ci], (1, 3, 1))
    hist["tubes_nums"] = [34, 20]
    for step in (1, 2, 3):
        random.seed(seed * 10 + step)
        np.random.seed(seed * 10 + step)
        sel, tgt = S.train_select(step, hist if step > 1 else None, targets, tubes, args)
        for b in range(2):
            ref_sel, ref_tgt = g["c%d_s%d_sel%d" % (ci, step, b)], g["c%d_s%d_tgt%d" % (ci, step, b)]
            assert sel[b].shape == ref_sel.shape and sel[b].dtype == ref_sel.dtype, (step, b)
            assert np.array_equal(sel[b], ref_sel), (step, b)
            assert np.array_equal(tgt[b], ref_tgt), (step, b)
    assert np.array_equal(S.tube_iou(targets[0][:, :, :4], hist["pred_loc"][:9]), g["c%d_iou" % ci], equal_nan=True)   # 0/0 pairs: nan in both


def test_tube_iou_edge_cases_match_reference():
    """padding tubes (all zero) score 0 against everything, touching boxes 0, degenerate pairs nan -- compute_tube_iou
    (utils/tube_utils.py:308-351) bit for bit."""
    from step_amd import selection as S
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "selection_golden.npz"))
    out = S.tube_iou(g["edge_t1"], g["edge_t2"])
    assert np.array_equal(out, g["edge_iou"], equal_nan=True)
    with pytest.raises(AssertionError):
        S.tube_iou(np.zeros((1, 2, 4), np.float32), np.zeros((1, 3, 4), np.float32))


@pytest.mark.parametrize("ei", range(6))
def test_select_proposals_edge_cases_match_reference(ei):
    """select_proposals (utils/utils.py:341-423) alone, same seeds: more ground truths than proposals, duplicate proposals,
    nothing above the threshold, neg_ratio 0, given / IoU-derived scores, the three sampling modes -- same (gt, proposal)
    pairs in the same order, same IoU table."""
    import random

    from step_amd import selection as S
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "selection_golden.npz"))
    thr, max_pos, neg_ratio = g["e%d_cfg" % ei]
    sc = g["e%d_scores" % ei] if ("e%d_scores" % ei) in g.files else None
    random.seed(500 + ei)
    np.random.seed(500 + ei)
    pos, neg, ious = S.select_proposals(g["e%d_gt" % ei], g["e%d_an" % ei], sc, float(thr), int(max_pos), str(g["e%d_sampling" % ei]),
                                        int(neg_ratio))
    assert np.array_equal(np.asarray(pos, np.int64).reshape(-1, 2), g["e%d_pos" % ei])
    assert np.array_equal(np.asarray(neg, np.int64).reshape(-1, 2), g["e%d_neg" % ei])
    assert np.array_equal(ious, g["e%d_ious" % ei], equal_nan=True)


def test_flat_adam_refuses_what_it_cannot_run():
    """No CPU fallback and no silent approximation: CPU parameters, non-fp32 masters and per-group betas raise."""
    import torch
    from step_amd import FlatAdam
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FlatAdam([torch.nn.Parameter(torch.zeros(8))], lr=1e-3)
    with pytest.raises(RuntimeError, match="fp32 master"):
        FlatAdam([torch.nn.Parameter(torch.zeros(8, dtype=torch.bfloat16))], lr=1e-3)
    a, b = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))
    with pytest.raises(ValueError, match="betas"):
        FlatAdam([{"params": [a]}, {"params": [b], "betas": (0.5, 0.9)}], lr=1e-3)


def test_extrapolate_tubes_matches_reference():
    """tube_utils.extrapolate_tubes (utils/tube_utils.py:159-176), numpy and torch forms, bit for bit -- including the clamps
    to [0, 399] the reference applies whatever the image size."""
    import torch
    from step_amd.tube_math import extrapolate_tubes
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "inference_modes_golden.npz"))
    t = g["ext_in"]
    assert np.array_equal(extrapolate_tubes(t.copy(), 3), g["ext_out_T3"])
    assert np.array_equal(extrapolate_tubes(np.tile(t, (1, 2, 1)), 6), g["ext_out_T6"])
    assert np.array_equal(extrapolate_tubes(torch.from_numpy(t), 3).numpy(), g["ext_out_T3"])
    assert np.array_equal(extrapolate_tubes(torch.from_numpy(np.tile(t, (1, 2, 1))), 6).numpy(), g["ext_out_T6"])
    r = g["ext_out_T3"]                                          # only the near corner is clamped below, the far one above
    assert r[..., :2].min() == 0.0 and r[..., 2:].max() == 399.0


def test_tube_iou_oracle_pinned_and_array_version_agrees():
    """oracle/selection_ref.py (scalar restatement) reproduces the reference's recorded compute_tube_iou outputs; the array
    version the product uses agrees with it bit for bit on random tubes (overlapping, disjoint, touching, padding, degenerate)."""
    from oracle import selection_ref as SR
    from step_amd import selection as S
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "selection_golden.npz"))
    for ci in range(int(g["n_cases"])):
        a, b = g["c%d_targets0" % ci][:, :, :4], g["c%d_hist_pred_loc" % ci][:9]
        assert np.array_equal(SR.tube_iou(a, b), g["c%d_iou" % ci], equal_nan=True)
    assert np.array_equal(SR.tube_iou(g["edge_t1"], g["edge_t2"]), g["edge_iou"], equal_nan=True)
    rs = np.random.RandomState(3)
    for _ in range(40):
        n1, n2, T = rs.randint(1, 6), rs.randint(1, 8), rs.randint(1, 5)
        def tubes(n):
            xy = rs.uniform(-20, 300, (n, T, 2)); wh = rs.uniform(-5, 160, (n, T, 2))
            t = np.concatenate([xy, xy + wh], 2).astype(np.float32)
            t[rs.rand(n) < 0.15] = 0                                 # padding tubes
            t[rs.rand(n, T) < 0.1] = 0                               # padding frames
            return np.round(t) if rs.rand() < 0.3 else t            # integer coordinates: exact ties / touching boxes
        a, b = tubes(n1), tubes(n2)
        if rs.rand() < 0.3:
            b[0] = a[0]
        assert np.array_equal(S.tube_iou(a, b), SR.tube_iou(a, b), equal_nan=True)


def test_tube_math_matches_reference_helpers():
    """step_amd.tube_math against the reference's own utils/tube_utils.py helpers on random inputs (tube_math_golden.npz):
    valid_tubes numpy + torch forms (clamp, then boxes not wider AND taller than 2 px become the whole image -- strict <),
    get_center_size / encode_coef / decode_coef (+1 pixel sizes, -1 on the far corner), flatten_tubes with and without the
    frame-index column and with an empty clip.  Bit for bit."""
    import torch
    from step_amd import tube_math as TM
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tube_math_golden.npz"))
    t = g["vt_in"]
    assert np.array_equal(TM.valid_tubes(t.copy(), 400, 400), g["vt_np"])
    assert np.array_equal(TM.valid_tubes(t.copy(), 320, 240), g["vt_np_320x240"])
    assert np.array_equal(TM.valid_tubes(torch.from_numpy(t.copy()), 400, 400).numpy(), g["vt_torch"])
    assert np.array_equal(g["vt_np"], g["vt_torch"])
    boxes, gt, deltas = (torch.from_numpy(g[k]) for k in ("boxes", "gt", "deltas"))
    assert np.array_equal(np.stack([v.numpy() for v in TM.get_center_size(boxes)]), g["center_size"])
    assert np.array_equal(TM.encode_coef(gt, boxes).numpy(), g["encode"])
    assert np.array_equal(TM.decode_coef(boxes, deltas).numpy(), g["decode"])
    tl = [g["flat_in%d" % i] for i in range(3)]
    for flag in (False, True):
        flat, nums = TM.flatten_tubes([x.copy() for x in tl], batch_idx=flag)
        assert np.array_equal(flat, g["flat_%d" % flag]) and list(nums) == list(g["flat_nums_%d" % flag])
    for mode, n in (("1", 34), ("2", 59), ("3", 84), ("4", 109)):            # the four --anchor_mode grids (data/ava.py:342-354)
        at = TM.anchor_tubes(mode, T=3)
        assert at.shape == (n, 3, 4) and at.dtype == np.float32
        assert np.array_equal(at[:, 0], g["anchors_mode%s" % mode]) and np.array_equal(at[:, 2], at[:, 0])
    assert TM.anchor_tubes("0", T=9).shape == (1, 9, 4) and not TM.anchor_tubes("0", T=9).any()


def test_deepcopy_owns_its_parameters_and_data_parallel_replicas_own_their_helpers():
    """copy.deepcopy(net): every ConvUnit of the copy must read the COPY's parameters (a closure over the original module
    would keep computing with the original's weights); nn.DataParallel replication is refused loudly."""
    import copy
    from types import SimpleNamespace as NS

    import torch

    import step_amd
    from step_amd import backbone

    cfg = NS(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False, fc_dim=256, pool_size=7,
             dropout=0.0, num_classes=60, cls_thresh=0.0, reg_thresh=0.0, max_pos_num=5, neg_ratio=2, NUM_SAMPLE=-1,
             topk=300, evaluate_topk=-1, T=3, iterative_mode="spatial", anchor_mode="1", temporal_mode="predict",
             pool_mode="align", scale_norm=2, det_net="two_branch", no_context=False, cls_only=False)
    nets = [step_amd.BaseNet(cfg)]
    try:
        nets.append(step_amd.TwoBranchNet(cfg))
    except Exception:                                            # (cfg fields of the head differ: the backbone check stands alone)
        pass
    for net in nets:
        c = copy.deepcopy(net)
        n_units = 0
        for (_, m), (_, mo) in zip(c.named_modules(), net.named_modules()):
            for k, u in vars(m).items():
                if isinstance(u, backbone.ConvUnit) and isinstance(u.owner, torch.nn.Module):
                    n_units += 1
                    assert u.owner is m, k
                    w = u.weight_fn()
                    mine = {id(p) for p in m.parameters()}
                    theirs = {id(p) for p in mo.parameters()}
                    if id(w) in mine or id(w) in theirs:         # (a torch.cat of parameters is a temporary)
                        assert id(w) in mine and id(w) not in theirs, k
                    if u.bn is not None:
                        assert any(u.bn is sub for sub in m.modules())
        assert n_units >= 10
    # torch.nn.parallel.replicate() calls this hook on every submodule: the replica gets helpers of its own, bound to itself and
    # versioned by the original (the replicated forward itself: tests/module_cases.py::case_data_parallel_replicas)
    for m in nets[0].modules():
        if isinstance(m, (backbone.Unit3D, backbone.Mixed)):
            r = m._replicate_for_data_parallel()
            for k, u in vars(m).items():
                if isinstance(u, (backbone.ConvUnit, backbone._FusedPointwise)):
                    ru = vars(r)[k]
                    assert ru is not u and ru.owner is r and ru._src is u


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_bench_rank_of_an_eight_gpu_launch_parses_its_arguments_and_environment(config):
    """The driver's scaling run is `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps K --warmup W`: every rank
    reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Without a GPU a rank of that launch must get exactly as far as the device check
    (arguments parsed, world size matched against --gpus, no spawn of its own), and a world size that contradicts --gpus must be named."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="3", LOCAL_RANK="3", WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--config", config] + (["--dtype", "bf16", "--select"] if config == "c4" else [])
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "needs a ROCm device" in (r.stderr + r.stdout), r.stderr[-800:]
    env["WORLD_SIZE"] = "4"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "--gpus 8 but WORLD_SIZE=4" in (r.stderr + r.stdout), r.stderr[-800:]
