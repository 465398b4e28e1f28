"""Module-level parity cases (BaseNet / Mixed / ContextNet / TwoBranchNet / ROINet / driver) against
the golden vectors produced by the reference (tests/golden) and the torch-CPU oracle.  `dev` is the
torch device the product modules run on: "cpu" under the test-only interpreter patch, "cuda" on the
GPU box."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import torch

import oracle
import step_amd
from oracle import i3d_ref as R
from step_amd import backbone, heads
from step_amd.driver import inference

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cfg(**kw):
    base = dict(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False, T=3,
                num_classes=60, fc_dim=256, dropout=0.0, pool_size=7, no_context=False, max_iter=3,
                NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3}, temporal_mode="predict", image_size=(400, 400), pool_mode="align")
    base.update(kw)
    return NS(**base)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def fill(mod, tag=""):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    mod.load_state_dict(R.fill_state_dict(shapes, tag))
    return mod


def np_(t):
    return t.detach().float().cpu().contiguous().numpy()


def record(name, value):
    """Keep a measured error level next to the GPU run's other outputs (gpurun_out/test_metrics.json; ignored elsewhere)."""
    import json
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ""), "gpurun_out")
    if not os.environ.get("GRAFT_REPO_ROOT") or not os.path.isdir(d):
        return
    f = os.path.join(d, "test_metrics.json")
    m = json.load(open(f)) if os.path.exists(f) else {}
    m[name] = value
    json.dump(m, open(f, "w"), indent=1, sort_keys=True)


def case_state_dict_contract(dev, golden):
    info = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    nets = {"BaseNet": step_amd.BaseNet(cfg()), "ContextNet": step_amd.ContextNet(cfg()),
            "TwoBranchNet": step_amd.TwoBranchNet(cfg()), "TwoBranchNet_cls_only": step_amd.TwoBranchNet(cfg(), cls_only=True)}
    for name, net in nets.items():
        sd = net.state_dict()
        assert list(sd.keys()) == list(info[name].keys()) or set(sd) == set(info[name]), name
        for k, v in sd.items():
            assert list(v.shape) == info[name][k], (name, k)
    assert sorted(k for k, p in nets["BaseNet"].named_parameters() if p.requires_grad) == sorted(info["BaseNet_trainable"])
    assert sorted(k for k, p in nets["TwoBranchNet"].named_parameters() if p.requires_grad) == sorted(info["TwoBranchNet_trainable"])
    # reference's train() override keeps BN in eval (networks.py:85-99)
    nets["BaseNet"].train()
    assert all(not m.training for m in nets["BaseNet"].modules() if isinstance(m, torch.nn.BatchNorm3d))


def _replicate_like_data_parallel(net, n=2):
    """What torch.nn.parallel.replicate does on every nn.DataParallel forward (torch/nn/parallel/replicate.py: each module's
    _replicate_for_data_parallel(), children re-pointed to the replicas, parameters handed over as NON-LEAF copies set as plain
    attributes, buffers as copies) -- with the copies on the module's own device, so that it runs on the CPU interpreter and on a
    one-GPU box (the real replicate() needs one CUDA device per replica)."""
    modules = list(net.modules())
    index = {m: i for i, m in enumerate(modules)}
    copies = [[m._replicate_for_data_parallel() for m in modules] for _ in range(n)]
    for row in copies:
        for r in row:
            r._former_parameters = {}
    for i, m in enumerate(modules):
        for key, child in m._modules.items():
            for j in range(n):
                copies[j][i]._modules[key] = None if child is None else copies[j][index[child]]
        for key, p in m._parameters.items():
            for j in range(n):
                r = copies[j][i]
                pc = None if p is None else (p.clone() if p.requires_grad else p.detach().clone())     # Broadcast.apply: differentiable
                setattr(r, key, pc)
                if pc is not None:
                    r._former_parameters[key] = pc
        for key, b in m._buffers.items():
            for j in range(n):
                setattr(copies[j][i], key, None if b is None else b.detach().clone())
    return [c[0] for c in copies]


class _TinyNet(torch.nn.Module):
    """stem-free slice of the backbone: Unit3D 3x3x3 -> Mixed -> pool (every helper kind a BaseNet holds)"""

    def __init__(self):
        super().__init__()
        self.a = backbone.Unit3D(8, 24, (3, 3, 3))
        self.b = backbone.Mixed(24, [8, 12, 16, 4, 8, 8])
        self.c = backbone.MaxPoolTF((1, 3, 3), (1, 2, 2))

    def forward(self, x):
        return self.c(self.b(self.a(x)))


def case_data_parallel_replicas(dev, golden):
    """nn.DataParallel compatibility (train.py:142-144, test.py:82-84, demo.py:79-81 wrap base_net / context_net): replicas made
    the way torch.nn.parallel.replicate makes them compute on THEIR tensors (not the original's), reuse a device's packed weights
    from one forward's replicas to the next, follow in-place weight updates of the original, and route gradients back to the
    original's parameters."""
    from step_amd import ops
    net = fill(_TinyNet(), "dp.").to(dev).eval()
    x = R.fill_tensor("dp.x", (2, 8, 3, 9, 10), "feat").permute(0, 2, 3, 4, 1).contiguous().to(dev)
    packs = []
    orig = ops.pack_conv_weight
    ops.pack_conv_weight = lambda *a, **k: (packs.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            full = net(x)
            n0 = len(packs)
            for rnd in range(2):
                reps = _replicate_like_data_parallel(net)
                ys = [r(x[j:j + 1]) for j, r in enumerate(reps)]
                assert torch.equal(torch.cat(ys), full), rnd
                # helpers are bound to the replica, not shared with the original
                assert reps[0].a._unit is not net.a._unit and reps[0].a._unit.owner is reps[0].a and reps[0].b._fused.owner is reps[0].b
            assert len(packs) == n0, (n0, len(packs))                 # same device, same versions: every image came from the cache
            # the replica's unit reads the replica's tensors (another device's copy under the real DataParallel), versions come from the original
            u = reps[1].a._unit
            assert u.weight_fn() is reps[1].a.conv3d.weight and u.weight_fn() is not net.a.conv3d.weight and u.bn is reps[1].a.batch3d
            assert u._wver(u.weight_fn())[0][:2] == (net.a.conv3d.weight.data_ptr(), net.a.conv3d.weight._version)
            # ... and an in-place update of the ORIGINAL (an optimizer step) reaches the next forward's replicas
            net.a.conv3d.weight.mul_(1.5)
            net.b.branch_1[1].conv3d.weight.add_(0.05)
            full2 = net(x)
            assert not torch.equal(full2, full)
            reps = _replicate_like_data_parallel(net)
            assert torch.equal(torch.cat([r(x[j:j + 1]) for j, r in enumerate(reps)]), full2)
    finally:
        ops.pack_conv_weight = orig
    # gradients: the replica's output depends on the original's parameters through the broadcast copies
    for p in net.parameters():                                        # (BN stays in eval mode, as BaseNet.train() keeps it)
        p.grad = None
    g = R.fill_tensor("dp.g", tuple(full.shape), "feat").to(dev)
    (net(x) * g).sum().backward()
    want = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    assert len(want) >= 6
    for p in net.parameters():
        p.grad = None
    reps = _replicate_like_data_parallel(net)
    sum((r(x[j:j + 1]) * g[j:j + 1]).sum() for j, r in enumerate(reps)).backward()
    for k, p in net.named_parameters():
        if k in want:
            assert p.grad is not None and rel(np_(p.grad), np_(want[k])) < 1e-4, (k, None if p.grad is None else rel(np_(p.grad), np_(want[k])))
    # torch's own wrapper: one visible device (or none) -> DataParallel calls the module itself; with a GPU also the real
    # replicate() / parallel_apply() path on device ids [0, 0]
    net.eval()
    with torch.no_grad():
        ref = net(x)
        dp = torch.nn.DataParallel(net)
        assert torch.equal(dp(x), ref)
        if str(dev).startswith("cuda"):
            try:
                dp2 = torch.nn.DataParallel(net, device_ids=[0, 0])
                y2 = dp2(x)
            except Exception as e:                                    # (a torch build that refuses duplicate device ids)
                y2 = None
                record("data_parallel_dup_ids", "refused: %s" % type(e).__name__)
            if y2 is not None:
                assert torch.equal(y2, ref)
                record("data_parallel_dup_ids", "ran")


def case_mixed_golden(dev, golden):
    g = golden("ops_golden")
    mx = fill(backbone.Mixed(24, [8, 12, 16, 4, 8, 8]), "golden.mixed.").to(dev).eval()
    x = R.fill_tensor("golden.mixed.in", (2, 24, 3, 9, 7), "feat").permute(0, 2, 3, 4, 1).contiguous().to(dev)
    with torch.no_grad():
        y = mx(x)
    assert rel(np_(y.permute(0, 4, 1, 2, 3)), g["mixed_out"]) < 1e-3
    assert rel(np_(y.permute(0, 4, 1, 2, 3)), g["mixed_out"]) < 5e-5


def case_basenet_c1_golden(dev, golden):
    """C1: [1,8,3,112,112] fp32 through the whole backbone vs the reference's own output."""
    g = golden("i3d_c1_golden")
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    x = R.fill_tensor("golden.c1.images", (1, 8, 3, 112, 112), "image").to(dev)
    stages = []
    hooks = [m.register_forward_hook(lambda _m, _i, o: stages.append(o)) for m in net.base_model]
    with torch.no_grad():
        y = net(x)
    for h in hooks:
        h.remove()
    assert tuple(y.shape) == (1, 2, 832, 7, 7)
    for i, s in enumerate(stages):
        ncdhw = s.permute(0, 4, 1, 2, 3)
        assert list(ncdhw.shape) == list(g["stage%d_shape" % i]), i
        step = int(g["stage%d_stats" % i][3])
        samp = np_(ncdhw).reshape(-1)[::step][:256]
        assert rel(samp, g["stage%d_sample" % i]) < 1e-3, (i, rel(samp, g["stage%d_sample" % i]))
    err = rel(np_(y), g["conv_feat"])
    assert err < 1e-3, err
    assert err < 1e-4, err          # in practice ~1e-6: fp32 MFMA is an exact FMA chain


def case_basenet_c1_16bit_error(dev, golden):
    """bf16 / fp16 storage (fp32 accumulate) through all 45 conv units: measured error against the
    reference's fp32 output.  8-bit-mantissa storage cannot meet the 1e-3 bar of the fp32 path;
    the bound asserted here is what 45 layers of one-rounding-per-layer give in practice."""
    g = golden("i3d_c1_golden")
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    x = R.fill_tensor("golden.c1.images", (1, 8, 3, 112, 112), "image").to(dev)
    # measured on MI355X: bf16 5.0e-3 (BENCH_r01 smoke line), fp16 below 1e-3; the bounds sit just above the measured level
    for dt, bound in ((torch.bfloat16, 8e-3), (torch.float16, 2e-3)):
        with torch.no_grad():
            y = net(x.to(dt))
        assert y.dtype == dt
        e = rel(np_(y), g["conv_feat"])
        record("basenet_c1_rel_err_%s" % str(dt).split(".")[-1], e)
        assert e < bound, (dt, e)


def case_conv_pool_fusion_in_basenet(dev, golden):
    """backbone.FUSE_CONV_POOL: maxPool3d_3a taken on conv3d_2c's tiles (ops.conv_forward_pre_pool, models/i3dpt.py:207-212) inside
    BaseNet.forward -- a clip whose 24 x 24 stage-2 maps the planner tiles 4 x 8 x 8 (as the 56 x 56 maps of C2): the fused call is
    really taken (ops.PROFILE sees its kernel and no stand-alone (1,3,3) pool behind conv3d_2c) and the network output is BIT-IDENTICAL
    to the run with the fusion off."""
    from step_amd import backbone as _bb
    from step_amd import ops as _ops
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    # (stage-2 maps 40 x 40 x 16 planes: 100 tiles of 4 x 8 x 8 -- enough for the 8-wave two-phase form the fusion lives in; on the
    # interpreter a smaller clip with the 8-wave form forced)
    shape = (1, 32, 3, 160, 160) if dev != "cpu" else (1, 16, 3, 96, 96)
    x = R.fill_tensor("golden.fuse.images", shape, "image").to(dev).to(torch.bfloat16)
    assert _bb.FUSE_CONV_POOL and _bb.FUSE_POINTWISE_INPUT
    with torch.no_grad():
        _ops.PROFILE, _ops.PROFILE_LIMIT = [], 1 << 30              # (record the launches, time nothing)
        try:
            y1 = net(x).clone()
            names = [r[0] for r in _ops.PROFILE]
        finally:
            _ops.PROFILE, _ops.PROFILE_LIMIT = None, None
        assert any("conv_tap_pre_pool_kernel" in n or "conv_tap_pre_pool_persist_kernel" in n for n in names), names
        assert sum("maxpool_sep_kernel" in n and " 1, 3, 3, 1, 2, 2" in n for n in names) == 0, names     # 2a rides in the stem, 3a in conv3d_2c
        try:
            _bb.FUSE_CONV_POOL = False
            y0 = net(x)
        finally:
            _bb.FUSE_CONV_POOL = True
    assert tuple(y1.shape) == ((1, 8, 832, 10, 10) if dev != "cpu" else (1, 4, 832, 6, 6))
    assert torch.equal(y1, y0), float((y1.float() - y0.float()).abs().max())


def case_basenet_forward_u8(dev, golden):
    """BaseNet.forward_u8 (uint8 frames [N,T,H,W,3]: the reference's ConvertFromInts / SubtractMeans / DivideStds, data/augmentations.py:68-111,
    inside the stem's frame staging, ops.stem_pool_forward_u8) is BIT-IDENTICAL to forward(clip_from_u8(frames)) -- default normalisation
    and a per-channel mean / std, bf16 and fp16 -- and really takes the uint8 stem (ops.PROFILE); fp32 falls back to the conversion pass."""
    from step_amd import backbone as _bb
    from step_amd import ops as _ops
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    g = torch.Generator().manual_seed(5)
    fr = torch.randint(0, 256, (2, 16, 64, 80, 3), dtype=torch.uint8, generator=g).to(dev)
    keep_flag = _bb.FUSE_STEM_U8
    with torch.no_grad():
        y_conv = net.forward_u8(fr, torch.bfloat16)               # the default form: conversion pass + forward
        assert torch.equal(y_conv, net(_ops.clip_from_u8(fr, torch.bfloat16)))
    _bb.FUSE_STEM_U8 = True                                       # opt-in form: the stem stages the uint8 frames itself
    try:
        _basenet_forward_u8(dev, net, fr, _ops)
    finally:
        _bb.FUSE_STEM_U8 = keep_flag


def _basenet_forward_u8(dev, net, fr, _ops):
    with torch.no_grad():
        for dt, kw in ((torch.bfloat16, {}), (torch.float16, dict(scale=1, mean=(0.4, 0.45, 0.5), std=(0.25, 0.2, 0.3)))):
            want = net(_ops.clip_from_u8(fr, dt, kw.get("scale", 2), kw.get("mean", (0.0, 0.0, 0.0)), kw.get("std", (1.0, 1.0, 1.0))))
            _ops.PROFILE, _ops.PROFILE_LIMIT = [], 1 << 30
            try:
                got = net.forward_u8(fr, dt, **kw)
                names = [r[0] for r in _ops.PROFILE]
            finally:
                _ops.PROFILE, _ops.PROFILE_LIMIT = None, None
            if dev != "cpu":                                       # (the fused forms live behind is_cuda checks: on the interpreter forward_u8 is the conversion pass + forward)
                assert any("stem_stream_kernel" in n and "true, true, true>" in n for n in names), names[:3]
            assert got.dtype == dt and torch.equal(got, want), (dt, float((got.float() - want.float()).abs().max()))
        y32 = net.forward_u8(fr[:1, :8], torch.float32)
        assert y32.dtype == torch.float32 and rel(np_(y32), np_(net(_ops.clip_from_u8(fr[:1, :8], torch.float32)))) == 0.0


def case_i3d_classifier_golden(dev, golden):
    """step_amd.I3D (the full Kinetics classifier, models/i3dpt.py:175-262): state_dict keys / shapes of the reference's
    module, and forward on the golden clip against what the reference returned (fp32: 1e-3; bf16: the argmax and a loose bound)."""
    g = golden("i3d_classifier_golden")
    net = step_amd.I3D(num_classes=24, dropout_prob=0.5)
    sd = net.state_dict()
    assert sorted(sd) == [str(k) for k in g["keys"]]
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == [str(v) for v in g["shapes"]]
    net.load_state_dict(R.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, "i3dcls."))
    net = net.to(dev).eval()
    x = R.fill_tensor("golden.i3dcls.clip", (1, 3, 16, 224, 224), "image").to(dev)
    with torch.no_grad():
        prob, logits = net(x)
    assert tuple(prob.shape) == (1, 24)
    assert rel(np_(logits), g["logits"]) < 1e-3, rel(np_(logits), g["logits"])
    assert float(np.abs(np_(prob) - g["prob"]).max()) < 1e-4
    with torch.no_grad():
        pb, lb = net(x.to(torch.bfloat16))
    assert rel(np_(lb), g["logits"]) < 3e-2 and int(pb.argmax()) == int(g["prob"].argmax())


def case_context_golden(dev, golden):
    g = golden("head_golden")
    net = fill(step_amd.ContextNet(cfg())).to(dev).eval()
    cf = R.fill_tensor("golden.ctx.feat", (1, 3, 832, 25, 25), "feat").to(dev)
    with torch.no_grad():
        y = net(cf)
    assert tuple(y.shape) == (1, 1024, 3, 1, 1)
    assert rel(np_(y), g["context_out"]) < 1e-4


def _twobranch(dev, golden, tl):
    g = golden("head_golden")
    net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev).eval()
    net.set_device(dev)
    pf = R.fill_tensor("golden.det.pooled%d" % tl, (2, tl, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx%d" % tl, (2, 1024, tl, 1, 1), "feat").to(dev)
    with torch.no_grad():
        o = net(pf, context_feat=cx)
    for nme, t in zip(("prob", "loc", "first", "last"), o[:4]):
        e = rel(np_(t), g["det_T%d_%s" % (tl, nme)])
        assert e < 1e-3, (tl, nme, e)
    return net, pf, cx


def case_twobranch_T3_and_losses_golden(dev, golden):
    g = golden("head_golden")
    net, pf, cx = _twobranch(dev, golden, 3)
    with torch.no_grad():
        o = net(pf, context_feat=cx, tubes=torch.from_numpy(g["loss_tubes"]).to(dev), targets=torch.from_numpy(g["loss_targets"]).to(dev))
    assert rel(np_(o[4]), g["loss_cls"]) < 1e-3
    assert rel(np_(o[5]), g["loss_loc"]) < 1e-3
    assert rel(np_(o[6]), g["loss_nbr"]) < 1e-3
    assert o[4].shape == (120,) and o[5].shape == (1,) and o[6].shape == (1,)


def case_twobranch_variants_golden(dev, golden):
    """The head's two other configurations against the reference: cls_only=True (no regressors: the three location outputs
    are the reference's (1,)-shaped zero placeholders) and no_context=True (global_cls without the ContextNet feature)."""
    g = golden("head_variants_golden")
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    for tag, net, kw in (("cls_only", step_amd.TwoBranchNet(cfg(), cls_only=True), dict(context_feat=cx)),
                         ("no_context", step_amd.TwoBranchNet(cfg(no_context=True)), dict())):
        net = fill(net, "det0.").to(dev).eval()
        net.set_device(dev)
        with torch.no_grad():
            o = net(pf, **kw)
        for nme, t in zip(("prob", "loc", "first", "last"), o[:4]):
            ref = g["%s_%s" % (tag, nme)]
            assert tuple(t.shape) == ref.shape, (tag, nme, tuple(t.shape), ref.shape)
            if np.abs(ref).max() == 0:
                assert float(t.abs().max()) == 0.0, (tag, nme)
            else:
                e = rel(np_(t), ref)
                assert e < 1e-3, (tag, nme, e)


def case_resample_bottleneck_concat_in_one_launch(dev, golden):
    """The heads' resample Bottleneck (models/two_branch.py:86-111: conv1 / conv2 over torch.cat((global, downsampled), 1)) without
    gradients in 16-bit storage: the convs over the concat run as ONE launch each (ConvUnit.cat -> step_conv_forward_cat,
    conv_pw2_kernel) where the planner streams the layer, with one fp32 accumulation over the whole K as the reference has -- no
    further from the fp32 torch restatement than the two accumulating launches (backbone.CAT_FUSE = False), whose partial sum is
    rounded to the storage type in between; with gradients wanted the two-launch units run (their autograd nodes)."""
    import torch.nn.functional as F
    from step_amd import backbone as _bb
    from step_amd import ops as _ops
    big = dev != "cpu"
    ia, ib, outp, planes, maps = (832, 256, 1024, 256, 100) if big else (96, 32, 128, 64, 84)
    blk = heads._BottleneckResample(ia, ib, outp, planes)
    g = torch.Generator().manual_seed(41)
    with torch.no_grad():
        for prm in blk.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) / (prm[0].numel() ** 0.5))
    blk = blk.to(dev)
    a = torch.relu(torch.randn(maps, 1, 7, 7, ia, generator=g)).to(dev)
    b = torch.relu(torch.randn(maps, 1, 7, 7, ib, generator=g)).to(dev)

    def ref():                                                    # fp32, NCHW, the reference's four convs
        x = torch.cat([a, b], -1).reshape(maps, 7, 7, ia + ib).permute(0, 3, 1, 2).float()
        w = {k: v.float() for k, v in blk.state_dict().items()}
        res = F.conv2d(x, w["conv1.weight"])
        o = F.relu(F.conv2d(x, w["conv2.weight"]))
        o = F.relu(F.conv2d(o, w["conv3.weight"], padding=1))
        return F.relu(F.conv2d(o, w["conv4.weight"]) + res).permute(0, 2, 3, 1).reshape(maps, 1, 7, 7, outp)
    want = np_(ref())
    a16, b16 = a.to(torch.bfloat16), b.to(torch.bfloat16)
    errs, names = {}, {}
    with torch.no_grad():
        for fuse in (True, False):
            try:
                _bb.CAT_FUSE = fuse
                _ops.PROFILE, _ops.PROFILE_LIMIT = [], 1 << 30
                y = blk(a16, b16)
                names[fuse] = [r[0] for r in _ops.PROFILE]
            finally:
                _bb.CAT_FUSE = True
                _ops.PROFILE, _ops.PROFILE_LIMIT = None, None
            errs[fuse] = rel(np_(y), want)
    assert sum("conv_pw2_kernel" in n for n in names[True]) >= 1, names[True]                  # conv1 at least (conv2: where the planner streams it)
    assert sum("conv_pw2_kernel" in n for n in names[False]) == 0, names[False]
    assert len(names[True]) < len(names[False]), (names[True], names[False])
    assert errs[True] < 1.5e-2 and errs[True] <= errs[False] * 1.25 + 1e-4, errs
    record("resample_bottleneck_bf16_rel_err_one_launch_vs_two", [errs[True], errs[False]])
    # gradients wanted: the two-launch units (no conv_pw2_kernel), same values as the no-grad two-launch run
    for prm in blk.parameters():
        prm.requires_grad_(True)
    try:
        _ops.PROFILE, _ops.PROFILE_LIMIT = [], 1 << 30
        yg = blk(a16, b16)
        ng = [r[0] for r in _ops.PROFILE]
    finally:
        _ops.PROFILE, _ops.PROFILE_LIMIT = None, None
    assert sum("conv_pw2_kernel" in n for n in ng) == 0, ng
    assert yg.requires_grad and rel(np_(yg), want) < 1.5e-2


def case_twobranch_T9_golden(dev, golden):
    _twobranch(dev, golden, 9)


def case_roinet_layouts(dev, golden):
    """ROINet on a channels-last backbone view and on a torch-contiguous tensor give the golden result."""
    g = golden("roi_nms_golden")
    conv = R.fill_tensor("golden.roi.conv", (2, 3, 16, 25, 25), "feat").to(dev)
    tubes = torch.from_numpy(g["tube_rois"]).to(dev)
    net = step_amd.ROINet("align", 7)
    out_nchw = net(conv, tubes)
    assert out_nchw.is_contiguous() and np.array_equal(np_(out_nchw), g["tube_out"])
    cl = conv.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)     # what BaseNet returns
    out_cl = net(cl, tubes)
    assert out_cl.permute(0, 2, 3, 1).is_contiguous() and np.array_equal(np_(out_cl), g["tube_out"])
    # the reference's call pattern: slice then .contiguous()  (utils/utils.py:48)
    out3 = net(cl[:, 0:3].contiguous(), tubes)
    assert np.array_equal(np_(out3), g["tube_out"])
    # a T-slice of a longer channels-last feature (what the step driver passes): zero-copy path
    long_cl = torch.cat([cl.new_zeros(2, 2, 16, 25, 25).permute(0, 1, 3, 4, 2), cl.permute(0, 1, 3, 4, 2),
                         cl.new_ones(2, 1, 16, 25, 25).permute(0, 1, 3, 4, 2)], 1).contiguous().permute(0, 1, 4, 2, 3)
    out4 = net(long_cl[:, 2:5], tubes)
    assert out4.permute(0, 2, 3, 1).is_contiguous() and np.array_equal(np_(out4), g["tube_out"])
    pool = step_amd.ROINet("pool", 7)
    a, b = pool(conv, tubes), pool(cl, tubes)
    assert np.array_equal(np_(a), np_(b))


def case_inference_golden(dev, golden, ntubes=11):
    """3-step inference on a synthetic backbone feature vs the reference's own utils.inference()."""
    g = golden("inference_golden")
    args = cfg()
    conv_feat = R.fill_tensor("golden.inf.feat", (2, 9, 832, 25, 25), "feat").to(dev)
    conv_cl = conv_feat.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    ctxnet = fill(step_amd.ContextNet(args)).to(dev).eval()
    nets = {"roi_net": step_amd.ROINet("align", 7)}
    for i in range(3):
        d = fill(step_amd.TwoBranchNet(args), "det%d." % i).to(dev).eval()
        d.set_device(dev)
        nets["det_net%d" % i] = d
    a = R.anchors()[:ntubes] * 400.0
    tl = [np.tile(a[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
    tl[1] = tl[1][::-1].copy()
    with torch.no_grad():
        context = ctxnet(conv_cl)
        assert rel(np_(context), g["n%d_context" % ntubes]) < 1e-4
        hist, _ = inference(args, conv_cl, context, nets, 3, tl)
    for i, h in enumerate(hist):
        assert list(h["tubes_nums"]) == list(g["n%d_step%d_nums" % (ntubes, i)])
        e = rel(np_(h["pred_prob"][:, 0]), g["n%d_step%d_pred_prob" % (ntubes, i)])
        assert e < 1e-3, (i, "prob", e)
        for k in ("pred_loc", "pred_first_loc", "pred_last_loc"):
            e = rel(np_(h[k]), g["n%d_step%d_%s" % (ntubes, i, k)])
            assert e < 1e-3, (i, k, e)


def case_inference_modes_golden(dev, golden):
    """temporal_mode "extrapolate" and "mean" (utils/utils.py:108-120) against the reference's own inference(): the tubes fed to
    step 3 are extended by linear extrapolation / by the tube's mean box instead of the head's neighbour regressions."""
    g = golden("inference_modes_golden")
    conv_feat = R.fill_tensor("golden.inf.feat", (2, 9, 832, 25, 25), "feat").to(dev)
    conv_cl = conv_feat.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    ctxnet = fill(step_amd.ContextNet(cfg())).to(dev).eval()
    nets = {"roi_net": step_amd.ROINet("align", 7)}
    for i in range(3):
        d = fill(step_amd.TwoBranchNet(cfg()), "det%d." % i).to(dev).eval()
        d.set_device(dev)
        nets["det_net%d" % i] = d
    a = R.anchors()[:11] * 400.0
    tl = [np.tile(a[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
    tl[1] = tl[1][::-1].copy()
    for mode in ("extrapolate", "mean"):
        args = cfg(temporal_mode=mode)
        with torch.no_grad():
            hist, traj = inference(args, conv_cl, ctxnet(conv_cl), nets, 3, tl)
        # the reference's trajectory form: per step, per clip (numpy proposals, class index [n,Tl])
        assert len(traj) == 3 and len(traj[1]) == 2 and isinstance(traj[1][0][0], np.ndarray)
        assert tuple(traj[1][0][1].shape) == (11, 3) and traj[1][0][1].dtype == torch.int64
        assert hist[1]["pred_first_loc"] is None and hist[1]["pred_last_loc"] is None        # utils.py:83-84
        props = np.concatenate([c[0] for c in traj[1]], axis=0)
        assert props.shape == (22, 9, 4)
        e = rel(props, g["%s_step1_proposals" % mode])
        assert e < 1e-3, (mode, "proposals", e)
        for i, h in enumerate(hist):
            for k, ref in (("pred_prob", g["%s_step%d_pred_prob" % (mode, i)]), ("pred_loc", g["%s_step%d_pred_loc" % (mode, i)])):
                got = np_(h[k][:, 0]) if k == "pred_prob" else np_(h[k])
                e = rel(got, ref)
                assert e < 1e-3, (mode, i, k, e)


def case_postprocess_golden(dev, golden):
    """driver.postprocess (batched per-(clip, class) device NMS, every refinement iteration, tensor ops only) against the rows
    the reference's own evaluation loop wrote for a seeded history with ragged clips (test.py:157-210, recorded by
    oracle/make_golden.py): same rows in the same order, boxes / scores bit-identical, and the same CSV text; all three
    evaluate_topk / topk settings including the reference's `[:-1]` slice."""
    from step_amd.driver import detections_csv, postprocess
    g = golden("postprocess_golden")
    nums = [int(v) for v in g["nums"]]
    hist = []
    for i in range(3):
        loc = torch.from_numpy(g["hist%d_loc" % i]).to(dev)
        prob = torch.from_numpy(g["hist%d_prob" % i]).to(dev)
        hist.append({"pred_prob": prob.unsqueeze(1).expand(-1, loc.shape[1], -1), "pred_loc": loc, "tubes_nums": nums})
    for tag, kw in (("all", dict(evaluate_topk=-1, topk=-1)), ("top20", dict(evaluate_topk=1, topk=20)), ("topm1", dict(evaluate_topk=5, topk=-1))):
        args = cfg(conf_thresh=0.01, nms_thresh=0.4, **kw)
        dets = postprocess(args, hist)
        assert len(dets) == 3 and all(len(d) == len(nums) for d in dets)
        meta, box, score, lines = [], [], [], []
        for it, clips in enumerate(dets):
            for b, d in enumerate(clips):
                m = int(d["scores"].shape[0])
                meta += [[it, b, int(c)] for c in d["labels"].cpu()]
                box.append(np_(d["boxes"]).reshape(m, 4))
                score.append(np_(d["scores"]))
                assert bool((d["tubes"] < nums[b]).all())
            lines += detections_csv(clips, [{"video_name": "vid%d" % b, "fid": 900 + b} for b in range(len(nums))])
        assert np.array_equal(np.asarray(meta, np.int32), g[tag + "_meta"]), tag
        assert np.array_equal(np.concatenate(box), g[tag + "_box"]), tag
        assert np.array_equal(np.concatenate(score), g[tag + "_score"]), tag
        assert lines == [str(x) for x in g[tag + "_lines"]], tag
    # the general path (more than 64 tubes per clip: mask / compaction as tensor operations + step_nms_batched) gives the same rows
    from step_amd import driver
    fastp = postprocess(cfg(conf_thresh=0.01, nms_thresh=0.4, evaluate_topk=-1, topk=-1), hist)
    for it in range(3):
        gen = driver._postprocess_general(hist[it], nums, 0.01, 0.4, -1, -1, 400.0, 400.0)
        for a_, b_ in zip(gen, fastp[it]):
            assert all(torch.equal(a_[k_], b_[k_]) for k_ in ("boxes", "scores", "labels", "tubes"))
    # NaN predictions: np.maximum / torch.clamp PROPAGATE a NaN and valid_tubes' `<` test then fails, so the reference turns a box with
    # any NaN coordinate into the whole frame -- the fused launch (fmaxf / fminf drop NaNs) must do the same as the tensor path
    hn = {"pred_prob": hist[0]["pred_prob"], "pred_loc": hist[0]["pred_loc"].clone(), "tubes_nums": nums}
    mid = hn["pred_loc"].shape[1] // 2
    for row, col in ((1, 0), (4, 3), (7, 1), (nums[0] + 2, 2)):
        hn["pred_loc"][row, mid, col] = float("nan")
    gen = driver._postprocess_general(hn, nums, 0.01, 0.4, -1, -1, 400.0, 400.0)
    fast = postprocess(cfg(conf_thresh=0.01, nms_thresh=0.4, evaluate_topk=-1, topk=-1), [hn])[0]
    for a_, b_ in zip(gen, fast):
        assert all(torch.equal(a_[k_], b_[k_]) for k_ in ("boxes", "scores", "labels", "tubes"))
        assert not bool(torch.isnan(b_["boxes"]).any())
    # one iteration only, and an empty clip in the batch
    only = postprocess(cfg(), hist, iterations=(2,))
    assert len(only) == 1
    h0 = {"pred_prob": hist[0]["pred_prob"][:18], "pred_loc": hist[0]["pred_loc"][:18], "tubes_nums": [11, 0, 7]}
    d0 = postprocess(cfg(), [h0])[0]
    assert d0[1]["scores"].numel() == 0
    full = postprocess(cfg(), hist, iterations=(0,))[0]
    assert torch.equal(d0[0]["boxes"], full[0]["boxes"]) and torch.equal(d0[2]["scores"], full[1]["scores"])
    # the one-launch row compaction (step_detect_compact, round 6) against the tensor-op form it replaced: the same rows, bit for bit
    assert driver.COMPACT_KERNEL
    try:
        driver.COMPACT_KERNEL = False
        old = postprocess(cfg(conf_thresh=0.01, nms_thresh=0.4, evaluate_topk=-1, topk=-1), hist)
        old0 = postprocess(cfg(), [h0])[0]
    finally:
        driver.COMPACT_KERNEL = True
    for it in range(3):
        for a_, b_ in zip(old[it], fastp[it]):
            assert all(torch.equal(a_[k_], b_[k_]) and a_[k_].dtype == b_[k_].dtype for k_ in ("boxes", "scores", "labels", "tubes"))
    for a_, b_ in zip(old0, d0):
        assert all(torch.equal(a_[k_], b_[k_]) for k_ in ("boxes", "scores", "labels", "tubes"))


def case_inference_golden_34(dev, golden):
    case_inference_golden(dev, golden, 34)


def case_e2e_c3_golden(dev, golden):
    """C3 end to end: AVA-shaped clip [1,36,3,400,400] -> BaseNet -> ContextNet -> 3-step inference."""
    g = golden("e2e_c3_golden")
    args = cfg()
    base = fill(step_amd.BaseNet(args)).to(dev).eval()
    ctxnet = fill(step_amd.ContextNet(args)).to(dev).eval()
    nets = {"roi_net": step_amd.ROINet("align", 7)}
    for i in range(3):
        d = fill(step_amd.TwoBranchNet(args), "det%d." % i).to(dev).eval()
        d.set_device(dev)
        nets["det_net%d" % i] = d
    x = R.fill_tensor("golden.c3.images", (1, 36, 3, 400, 400), "image").to(dev)
    a = R.anchors()[:11] * 400.0
    with torch.no_grad():
        conv_feat = base(x)
        assert tuple(conv_feat.shape) == (1, 9, 832, 25, 25)
        f = np_(conv_feat).reshape(-1)
        step = int(g["conv_feat_stats"][3])
        assert rel(f[::step][:256], g["conv_feat_sample"]) < 1e-3
        assert rel(np_(conv_feat[0, :, ::13, ::4, ::4]), g["conv_feat_slice"]) < 1e-3
        context = ctxnet(conv_feat)
        assert rel(np_(context), g["context"]) < 1e-3
        hist, _ = inference(args, conv_feat, context, nets, 3, [np.tile(a[:, None, :], (1, 3, 1)).astype(np.float32)])
    for i, h in enumerate(hist):
        assert rel(np_(h["pred_prob"][:, 0]), g["step%d_pred_prob" % i]) < 1e-3, i
        for k in ("pred_loc", "pred_first_loc", "pred_last_loc"):
            e = rel(np_(h[k]), g["step%d_%s" % (i, k)])
            assert e < 1e-3, (i, k, e)


def case_training_step_matches_torch_autograd(dev, golden):
    """One TwoBranchNet training step (losses + backward) on the MARGIN weights (oracle.i3d_ref.fill_state_dict_margin: every ReLU
    pre-activation of the head is structurally >= 0.35 rms away from zero, so no mask can flip under a different summation order):
    loss and every trainable parameter's gradient of the HIP module against (a) the torch-CPU restatement under autograd on the same
    weights / inputs, all elements, and (b) the REFERENCE's own TwoBranchNet under its own autograd (head_grad_margin_golden.npz,
    `python -m oracle.make_golden head_grad_margin`: per-parameter L2 norm and a strided 512-element sample) -- 1e-3."""
    g = golden("head_golden")
    rg = golden("head_grad_margin_golden")
    net = step_amd.TwoBranchNet(cfg())
    net.load_state_dict(R.fill_state_dict_margin({k: tuple(v.shape) for k, v in net.state_dict().items()}, "det0."))
    net = net.to(dev)
    net.set_device(dev)
    net.train()
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)
    o = net(pf, context_feat=cx, tubes=tubes, targets=targets)
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    assert rel(np.concatenate([np_(t).reshape(-1) for t in o[:4]]), rg["outputs"]) < 1e-3
    assert abs(float(np_(loss)) - float(rg["loss"])) < 1e-3 * abs(float(rg["loss"]))
    # (a) the restatement with autograd, every element
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "batch3d" not in k)
          for k, v in net.state_dict().items()}
    oo = R.twobranch_forward(pf.cpu(), cx.cpu(), sd, tubes=tubes.cpu(), targets=targets.cpu())
    (oo[4].mean() + 5.0 * oo[5].mean() + oo[6].mean()).backward()
    checked, worst = 0, 0.0
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, k
        a, b = np_(p.grad).astype(np.float64), sd[k].grad.numpy().astype(np.float64)
        e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        worst = max(worst, e)
        assert e < 1e-3, (k, e)
        checked += 1
    record("head_grad_margin_worst_rel_l2", worst)
    info = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
    assert checked == len(info["TwoBranchNet_trainable"])
    # (b) the reference's own autograd
    params = dict(net.named_parameters())
    for k in (str(n) for n in rg["names"]):
        gr = params[k].grad.detach().reshape(-1)
        a = np_(gr[::int(rg["step." + k])][:512]).astype(np.float64)
        b = rg["sample." + k].astype(np.float64)
        n_ = float(gr.double().norm())
        assert abs(n_ - float(rg["norm." + k])) <= 1e-3 * float(rg["norm." + k]), (k, n_, float(rg["norm." + k]))
        assert np.linalg.norm(a - b) <= 1e-3 * max(np.linalg.norm(b), 1e-30) + 1e-12, k


def case_training_step_generic_weights(dev, golden):
    """The same step on the GENERIC filler weights against the reference's own autograd (head_grad_golden.npz).  With ~300 k ReLU
    pre-activations of ordinary magnitude one always lies within fp32 summation noise of zero; ONE flipped mask element moves the
    upstream gradients of this two-tube batch by 0.4-1.5 % in relative L2 (observed), with no flip the agreement is ~1e-6.  This case
    therefore only bounds the damage (5e-2); the 1e-3 parity statement is case_training_step_matches_torch_autograd above."""
    g = golden("head_golden")
    net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
    net.set_device(dev)
    net.train()
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)
    o = net(pf, context_feat=cx, tubes=tubes, targets=targets)
    loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    loss.backward()
    rg = golden("head_grad_golden")
    assert abs(float(np_(loss)) - float(rg["loss"])) < 1e-3 * abs(float(rg["loss"]))
    params = dict(net.named_parameters())
    for k in (str(n) for n in rg["names"]):
        gr = params[k].grad.detach().reshape(-1)
        a = np_(gr[::int(rg["step." + k])][:512]).astype(np.float64)
        b = rg["sample." + k].astype(np.float64)
        n_ = float(gr.double().norm())
        assert abs(n_ - float(rg["norm." + k])) <= 5e-2 * float(rg["norm." + k]), (k, n_, float(rg["norm." + k]))
        assert np.linalg.norm(a - b) <= 5e-2 * max(np.linalg.norm(b), 1e-30) + 1e-12, k


def _grad_check(named_params, sd, tol, what):
    errs = []
    for k, p in named_params:
        if not p.requires_grad:
            continue
        assert p.grad is not None, (what, k)
        a, b = np_(p.grad).astype(np.float64), sd[k].grad.numpy().astype(np.float64)
        errs.append((float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)), k))
    bad = [(k, "%.2e" % e) for e, k in errs if not e < tol]
    assert not bad, (what, bad, "all: " + " ".join("%s=%.1e" % (k.replace("base_model.", "").replace(".conv3d.weight", ""), e) for e, k in errs))
    return len(errs)


def _oracle_sd(net):
    return {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "batch3d" not in k)
            for k, v in net.state_dict().items()}


def case_basenet_backward_matches_oracle_autograd(dev, golden):
    """a-19, the backbone leg of train.py:257-348: a scalar back-propagated through the WHOLE BaseNet in training mode --
    _StemFn (stem weight gradient), every _MaxPoolFn, Mixed's autograd branch (torch.cat of the four branches), the 3x3x3
    and 1x1x1 data-gradient convs and the weight-gradient kernels -- and all 45 trainable gradients compared with torch
    autograd through oracle/i3d_ref.basenet_forward on the same weights, fp32, relative L2 < 1e-3.  The loss sums every
    output element with a signed weight, so a ReLU whose pre-activation sits within rounding of zero moves the result by
    ~1e-6, not by per cents as in the tiny head fixture.  C1 clip on the GPU, a 32x32 clip on the interpreter."""
    shape = (1, 8, 3, 112, 112) if dev != "cpu" else (1, 4, 3, 32, 32)
    net = fill(step_amd.BaseNet(cfg())).to(dev)
    net.train()
    x = (torch.rand(*shape, generator=torch.Generator().manual_seed(10)) * 2 - 1).to(dev)
    y = net(x)
    wgt = R.fill_tensor("golden.bwd.base.w", tuple(y.shape), "image")
    (y * wgt.to(dev)).sum().backward()
    sd = _oracle_sd(net)
    yo = R.basenet_forward(x.cpu(), sd)
    assert rel(np_(y), yo.detach().numpy()) < 1e-4
    (yo * wgt).sum().backward()
    n = _grad_check(net.named_parameters(), sd, 1e-3, "BaseNet")
    assert n == 45
    # ... and the gradients the REFERENCE's own BaseNet produced under its own autograd for the same clip and weighting
    # (base_grad_golden.npz, `python -m oracle.make_golden base_grad`: L2 norm + strided 512-element sample per tensor)
    rg, tag = golden("base_grad_golden"), ("gpu" if dev != "cpu" else "emul")
    assert abs(float(y.detach().double().norm()) - float(rg[tag + ".out_l2"])) < 1e-3 * float(rg[tag + ".out_l2"]), (float(y.detach().double().norm()), float(rg[tag + ".out_l2"]))
    params = dict(net.named_parameters())
    names = [str(k_) for k_ in rg[tag + ".names"]]
    assert len(names) == 45
    for k in names:
        gr = params[k].grad.detach().reshape(-1)
        a = np_(gr[::int(rg["%s.step.%s" % (tag, k)])][:512]).astype(np.float64)
        b = rg["%s.sample.%s" % (tag, k)].astype(np.float64)
        nr = float(rg["%s.norm.%s" % (tag, k)])
        # (a 512-element subsample is noisier than the all-element L2 above, and the fixture itself is ~2e-4 from the restatement:
        # 5e-3 on the GPU's C1-sized clip; a wrong kernel is off by O(1))
        tol_ = 1e-3 if dev == "cpu" else 5e-3
        assert abs(float(gr.double().norm()) - nr) <= tol_ * nr, (k, float(gr.double().norm()), nr)
        assert np.linalg.norm(a - b) <= tol_ * max(np.linalg.norm(b), 1e-30), (k, float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)))


def case_basenet_batch_statistics_bn_golden(dev, golden):
    """--freeze_stats False (models/networks.py:85-99: BatchNorm stays in TRAINING mode; models/i3dpt.py:95-110) with trainable BN affine,
    against what the REFERENCE's own BaseNet produced under its own autograd (bn_train_golden.npz, `python -m oracle.make_golden
    bn_train`) on two-clip batches.  Batch statistics run on csrc/bn.hip; nothing on this path is a torch batch_norm kernel.
      * whole net, forward: output, every BN layer's running statistics after the step (momentum 0.1, unbiased variance) and
        num_batches_tracked at 1e-3 (48x48 clips; on the GPU also the C1-sized 112x112 clips);
      * backward, piece by piece at 1e-4 against torch autograd through the restatement (itself pinned to the reference's 135 gradients
        at 2e-4, tests/test_oracle_golden.py): the stem unit (raw stem conv -> BN -> ReLU) and one Inception block (1x1x1 and 3x3x3 units,
        the pooled branch, gamma / beta / conv weights / the gradient handed upstream);
      * whole net, backward: all 135 gradients against the reference's at 5e-2.  A batch-normalised pre-activation is centred on zero, so
        among ~1e5 of them a handful sit within fp32 rounding of the ReLU threshold and fall on the other side under a different (equally
        valid) summation order; each flipped mask element moves a gradient sum by one whole term (measured: the restatement in fp32
        against itself in fp64 differs by up to 6e-3 on C1-sized clips, the HIP path by 1-2.5e-2 from the fp32 restatement, with the
        differing output elements at |y| < 3e-5).  The piecewise checks above are what pin the arithmetic."""
    g = golden("bn_train_golden")
    cases = [("emul", (2, 4, 3, 48, 48), True)] if dev == "cpu" else [("gpu", (2, 8, 3, 48, 48), True), ("c1", (2, 8, 3, 112, 112), False)]
    for tag, shape, with_grads in cases:
        net = fill(step_amd.BaseNet(cfg(freeze_stats=False, freeze_affine=False))).to(dev)
        net.train()
        assert all(m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm3d))
        x = (torch.rand(*shape, generator=torch.Generator().manual_seed(11)) * 2 - 1).to(dev)
        y = net(x)
        f = y.detach().reshape(-1)
        assert abs(float(f.double().norm()) - float(g[tag + ".out_l2"])) < 1e-3 * float(g[tag + ".out_l2"])
        assert rel(np_(f[::int(g[tag + ".out_step"])][:4096]), g[tag + ".out_sample"]) < 1e-3
        sd = net.state_dict()
        run = np.concatenate([np_(sd[str(k)]).reshape(-1) for k in g[tag + ".running_keys"]])
        assert rel(run, g[tag + ".running"]) < 1e-3
        assert all(int(sd[k]) == 1 for k in sd if k.endswith("num_batches_tracked"))
        if not with_grads:
            continue
        wgt = R.fill_tensor("golden.bn_train.w", tuple(y.shape), "image")
        (y * wgt.to(dev)).sum().backward()
        params = dict(net.named_parameters())
        names = [str(k_) for k_ in g[tag + ".names"]]
        assert len(names) == 135
        worst = 0.0
        for k in names:
            gr = params[k].grad.detach().reshape(-1)
            a = np_(gr[::int(g["%s.step.%s" % (tag, k)])][:512]).astype(np.float64)
            b = g["%s.sample.%s" % (tag, k)].astype(np.float64)
            nr = float(g["%s.norm.%s" % (tag, k)])
            e1 = abs(float(gr.double().norm()) - nr) / max(nr, 1e-30)
            e2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
            worst = max(worst, e1, e2)
            assert e1 <= 5e-2 and e2 <= 5e-2, (k, e1, e2)
        record("bn_train_worst_grad_rel_" + tag, worst)
        # eval mode afterwards folds the UPDATED running statistics into the conv epilogues again
        net.eval()
        with torch.no_grad():
            ye = net(x)
            yo = R.basenet_forward(x.cpu(), {k: v.detach().cpu() for k, v in net.state_dict().items()})
        assert rel(np_(ye), yo.numpy()) < 1e-4
    # ---- backward piece by piece against autograd through the restatement
    gen = torch.Generator().manual_seed(3)
    # (a) the stem unit: raw stem conv -> batch-statistics BN -> ReLU
    u = step_amd.backbone.Unit3D(3, 64, (7, 7, 7), (2, 2, 2))
    shapes = {k: tuple(v.shape) for k, v in u.state_dict().items()}
    u.load_state_dict({k: v for k, v in zip(shapes, (R.fill_state_dict({"base_model.0." + k: s_ for k, s_ in shapes.items()})["base_model.0." + k] for k in shapes))})
    u = u.to(dev)
    u.train()
    x = (torch.rand(2, 6, 3, 24, 24, generator=gen) * 2 - 1)
    sdo = {"p." + k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in u.state_dict().items()}
    y = u(x.to(dev))
    w_ = torch.randn(tuple(y.shape), generator=gen)
    (y * w_.to(dev)).sum().backward()
    yo = R.unit3d(x.permute(0, 2, 1, 3, 4), sdo, "p", stride=(2, 2, 2), train_bn=True)
    (yo * w_.permute(0, 4, 1, 2, 3)).sum().backward()
    assert rel(np_(y.permute(0, 4, 1, 2, 3)), yo.detach().numpy()) < 1e-4
    for k, p in u.named_parameters():
        a, b = np_(p.grad).astype(np.float64), sdo["p." + k].grad.numpy().astype(np.float64)
        assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), ("stem", k, float(np.linalg.norm(a - b) / np.linalg.norm(b)))
    # (b) one Inception block (mixed_4f's channel plan) on a 2 x 3 x 3 map, two clips
    m = step_amd.backbone.Mixed(*step_amd.backbone.MIXED_CFG["4f"])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    filled = R.fill_state_dict({"base_model.12." + k: s_ for k, s_ in shapes.items()})
    m.load_state_dict({k: filled["base_model.12." + k] for k in shapes})
    m = m.to(dev)
    m.train()
    sdo = {"p." + k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in m.state_dict().items()}
    x0 = torch.relu(torch.randn(2, 2, 3, 3, 528, generator=gen))
    xa = x0.clone().to(dev).requires_grad_(True)
    y = m(xa)
    w_ = torch.randn(tuple(y.shape), generator=gen)
    (y * w_.to(dev)).sum().backward()
    xo = x0.permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    yo = R.mixed(xo, sdo, "p", train_bn=True)
    (yo * w_.permute(0, 4, 1, 2, 3)).sum().backward()
    assert rel(np_(y.permute(0, 4, 1, 2, 3)), yo.detach().numpy()) < 1e-4
    ga, gb = np_(xa.grad.permute(0, 4, 1, 2, 3)).astype(np.float64), xo.grad.numpy().astype(np.float64)
    assert np.linalg.norm(ga - gb) <= 1e-4 * np.linalg.norm(gb)
    n = 0
    for k, p in m.named_parameters():
        a, b = np_(p.grad).astype(np.float64), sdo["p." + k].grad.numpy().astype(np.float64)
        assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b), ("mixed", k, float(np.linalg.norm(a - b) / np.linalg.norm(b)))
        n += 1
    assert n == 18
    sdn = m.state_dict()
    for k in shapes:
        if "running" in k:
            assert rel(np_(sdn[k]), sdo["p." + k].detach().numpy()) < 1e-5, k


def case_contextnet_backward_matches_oracle_autograd(dev, golden):
    """The ContextNet leg: MaxPoolTF((1,3,3),(1,2,2)) -> mixed_5b -> mixed_5c -> 13x13 average with gradients, parameter
    gradients AND the gradient handed back to the backbone feature, against autograd through the oracle."""
    T = 2 if dev != "cpu" else 1
    net = fill(step_amd.ContextNet(cfg())).to(dev)
    net.train()
    # (a generator-drawn feature, not the sin-hash filler: that one is quasi-periodic, two pool windows of one channel can
    # hold values one ulp apart and the max-pool gradient then goes to whichever the summation order favours)
    cf = torch.relu(torch.randn(1, T, 832, 25, 25, generator=torch.Generator().manual_seed(11))) * 1.5
    a = cf.clone().to(dev).requires_grad_(True)
    y = net(a)
    wgt = R.fill_tensor("golden.bwd.ctx.w", tuple(y.shape), "image")
    (y * wgt.to(dev)).sum().backward()
    sd = _oracle_sd(net)
    b = cf.clone().requires_grad_(True)
    yo = R.contextnet_forward(b, sd)
    assert rel(np_(y), yo.detach().numpy()) < 1e-4
    (yo * wgt).sum().backward()
    n = _grad_check(net.named_parameters(), sd, 1e-3, "ContextNet")
    assert n == 12
    ga, gb = np_(a.grad).astype(np.float64), b.grad.numpy().astype(np.float64)
    e = float(np.linalg.norm(ga - gb) / np.linalg.norm(gb))
    assert e < 1e-3, e


def case_base_context_chain_backward(dev, golden):
    """BaseNet -> ContextNet chained on an [1,8,3,400,400] clip (feature [1,2,832,25,25]): the loss reads both the context
    vector and the backbone feature, as the training step does (heads read conv_feat through ROIAlign, train.py:296-331), so
    the backbone's gradients are the SUM of two paths.  All 57 trainable gradients against the oracle's autograd."""
    base = fill(step_amd.BaseNet(cfg())).to(dev)
    ctxn = fill(step_amd.ContextNet(cfg())).to(dev)
    base.train()
    ctxn.train()
    x = (torch.rand(1, 8, 3, 400, 400, generator=torch.Generator().manual_seed(12)) * 2 - 1).to(dev)
    cf = base(x)
    cx = ctxn(cf)
    w1 = R.fill_tensor("golden.bwd.chain.w1", tuple(cx.shape), "image")
    w2 = R.fill_tensor("golden.bwd.chain.w2", tuple(cf.shape), "image") * 0.01
    ((cx * w1.to(dev)).sum() + (cf * w2.to(dev)).sum()).backward()
    sdb, sdc = _oracle_sd(base), _oracle_sd(ctxn)
    cfo = R.basenet_forward(x.cpu(), sdb)
    cxo = R.contextnet_forward(cfo, sdc)
    ((cxo * w1).sum() + (cfo * w2).sum()).backward()
    # A 400x400 clip with a uniform random filler puts a few pre-activations within ~1e-7 of zero; one that lands on the other
    # side of its ReLU under a different summation order moves every gradient UPSTREAM of it by ~1 % in relative L2 (DESIGN.md
    # section 4; measured here: 7.8e-3 on the stem weight, the most upstream tensor).  The bound leaves room for that; a wrong
    # kernel is off by O(1), and the C1-sized cases above hold 1e-3 on inputs chosen to stay clear of the ReLU boundary.
    assert _grad_check(base.named_parameters(), sdb, 3e-2, "chain.BaseNet") == 45
    assert _grad_check(ctxn.named_parameters(), sdc, 3e-2, "chain.ContextNet") == 12


def case_flat_adam_matches_torch(dev, golden):
    """step_amd.optim.FlatAdam against torch.optim.Adam (train.py:126) on the parameter groups utils/solver.py builds
    (bias: 2x lr, no decay): three steps with an lr change in between (the schedulers rewrite group['lr']), a stray
    .grad replacement, the fused gradient clear, the autograd version bump and a state_dict round trip INTO torch's Adam."""
    torch.manual_seed(5)
    shapes = [(7, 3, 1, 3, 3), (7,), (5, 7), (5,), (130,)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]

    def groups(ps):
        return [{"params": [p], "lr": 2e-3 if p.dim() == 1 else 1e-3, "weight_decay": 0 if p.dim() == 1 else 1e-4} for p in ps]

    o_ref = torch.optim.Adam(groups(ref), lr=1e-3)
    o = step_amd.FlatAdam(groups(mine), lr=1e-3)
    assert len(o.param_groups) == len(o_ref.param_groups) and o.param_groups[1]["lr"] == 2e-3
    # the reference's schedulers are torch _LRScheduler subclasses (utils/solver.py:96,141): they must accept it
    sch = [torch.optim.lr_scheduler.LambdaLR(x, lambda k: 0.5 if k == 1 else 1.0) for x in (o_ref, o)]
    for p, q in zip(ref, mine):
        assert torch.equal(p.detach(), q.detach().cpu())         # re-homing into the arena keeps the values
    for it in range(3):
        assert [g_["lr"] for g_ in o.param_groups] == [g_["lr"] for g_ in o_ref.param_groups]
        assert o.param_groups[0]["lr"] == (5e-4 if it == 1 else 1e-3)
        o_ref.zero_grad()
        if it != 2:
            o.zero_grad()                                        # it == 2 relies on the clear fused into step 1
        vers = [q._version for q in mine]
        for i, (p, q) in enumerate(zip(ref, mine)):
            w = torch.randn(p.shape)
            (p * w).sum().mul(3.0).backward()
            if it == 0 and i == 2:
                q.grad = (w * 1.5).to(dev)                       # a caller that replaced .grad (not doubled below, hence 1.5):
                stray = q                                        # step() folds it back into the arena
            else:
                (q * w.to(dev)).sum().mul(3.0).backward()
        o_ref.step()
        o.flat_grad.mul_(0.5)                                    # grad_scale 2 on halved gradients == the same update
        o.step(grad_scale=2.0, zero_grad=(it == 1))
        for x in sch:
            x.step()
        assert stray.grad.data_ptr() == o.flat_grad.data_ptr() + 4 * 256     # 189 -> 192, 7 -> 64 elements before it
        for p, q, v in zip(ref, mine, vers):
            assert q._version > v
            assert rel(np_(q), p.detach().numpy()) < 2e-6, (it, tuple(p.shape))
        if it == 1:
            assert float(o.flat_grad.abs().max()) == 0.0
    sd = o.state_dict()
    o2 = torch.optim.Adam(groups([torch.nn.Parameter(q.detach().cpu().clone()) for q in mine]), lr=1e-3)
    o2.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in st.items()} for k, st in sd["state"].items()},
                        "param_groups": sd["param_groups"]})
    for k, p in enumerate(ref):
        assert rel(o2.state[o2.param_groups[k]["params"][0]]["exp_avg_sq"].numpy(), o_ref.state[p]["exp_avg_sq"].numpy()) < 1e-5
        assert float(o2.state[o2.param_groups[k]["params"][0]]["step"]) == 3.0
    o3 = step_amd.FlatAdam(groups([torch.nn.Parameter(q.detach().clone()) for q in mine]), lr=1e-3)
    o3.load_state_dict(o_ref.state_dict())
    assert o3.step_count == 3 and rel(np_(o3.exp_avg), np_(o.exp_avg)) < 1e-5


def case_wgrad_into_and_targets(dev, golden):
    """ops.conv_wgrad(into=...) adds to what the buffer holds (the kernel's accumulate mode) and refuses buffers of the wrong
    kind; backbone._wgrad_target only offers a parameter's .grad when the effective weight IS the parameter."""
    from step_amd import backbone, ops
    torch.manual_seed(2)
    x = torch.randn(1, 2, 5, 6, 8).to(dev)
    gy = torch.randn(1, 2, 5, 6, 16).to(dev)
    ref = ops.conv_wgrad(x, gy, 16, (1, 3, 3))
    buf = torch.full((16, 8, 1, 3, 3), 2.0, device=dev)
    out = ops.conv_wgrad(x, gy, 16, (1, 3, 3), into=buf)
    assert out is buf and rel(np_(buf) - 2.0, np_(ref)) < 1e-5
    for bad in (torch.zeros(16, 8, 1, 3, 3, dtype=torch.float64, device=dev), torch.zeros(16, 8, 1, 3, 2, device=dev),
                torch.zeros(16, 16, 1, 3, 3, device=dev)[:, ::2]):
        try:
            ops.conv_wgrad(x, gy, 16, (1, 3, 3), into=bad)
        except RuntimeError:
            continue
        raise AssertionError("conv_wgrad(into=...) accepted a %s %s buffer" % (bad.dtype, tuple(bad.shape)))
    conv = torch.nn.Conv3d(8, 16, (1, 3, 3), bias=False).to(dev)
    plain = backbone.ConvUnit(conv, lambda m: m.weight, (1, 3, 3))
    sliced = backbone.ConvUnit(conv, lambda m: m.weight, (1, 3, 3), cin_slice=(0, 4))
    assert backbone._wgrad_target(plain, plain.effective_weight()) is None          # no gradient buffer yet
    conv.weight.grad = torch.zeros_like(conv.weight)
    assert backbone._wgrad_target(plain, plain.effective_weight()) is conv.weight.grad
    assert backbone._wgrad_target(sliced, sliced.effective_weight()) is None        # a channel slice goes through autograd
    conv.weight.requires_grad_(False)
    assert backbone._wgrad_target(plain, plain.effective_weight()) is None


def case_wgrad_into_grad_matches_autograd(dev, golden):
    """backbone.wgrad_into_grad(): weight gradients accumulated straight into .grad on a side stream equal the ones autograd
    delivers (same kernels, same operands; only the fp32 atomics' order differs), they ADD to what .grad already holds, and
    parameters the shortcut does not cover (biases, permuted / sliced weights) still arrive through autograd."""
    from step_amd import backbone
    g = golden("head_golden")
    net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
    net.set_device(dev)
    net.train()
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)

    def loss():
        o = net(pf, context_feat=cx, tubes=tubes, targets=targets)
        return o[4].mean() + 5.0 * o[5].mean() + o[6].mean()

    loss().backward()
    ref = {k: p.grad.clone() for k, p in net.named_parameters() if p.requires_grad}
    for p in net.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)                         # persistent buffers holding something already
    with backbone.wgrad_into_grad():
        loss().backward()
    direct = 0
    for k, p in net.named_parameters():
        if not p.requires_grad:
            continue
        a, b = np_(p.grad).astype(np.float64) - 1.0, np_(ref[k]).astype(np.float64)
        e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        assert e < 1e-4, (k, e)
    assert not backbone._PENDING[0] and not backbone.WGRAD_INTO_GRAD


def case_train_select_device_front_end(dev, golden):
    """SURVEY 8 f-3 on the device: train_select with the previous step's predictions as DEVICE tensors -- class-score means,
    valid_tubes and the IoU table from one step_select_prepare launch and one device-to-host copy -- against what the reference's own
    train_select returned for the same inputs and the same `random` / `numpy.random` seeds (selection_golden.npz): same tubes, same
    order, same target rows, bit for bit; all six recorded cases (sampling modes, top-k, score ties, an invalid box, three temporal
    modes)."""
    import random

    from step_amd import selection as S
    from step_amd.tube_math import generate_anchors
    g = np.load(os.path.join(GOLDEN, "selection_golden.npz"))
    for ci in range(6):
        args = NS(T=3, NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3}, max_iter=3, num_classes=60, image_size=(400, 400),
                  cls_thresh=[0.2, 0.35, 0.5], reg_thresh=[0.2, 0.35, 0.5], max_pos_num=int(g["c%d_max_pos_num" % ci]),
                  neg_ratio=int(g["c%d_neg_ratio" % ci]), topk=int(g["c%d_topk" % ci]),
                  selection_sampling=str(g["c%d_sampling" % ci]), temporal_mode=str(g["c%d_mode" % ci]))
        seed = int(g["c%d_seed" % ci])
        targets = [g["c%d_targets%d" % (ci, b)] for b in range(2)]
        anchors = (generate_anchors() * 400.0).astype(np.float32)
        tubes = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(2)]
        hist = {k: torch.from_numpy(g["c%d_hist_%s" % (ci, k)]).to(dev) for k in ("pred_loc", "pred_first_loc", "pred_last_loc")}
        hist["pred_prob"] = torch.from_numpy(np.tile(g["c%d_hist_pred_prob" % ci], (1, 3, 1))).to(dev)
        hist["tubes_nums"] = [34, 20]
        for step in (2, 3):
            random.seed(seed * 10 + step)
            np.random.seed(seed * 10 + step)
            sel, tgt = S.train_select(step, hist, targets, tubes, args, device=True)
            for b in range(2):
                ref_sel, ref_tgt = g["c%d_s%d_sel%d" % (ci, step, b)], g["c%d_s%d_tgt%d" % (ci, step, b)]
                assert sel[b].shape == ref_sel.shape and sel[b].dtype == ref_sel.dtype, (ci, step, b)
                assert np.array_equal(sel[b], ref_sel), (ci, step, b)
                assert np.array_equal(tgt[b], ref_tgt), (ci, step, b)


def case_training_iteration_with_selection(dev, golden):
    """The whole iteration of train.py:257-348 on one AVA-shaped clip (step_amd.workloads.C4SelectTrainStep): no-grad
    inference, train_select between the steps, three heads, backward, fused Adam.  Checks what is size-independent: per step
    at most max_pos_num positives + neg_ratio x negatives per clip, a finite loss, every trainable tensor updated through the
    flat arena (autograd version bumped, packed-weight caches refreshed: the second step's loss differs), and determinism --
    the same seeds give the same selection and bit-identical losses."""
    import random
    from step_amd import workloads

    def run():
        random.seed(7)
        np.random.seed(7)
        w = workloads.C4SelectTrainStep(dev, batch=1, seed=11)
        for g_ in w.opt.param_groups:
            g_["lr"] = 1e-4
        p0 = w.opt.flat_param.clone()
        vers = [p._version for p in w.params]
        l1 = float(w.step())
        sel1 = [list(x) for x in w.selected]
        l2 = float(w.step())
        assert all(p._version > v for p, v in zip(w.params, vers))
        moved = float((w.opt.flat_param - p0).abs().max())
        return l1, l2, sel1, moved, w

    l1, l2, sel, moved, w = run()
    assert np.isfinite(l1) and np.isfinite(l2) and l1 != l2
    assert len(sel) == 3 and all(1 <= n <= 5 + 2 * 5 for s_ in sel for n in s_), sel
    assert 0 < moved < 1e-2                                      # Adam: at most ~lr per step
    assert float(w.opt.flat_grad.abs().max()) == 0.0             # cleared inside the optimizer pass
    m1, m2, sel2, _, _ = run()
    assert sel2 == sel and m1 == l1


def case_training_step_16bit_storage(dev, golden):
    """The same training step with bf16 activations (fp32 master weights, fp32 weight gradients): every gradient is
    finite and follows the fp32 run (16-bit activations and data gradients: a few per cent in relative L2).  Exercises
    the 16-bit data-gradient path of layers whose channel counts are not multiples of 8 (60 classes, 12 regressor
    columns), which must be padded to the kernels' 16-byte channel vectors."""
    g = golden("head_golden")
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)
    grads = {}
    for dt in (torch.float32, torch.bfloat16):
        net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
        net.set_device(dev)
        net.train()
        o = net(pf.to(dt), context_feat=cx.to(dt), tubes=tubes, targets=targets)
        (o[4].mean() + 5.0 * o[5].mean() + o[6].mean()).backward()
        grads[dt] = {k: np_(p.grad).astype(np.float64) for k, p in net.named_parameters() if p.requires_grad}
        assert all(np.isfinite(v).all() for v in grads[dt].values())
    worst = 0.0
    for k, a in grads[torch.bfloat16].items():
        b = grads[torch.float32][k]
        worst = max(worst, float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)))
    assert worst < 0.15, worst


def case_fp16_training_step_loss_scaling(dev, golden):
    """The reference's only 16-bit training mode (train.py:136-139 apex amp O1 with dynamic loss scaling, :342-345 amp.scale_loss): fp16
    activations, fp32 master weights, the loss multiplied by the scale on the device (step_amd.LossScaler), FlatAdam.step(scaler=...)
    = overflow scan + unscale + skip-or-step + scale update in one call.  (1) a step with scale 2^12 moves the parameters like the
    step without scaling (the scale is a power of two: the fp32 gradients agree up to the fp16 rounding of the activation gradients);
    (2) a scale that overflows the fp16 gradients is SKIPPED: parameters, moments and step count untouched, gradients cleared, scale
    halved -- and the next step at a sane scale goes through."""
    g = golden("head_golden")
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev).half()
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev).half()
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)

    def build():
        net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
        net.set_device(dev)
        net.train()
        opt = step_amd.FlatAdam([p for p in net.parameters() if p.requires_grad], lr=1e-4, capturable=True)
        return net, opt

    def loss_of(net):
        o = net(pf, context_feat=cx, tubes=tubes, targets=targets)
        return o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
    net0, opt0 = build()
    loss_of(net0).backward()
    g0 = opt0.flat_grad.clone()
    opt0.step(zero_grad=True)
    net1, opt1 = build()
    sc = step_amd.LossScaler(torch.device(dev), init_scale=2.0 ** 12, growth_interval=2)
    sc.scale_loss(loss_of(net1)).backward()
    g1 = opt1.flat_grad.clone()
    assert bool(torch.isfinite(g1).all())
    assert float((g1 / 4096.0 - g0).double().norm() / g0.double().norm()) < 2e-2
    p_before = opt1.flat_param.clone()
    opt1.step(scaler=sc, zero_grad=True)
    assert opt1.step_count == 1 and sc.scale == 4096.0 and float(sc.state[1]) == 1.0 and not bool(opt1.flat_grad.any())
    d0, d1 = (opt0.flat_param - p_before).double(), (opt1.flat_param - p_before).double()
    assert float((d1 - d0).norm() / d0.norm()) < 5e-2, float((d1 - d0).norm() / d0.norm())      # Adam's first step is sign-like: a few flipped tiny gradients
    # (2) overflow: a scale no fp16 gradient survives
    sc.state[0] = 2.0 ** 40
    p_before, m_before = opt1.flat_param.clone(), opt1.exp_avg.clone()
    sc.scale_loss(loss_of(net1)).backward()
    assert not bool(torch.isfinite(opt1.flat_grad).all())
    opt1.step(scaler=sc, zero_grad=True)
    assert opt1.step_count == 1 and torch.equal(opt1.flat_param, p_before) and torch.equal(opt1.exp_avg, m_before)
    assert sc.scale == 2.0 ** 39 and float(sc.state[1]) == 0.0 and not bool(opt1.flat_grad.any())
    sc.state[0] = 2.0 ** 10
    sc.scale_loss(loss_of(net1)).backward()
    opt1.step(scaler=sc, zero_grad=True)
    assert opt1.step_count == 2 and not torch.equal(opt1.flat_param, p_before) and bool(torch.isfinite(opt1.flat_param).all())
    sc.scale_loss(loss_of(net1)).backward()
    opt1.step(scaler=sc, zero_grad=True)
    assert opt1.step_count == 3 and sc.scale == 2.0 ** 11 and float(sc.state[1]) == 0.0          # two clean steps in a row: growth
    sd = sc.state_dict()
    sc2 = step_amd.LossScaler(torch.device(dev))
    sc2.load_state_dict(sd)
    assert sc2.scale == sc.scale and sc2.growth_interval == 2


def case_reg_unit_pack_follows_weight_updates(dev, golden):
    """The fused regressor unit (three Linear layers as one GEMM) in GRAD mode: its weight_fn is a fresh torch.cat per call, so
    the packed-weight cache must be keyed on the three parameters -- a temporary's (data_ptr, _version) can repeat after an
    optimizer step and silently serve stale weights to the forward while backward sees fresh ones.  Also: the stem's BN affine
    receives gradients when it is trainable (--freeze_affine False)."""
    net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
    net.set_device(dev)
    net.train()
    u = net._reg_unit()
    assert u is net._u_reg
    p1 = u.packed(torch.float32)
    assert u.packed(torch.float32) is p1                         # steady state: a cache hit, no re-pack per call
    p1 = p1.clone()
    for it in range(3):
        with torch.no_grad():
            net.neighbor_reg1.weight.add_(0.5)
        junk = [torch.empty_like(net.local_reg.weight) for _ in range(it)]      # perturb the allocator's reuse pattern
        p2 = net._reg_unit().packed(torch.float32)
        assert not torch.equal(p1, p2), it
        with torch.no_grad():
            fresh = net._reg_unit().packed(torch.float32)        # the no-grad path builds its own cat + pack
        assert torch.equal(p2, fresh), it
        p1 = p2.clone()
        del junk
    # stem BN affine gradient: trainable affine -> gradients equal the oracle's autograd
    base = fill(step_amd.BaseNet(cfg(freeze_affine=False))).to(dev)
    base.train()
    stem = base.base_model[0]
    assert stem.batch3d.weight.requires_grad
    x = R.fill_tensor("golden.stemaff.images", (1, 6, 3, 20, 20), "image").to(dev)
    y = stem(x)
    wgt = R.fill_tensor("golden.stemaff.w", tuple(y.shape), "feat").to(dev)
    (y * wgt).sum().backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
          for k, v in base.state_dict().items() if k.startswith("base_model.0.")}
    yo = R.unit3d(x.cpu().permute(0, 2, 1, 3, 4), sd, "base_model.0", stride=(2, 2, 2))
    (yo.permute(0, 2, 3, 4, 1) * wgt.cpu()).sum().backward()
    for k in ("conv3d.weight", "batch3d.weight", "batch3d.bias"):
        a = np_(dict(stem.named_parameters())[k].grad).astype(np.float64)
        b = sd["base_model.0." + k].grad.numpy().astype(np.float64)
        e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        assert e < 1e-3, (k, e)


def case_batched_repack_follows_weight_updates(dev, golden):
    """After an optimizer step every packed weight image (forward and data-gradient, channel slices and permutations included)
    is refreshed by ONE step_conv_pack_weights launch into the buffers the units already hold, bit-identical to packing each
    effective weight on its own; units that were never used are left alone; STEP_BATCH_PACK=0 restores the per-weight path."""
    from step_amd import backbone, ops
    g = golden("head_golden")
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)
    net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
    net.set_device(dev)
    net.train()
    mixed = backbone.Mixed(32, (16, 16, 24, 8, 16, 8)).to(dev).eval()
    for p in mixed.parameters():
        p.requires_grad_(p.dim() == 5)
    xm = R.fill_tensor("golden.repack.x", (1, 4, 6, 6, 32), "feat").to(dev).requires_grad_(True)
    calls = {"batched": 0, "single": 0, "single_d": 0}
    orig = (ops.pack_conv_weights, ops.pack_conv_weight, ops.pack_conv_weight_dgrad)

    def count(name, fn):
        def f(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return f
    ops.pack_conv_weights, ops.pack_conv_weight, ops.pack_conv_weight_dgrad = (count("batched", orig[0]), count("single", orig[1]),
                                                                              count("single_d", orig[2]))
    try:
        def step():
            o = net(pf, context_feat=cx, tubes=tubes, targets=targets)
            (o[4].mean() + 5.0 * o[5].mean() + o[6].mean() + mixed(xm).square().mean()).backward()

        def units():
            return [u for u in list(backbone._UNITS) if u.version_fn is None and u._packed
                    and any(u.owner is m for mod in (net, mixed) for m in mod.modules())]

        def check():
            n = 0
            for u in units():
                w = u.weight_fn()
                for key, (ver, buf) in u._packed.items():
                    assert ver == backbone._ver(w)
                    want = orig[1](u.effective_weight(), key[0]) if len(key) == 2 else orig[2](u.effective_weight(), key[0], key[2])
                    assert torch.equal(buf, want), (type(u.owner).__name__, key)
                    n += 1
            return n

        step()
        first = dict(calls)
        assert first["batched"] == 0 and first["single"] > 5 and first["single_d"] > 5       # nothing to refresh yet
        n_images = check()
        ptrs = {(id(u), k): v[1].data_ptr() for u in units() for k, v in u._packed.items()}
        for it in range(2):
            with torch.no_grad():
                for p in list(net.parameters()) + list(mixed.parameters()):
                    if p.requires_grad:
                        p.add_(0.03 * (it + 1))
            before = dict(calls)
            step()
            # one launch refreshed every image; the fused units keyed on several parameters (version_fn) pack their own
            assert calls["batched"] == before["batched"] + 1, calls
            assert calls["single_d"] - before["single_d"] <= 2 and calls["single"] - before["single"] <= 2, (before, calls)
            assert check() == n_images
            assert ptrs == {(id(u), k): v[1].data_ptr() for u in units() for k, v in u._packed.items()}   # refreshed in place
        # the per-weight path (STEP_BATCH_PACK=0)
        backbone.BATCH_PACK = False
        with torch.no_grad():
            for p in mixed.parameters():
                if p.requires_grad:
                    p.add_(0.01)
        before = dict(calls)
        (mixed(xm).square().mean()).backward()
        assert calls["batched"] == before["batched"] and calls["single"] > before["single"]
        check()
    finally:
        backbone.BATCH_PACK = True
        ops.pack_conv_weights, ops.pack_conv_weight, ops.pack_conv_weight_dgrad = orig


def case_stem_backward_16bit(dev, golden):
    """The stem's weight gradient with bf16 activations (frozen affine): ReLU mask x scale in one HIP pass and the 16-bit
    matrix-instruction kernel (step_stem_wgrad16).  Against the same bf16 run through the fp32-MFMA kernel (STEP_WGRAD16=0) the
    only difference is the bf16 rounding of the activation gradient (< 5e-3); both follow the fp32 run to the bf16 forward's own
    error (quantized clip, ReLU decisions: a few per cent).  A clip width outside the 16-bit kernel's contract (W % 8) takes the
    fp32-MFMA kernel."""
    from step_amd import backbone, ops
    for shape in ((2, 6, 3, 24, 40), (1, 5, 3, 18, 20)):
        grads = {}
        for tag, dt, w16 in (("f32", torch.float32, True), ("new", torch.bfloat16, True), ("old", torch.bfloat16, False)):
            base = fill(step_amd.BaseNet(cfg())).to(dev)
            base.train()
            stem = base.base_model[0]
            assert not stem.batch3d.weight.requires_grad
            x = R.fill_tensor("golden.stem16.images", shape, "image").to(dev).to(dt)
            used = []
            orig, keep = ops.stem_wgrad16, backbone.WGRAD16

            def spy(*a, **k):
                r = orig(*a, **k)
                used.append(r is not None)
                return r
            ops.stem_wgrad16, backbone.WGRAD16 = spy, w16
            try:
                y = stem(x)
                wgt = R.fill_tensor("golden.stem16.w", tuple(y.shape), "feat").to(dev)
                (y.float() * wgt).sum().backward()
            finally:
                ops.stem_wgrad16, backbone.WGRAD16 = orig, keep
            assert used == ([shape[-1] % 8 == 0] if tag == "new" else []), (shape, tag, used)
            grads[tag] = np_(stem.conv3d.weight.grad).astype(np.float64)
        b = grads["f32"]
        e_new, e_old = (float(np.linalg.norm(grads[k] - b) / np.linalg.norm(b)) for k in ("new", "old"))
        e_no = float(np.linalg.norm(grads["new"] - grads["old"]) / np.linalg.norm(b))
        assert np.isfinite(grads["new"]).all() and e_new < 6e-2 and e_new < 1.2 * e_old + 1e-3 and e_no < 5e-3, (shape, e_new, e_old, e_no)


def case_loss_masks_without_host_branches(dev, golden):
    """The heads' losses without the reference's `if mask.sum():` host branches (heads.SYNC_FREE_LOSSES, the default: the fused tail
    heads._HeadOutputsFn -- one launch forward, one backward -- and, with FUSED_HEAD_OUTPUTS off, its element-wise torch form) against
    the branching form: the element-wise free form equals it bit for bit when positives exist, the fused launch within 2e-6 (its sums
    run in a different, fixed order); a batch without positives gives exactly zero losses and zero gradients in all three forms (the
    reference's one-element zero classification loss becomes N*classes zeros: same mean)."""
    from step_amd import heads
    g = golden("head_golden")
    pf = R.fill_tensor("golden.det.pooled3", (2, 3, 832, 7, 7), "feat").to(dev)
    cx = R.fill_tensor("golden.det.ctx3", (2, 1024, 3, 1, 1), "feat").to(dev)
    tubes, targets = torch.from_numpy(g["loss_tubes"]).to(dev), torch.from_numpy(g["loss_targets"]).to(dev)
    empty = targets.clone()
    empty[:, :, 4:6] = 0
    keep, keepf = heads.SYNC_FREE_LOSSES, heads.FUSED_HEAD_OUTPUTS
    out = {}
    try:
        for free in (True, "torch", False):
            heads.SYNC_FREE_LOSSES = bool(free)
            heads.FUSED_HEAD_OUTPUTS = free is True
            for tag, tg in (("pos", targets), ("none", empty)):
                if free == "torch" and tag == "none" and dev == "cpu":
                    out[(free, tag)] = out[(True, tag)]                 # (interpreter time; the GPU run covers it)
                    continue
                net = fill(step_amd.TwoBranchNet(cfg()), "det0.").to(dev)
                net.set_device(dev)
                net.train()
                for m in net.modules():
                    if isinstance(m, torch.nn.Dropout):
                        m.p = 0.0
                o = net(pf, context_feat=cx, tubes=tubes, targets=tg)
                loss = o[4].mean() + 5.0 * o[5].mean() + o[6].mean()
                gsum = 0.0
                if loss.requires_grad:
                    loss.backward()
                    gsum = sum(float(p.grad.abs().sum()) for p in net.parameters() if p.grad is not None)
                out[(free, tag)] = ([np_(o[i]) for i in (4, 5, 6)], gsum)
    finally:
        heads.SYNC_FREE_LOSSES, heads.FUSED_HEAD_OUTPUTS = keep, keepf
    for i in range(3):
        assert np.array_equal(out[("torch", "pos")][0][i], out[(False, "pos")][0][i]), i
        assert np.abs(out[(True, "pos")][0][i] - out[(False, "pos")][0][i]).max() <= 2e-6 * max(1.0, np.abs(out[(False, "pos")][0][i]).max()), i
    for free in (True, "torch"):
        assert out[(free, "pos")][1] > 0 and abs(out[(free, "pos")][1] - out[(False, "pos")][1]) <= 1e-5 * out[(False, "pos")][1]
    for free in (True, "torch", False):
        ls, gsum = out[(free, "none")]
        assert all(float(np.abs(l).max()) == 0.0 for l in ls) and gsum == 0.0, (free, gsum)
    assert out[(False, "none")][0][0].size == 1 and out[(True, "none")][0][0].size == out[(True, "pos")][0][0].size


def _full_size_oracle(golden, tag, name, shape):
    """The oracle's fp32 conv_feat of the pinned full-size clip (seconds on the host), itself checked here against the digest the
    imported reference left in tests/golden/full_size_golden.npz."""
    g = golden("full_size_golden")
    xo = R.fill_tensor(name, shape, "image")
    with torch.no_grad():
        yo = R.basenet_forward(xo, R.fill_state_dict(R.backbone_shapes())).contiguous()
    f = yo.reshape(-1)
    st = int(g[tag + ".out_stats"][3])
    assert rel(f[::st][:4096].numpy(), g[tag + ".out_sample"]) < 1e-5
    return xo, yo.numpy()


def case_c2_full_size_properties(dev, golden):
    """BASELINE C2 at its full size (8 x [32,3,224,224], bf16), the config the bench number is quoted on:
      * ORACLE PARITY AT THIS SIZE: clip 3 of the batch is the pinned clip `golden.c2.images`; the fp32 HIP run of it is held to the
        oracle's full [1,8,832,14,14] tensor at 1e-3 (north_star's bound), and the bf16 / fp16 errors are measured AGAINST THE ORACLE
        (not against the HIP fp32 run), recorded and bounded; the oracle itself is pinned at this shape by full_size_golden.npz
        (digests of the imported reference, models/networks.py:69-83);
      * clip independence: every tensor on the path is per clip, so a clip's features must not depend on which batch
        it travels in (the launch planner picks other tile shapes / channel groupings for other batch sizes; the K
        order of every accumulation is the same) -> the batch-of-8 result equals the clip computed alone, BIT-EXACT;
      * permutation equivariance of the batch axis (bit-exact);
      * the fusion toggles leave the batch bit-identical."""
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    g = torch.Generator().manual_seed(123)
    x = torch.rand(8, 32, 3, 224, 224, generator=g) * 2 - 1
    xo, yo = _full_size_oracle(golden, "c2", "golden.c2.images", (1, 32, 3, 224, 224))
    x[3:4] = xo
    x = x.to(dev)
    xb = x.to(torch.bfloat16)
    with torch.no_grad():
        y8 = net(xb).clone()
        assert tuple(y8.shape) == (8, 8, 832, 14, 14) and bool(torch.isfinite(y8.float()).all())
        y1 = net(xb[3:4]).clone()
        assert torch.equal(y8[3:4], y1), float((y8[3:4].float() - y1.float()).abs().max())
        perm = torch.tensor([5, 2, 7, 0, 3, 6, 1, 4], device=dev)
        yp = net(xb[perm])
        assert torch.equal(yp, y8[perm])
        yf = net(x[3:4])
        ef = rel(np_(yf), yo)
        record("c2_full_size_fp32_vs_oracle", ef)
        assert ef < 1e-3, ef                                       # north_star: within 1e-3 rel of the reference CPU path (measured ~1e-6)
        eb = rel(np_(y1), yo)
        record("c2_full_size_bf16_vs_oracle", eb)
        assert eb < 1e-2, eb                                       # one bf16 rounding per layer over 45 layers (C1: 5.0e-3 of 8e-3)
        eh = rel(np_(net(x[3:4].half())), yo)
        record("c2_full_size_fp16_vs_oracle", eh)
        assert eh < 1e-3, eh                                       # fp16 storage is INSIDE north_star's 1e-3 at the headline size (measured 9.3e-4); bench.py reports its rate under "fp16"
        e = rel(np_(y1), np_(yf))
        record("c2_full_size_bf16_vs_fp32", e)
        assert e < 1e-2, e
        # conv3d_2b evaluated inside conv3d_2c's launch (backbone.FUSE_POINTWISE_INPUT, ops.conv_forward_pre) == the two units
        # launched one after the other, BIT-EXACT (same K order, same 16-bit rounding of the tensor between them)
        # ... and the 14x14 blocks' pool + fused 1x1x1 triple in one grid (backbone.POOL_WITH_POINTWISE, ops.pool_conv_forward) == two launches
        # ... and maxPool3d_2a taken on the stem's tiles (backbone.FUSE_STEM_POOL, ops.stem_pool_forward: 7 x 7 tiles per frame, seams
        # completed by the second launch) == the stem and the pool as two launches
        from step_amd import backbone as _bb
        # ... and maxPool3d_3a taken on conv3d_2c's tiles (backbone.FUSE_CONV_POOL, ops.conv_forward_pre_pool: 49 tiles of 8 x 8 per plane,
        # seams completed by pool_seam_fix_kernel) == conv3d_2c and the pool as two calls
        assert _bb.FUSE_POINTWISE_INPUT and _bb.POOL_WITH_POINTWISE and _bb.FUSE_STEM_POOL and _bb.FUSE_CONV_POOL
        try:
            _bb.FUSE_CONV_POOL = False
            y8p = net(xb)
            _bb.FUSE_POINTWISE_INPUT = False
            _bb.POOL_WITH_POINTWISE = False
            _bb.FUSE_STEM_POOL = False
            y8u = net(xb)
        finally:
            _bb.FUSE_POINTWISE_INPUT = True
            _bb.POOL_WITH_POINTWISE = True
            _bb.FUSE_STEM_POOL = True
            _bb.FUSE_CONV_POOL = True
        assert torch.equal(y8p, y8), float((y8p.float() - y8.float()).abs().max())
        assert torch.equal(y8u, y8), float((y8u.float() - y8.float()).abs().max())


def case_c5_full_size_properties(dev, golden):
    """BASELINE C5 at one GPU's share (4 x [64,3,400,400], fp16) -- the long-clip stress shape: ORACLE PARITY AT THIS SIZE (clip 2 is the
    pinned clip `golden.c5.images`: fp32 HIP within 1e-3 of the oracle's full [1,16,832,25,25] tensor, the fp16 error against the oracle
    recorded and bounded; the oracle pinned by full_size_golden.npz), clip independence (a clip alone == the clip inside the batch of 4,
    BIT-EXACT, although the planner tiles the two launches differently), finite outputs of the right shape."""
    net = fill(step_amd.BaseNet(cfg())).to(dev).eval()
    g = torch.Generator().manual_seed(321)
    x = torch.rand(4, 64, 3, 400, 400, generator=g) * 2 - 1
    xo, yo = _full_size_oracle(golden, "c5", "golden.c5.images", (1, 64, 3, 400, 400))
    x[2:3] = xo
    x = x.to(dev)
    xh = x.to(torch.float16)
    with torch.no_grad():
        y4 = net(xh).clone()
        assert tuple(y4.shape) == (4, 16, 832, 25, 25) and y4.dtype == torch.float16 and bool(torch.isfinite(y4.float()).all())
        y1 = net(xh[2:3]).clone()
        assert torch.equal(y4[2:3], y1), float((y4[2:3].float() - y1.float()).abs().max())
        yf = net(x[2:3])
        ef = rel(np_(yf), yo)
        record("c5_full_size_fp32_vs_oracle", ef)
        assert ef < 1e-3, ef
        eh = rel(np_(y1), yo)
        record("c5_full_size_fp16_vs_oracle", eh)
        assert eh < 4e-3, eh
        e = rel(np_(y1), np_(yf))
        record("c5_full_size_fp16_vs_fp32", e)
        assert e < 4e-3, e


def case_nms_operator_api(dev, golden):
    """`step_amd.roi_layers.nms` (reference: roi_layers/nms.py:38, csrc/nms.h:34-52) as the reference's callers use it: CPU tensors
    (test.py:158-160), GPU tensors, double boxes (nms_cpu.cpp:95 dispatch), 16-bit boxes (apex float_function), empty input --
    the kept ORIGINAL indices, ascending, int64 on the CPU, bit-exact against the reference operator's restatement."""
    import importlib
    from step_amd.roi_layers import nms
    mod = importlib.import_module("step_amd.roi_layers.nms")
    saved = mod._device
    if dev == "cpu":                                                # interpreter build: "the device" of the upload path is the host (test-side patch)
        mod._device = lambda: torch.device("cpu")
    try:
        _nms_operator_api(dev, nms)
    finally:
        mod._device = saved


def _nms_operator_api(dev, nms):
    rs = np.random.RandomState(77)
    for k in (1, 7, 40, 333):
        c = rs.rand(k, 2) * 200
        wh = rs.rand(k, 2) * 80 + 4
        boxes = np.concatenate([c, c + wh], 1)
        boxes[k // 2] = boxes[0]                                    # an exact duplicate: IoU = 1 >= thr
        scores = rs.rand(k)
        scores[k // 3] = scores[0]                                  # a score tie (stable order decides)
        b32, s32 = boxes.astype(np.float32), scores.astype(np.float32)
        want = oracle.nms(b32, s32, 0.4)
        for place in ("cpu", dev):
            got = nms(torch.from_numpy(b32).to(place), torch.from_numpy(s32).to(place), 0.4)
            assert got.device.type == "cpu" and got.dtype == torch.int64 and np.array_equal(got.numpy(), want), (k, place)
        got64 = nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.4)
        assert np.array_equal(got64.numpy(), oracle.nms_f64(boxes, scores, 0.4)), k
        bh = torch.from_numpy(b32).half()
        goth = nms(bh.to(dev), torch.from_numpy(s32).to(dev), 0.4)
        assert np.array_equal(goth.numpy(), oracle.nms(bh.float().numpy(), s32, 0.4)), k
    e = nms(torch.zeros(0, 4), torch.zeros(0), 0.4)
    assert e.numel() == 0 and e.dtype == torch.int64


CPU_CASES = ["case_state_dict_contract", "case_mixed_golden", "case_basenet_c1_golden", "case_context_golden",
             "case_twobranch_T3_and_losses_golden", "case_roinet_layouts", "case_training_step_matches_torch_autograd",
             "case_training_step_generic_weights",
             "case_flat_adam_matches_torch", "case_wgrad_into_and_targets", "case_twobranch_variants_golden",
             "case_reg_unit_pack_follows_weight_updates", "case_basenet_backward_matches_oracle_autograd",
             "case_contextnet_backward_matches_oracle_autograd", "case_basenet_batch_statistics_bn_golden", "case_postprocess_golden",
             "case_batched_repack_follows_weight_updates", "case_stem_backward_16bit",
             "case_loss_masks_without_host_branches", "case_data_parallel_replicas", "case_train_select_device_front_end", "case_nms_operator_api",
             "case_resample_bottleneck_concat_in_one_launch"]
GPU_CASES = CPU_CASES + ["case_basenet_forward_u8", "case_conv_pool_fusion_in_basenet", "case_fp16_training_step_loss_scaling", "case_base_context_chain_backward", "case_wgrad_into_grad_matches_autograd", "case_training_iteration_with_selection", "case_training_step_16bit_storage", "case_c2_full_size_properties", "case_c5_full_size_properties", "case_basenet_c1_16bit_error", "case_twobranch_T9_golden", "case_i3d_classifier_golden", "case_inference_golden", "case_inference_modes_golden", "case_inference_golden_34", "case_e2e_c3_golden"]
