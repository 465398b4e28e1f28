"""oracle/check_callers.py -- BUILD-CONTAINER ONLY: the reference's own call sites, unchanged, on top of the drop-in.

"Drops into train.py / test.py / demo.py unchanged" is otherwise argued from signatures and state_dict keys.  This script
runs the reference's OWN `utils.utils.inference` (utils/utils.py:15-131 -- flatten_tubes, the per-tube `.item()` context
gather, the numpy round trip, valid_tubes) with `dropin/` ahead of the reference on sys.path, so that

    from models import ROINet, TwoBranchNet                      -> dropin/models/__init__.py            -> step_amd
    from external.maskrcnn_benchmark.roi_layers import ROIAlign   -> dropin/external/.../__init__.py      -> step_amd.roi_layers

resolve to this repository, with the kernels executed by the host SIMT interpreter (tests/emul) at a reduced feature size, and
compares the `history` it returns with `step_amd.driver.inference` on the same inputs (tolerance 1e-5: the same kernels, a
different host glue).  Nothing of the reference travels to the GPU box; this is test infrastructure.

    python -m oracle.check_callers
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STEP_REFERENCE", "/root/reference")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("the reference tree is not available (%s): build-container check only" % REF)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "dropin"))          # dropin first: `models` and `external...roi_layers` are ours
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import models                                              # dropin/models -> step_amd
    import external.maskrcnn_benchmark.roi_layers as roi_layers
    import utils.utils as ref_utils                            # the REFERENCE's driver
    import step_amd
    from oracle import i3d_ref as R
    from step_amd import driver
    from tests.emul.patch import emulated_kernels

    assert models.TwoBranchNet is step_amd.TwoBranchNet and models.ROINet is step_amd.ROINet
    assert roi_layers.ROIAlign.__module__.startswith("step_amd")
    assert ref_utils.__file__.startswith(REF)
    # (CPU-only artefact of the reference: `.cpu().numpy()` aliases the recorded predictions, which valid_tubes then clamps in
    # place; its GPU path copies.  Give valid_tubes a copy, as oracle/make_golden.py does.)
    _vt = ref_utils.valid_tubes
    ref_utils.valid_tubes = lambda tubes, *a, **kw: _vt(tubes.copy(), *a, **kw)

    args = types.SimpleNamespace(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False, T=3,
                                 num_classes=60, fc_dim=256, dropout=0.0, pool_size=7, no_context=False, max_iter=3,
                                 NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3}, temporal_mode="predict", image_size=(400, 400), pool_mode="align")
    torch.set_num_threads(8)
    with emulated_kernels(), torch.no_grad():
        nets = {"roi_net": models.ROINet(args.pool_mode, args.pool_size).eval()}      # (test.py:67)
        for i in range(args.max_iter):
            net = models.TwoBranchNet(args)
            shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            net.load_state_dict(R.fill_state_dict(shapes, "det%d." % i))
            nets["det_net%d" % i] = net.eval()
        B, K = 1, 2
        conv_feat = R.fill_tensor("callers.feat", (B, 9, 832, 25, 25), "feat")
        context = R.fill_tensor("callers.ctx", (B, 1024, 9, 1, 1), "feat")
        rs = np.random.RandomState(3)
        tubes = []
        for _ in range(B):
            xy = rs.uniform(20, 200, (K, 1, 2))
            box = np.concatenate([xy, xy + rs.uniform(80, 180, (K, 1, 2))], 2).astype(np.float32)
            tubes.append(np.tile(box, (1, args.T, 1)))
        h_ref, _ = ref_utils.inference(args, conv_feat, context, nets, args.max_iter, [t.copy() for t in tubes])
        h_own, _ = driver.inference(args, conv_feat, context, nets, args.max_iter, [t.copy() for t in tubes])
    worst = 0.0
    for i, (a, b) in enumerate(zip(h_ref, h_own)):
        assert a["tubes_nums"] == b["tubes_nums"]
        for k in ("pred_prob", "pred_loc", "pred_first_loc", "pred_last_loc"):
            x, y = a[k].float().numpy(), b[k].float().numpy()
            assert x.shape == y.shape, (i, k, x.shape, y.shape)
            e = float(np.abs(x - y).max() / max(np.abs(x).max(), 1e-30))
            worst = max(worst, e)
            assert e < 1e-5, (i, k, e)
    print("reference utils.utils.inference over dropin/ models == step_amd.driver.inference: %d iterations, worst rel. difference %.2e" %
          (len(h_ref), worst))

    # ---- the wrapper code of test.py:62-98 / train.py:129-148 / demo.py:62-96, line for line, over the drop-in: nets built from
    # `models`, base_net / context_net wrapped in torch.nn.DataParallel UNCONDITIONALLY, checkpoints (whose keys carry the wrapper's
    # "module." prefix, train.py:380-381) loaded THROUGH the wrapper, `.eval()` on every entry, then
    # nets['base_net'](images) -> nets['context_net'](conv_feat) -> the reference's inference().  With no / one visible device
    # DataParallel calls the module itself; tests/module_cases.py::case_data_parallel_replicas covers the replicated path.
    from collections import OrderedDict
    with emulated_kernels(), torch.no_grad():
        nets = OrderedDict()
        nets["base_net"] = models.BaseNet(args)
        nets["roi_net"] = models.ROINet(args.pool_mode, args.pool_size)
        for i in range(args.max_iter):
            nets["det_net%d" % i] = models.TwoBranchNet(args)
        nets["context_net"] = models.ContextNet(args)
        checkpoint = {}
        for key in ("base_net", "context_net"):
            shapes = {k: tuple(v.shape) for k, v in nets[key].state_dict().items()}
            checkpoint[key] = {"module." + k: v for k, v in R.fill_state_dict(shapes).items()}
        for i in range(args.max_iter):
            shapes = {k: tuple(v.shape) for k, v in nets["det_net%d" % i].state_dict().items()}
            checkpoint["det_net%d" % i] = R.fill_state_dict(shapes, "det%d." % i)
        nets["base_net"] = torch.nn.DataParallel(nets["base_net"])
        nets["context_net"] = torch.nn.DataParallel(nets["context_net"])
        for i in range(args.max_iter):
            nets["det_net%d" % i].set_device("cpu")
        nets["base_net"].load_state_dict(checkpoint["base_net"])
        nets["context_net"].load_state_dict(checkpoint["context_net"])
        for i in range(args.max_iter):
            nets["det_net%d" % i].load_state_dict(checkpoint["det_net%d" % i])
        for _, net in nets.items():
            net.eval()
        images = R.fill_tensor("golden.c1.images", (1, 8, 3, 112, 112), "image")
        feat = nets["base_net"](images)                         # test.py:141
        g = np.load(os.path.join(ROOT, "tests", "golden", "i3d_c1_golden.npz"))["conv_feat"]
        e1 = float(np.abs(feat.float().numpy() - g).max() / np.abs(g).max())
        assert tuple(feat.shape) == g.shape and e1 < 1e-4, e1
        context_feat = nets["context_net"](conv_feat[:, :3])  # test.py:144 (three frames: the interpreter's budget)
        gc = np.load(os.path.join(ROOT, "tests", "golden", "head_golden.npz"))
        assert tuple(context_feat.shape) == (B, 1024, 3, 1, 1)
        h2, _ = ref_utils.inference(args, conv_feat, context, nets, args.max_iter, [t.copy() for t in tubes])      # test.py:152
        for a, b in zip(h2, h_ref):
            assert torch.equal(a["pred_prob"], b["pred_prob"]) and torch.equal(a["pred_loc"], b["pred_loc"])
    print("test.py's wrapper code (DataParallel(base_net) / DataParallel(context_net), module.-prefixed checkpoints, .eval()) over dropin/: "
          "base_net C1 rel. error %.2e, inference history identical" % e1)


if __name__ == "__main__":
    main()
