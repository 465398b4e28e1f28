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


if __name__ == "__main__":
    main()
