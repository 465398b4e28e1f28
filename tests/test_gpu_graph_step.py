"""The whole C4 training step captured in one HIP graph (step_amd.workloads.C4TrainStep.capture) against the same steps launched
eagerly: same parameter trajectory (up to the fp32 summation order of the weight-gradient atomics), the device-side Adam step
counter advances on every replay, the loss tensor is refreshed in place."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_captured_training_step_follows_the_eager_steps(dtype):
    from step_amd import workloads

    dev = torch.device("cuda:0")
    steps, warm = 5, 2
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(7)
        w = workloads.C4TrainStep(dev, batch=1, seed=123, dtype=dtype, capturable=(mode == "graph"))
        p0 = w.opt.flat_param.clone()
        losses = []
        if mode == "graph":
            w.capture(warmup=warm)                              # runs `warm` eager steps, records one more
            assert w.opt.step_count == warm
            for _ in range(steps - warm):
                losses.append(float(w.step()))
            assert w.graph is not None and w.opt.step_count == steps
        else:
            for i in range(steps):
                l = float(w.step())
                if i >= warm:
                    losses.append(l)
        torch.cuda.synchronize()
        runs[mode] = ((w.opt.flat_param - p0).double().cpu().numpy(), np.array(losses), w.opt.exp_avg.double().cpu().numpy())
        del w
        torch.cuda.empty_cache()
    (da, la, ma), (db, lb, mb) = runs["eager"], runs["graph"]
    assert np.isfinite(db).all() and np.abs(db).max() > 0
    # Adam normalises every gradient: a parameter whose gradient is at the noise floor can move either way, so compare the
    # update DIRECTION over the whole arena and the first moment (linear in the gradients)
    cos = float((da * db).sum() / (np.linalg.norm(da) * np.linalg.norm(db)))
    em = float(np.linalg.norm(ma - mb) / np.linalg.norm(ma))
    tol_l = 2e-2 if dtype == torch.bfloat16 else 1e-3
    assert cos > 0.995 and em < 2e-2, (cos, em)
    assert np.all(np.abs(la - lb) <= tol_l * np.abs(la)), (la, lb)
    assert len(set(np.round(lb, 10))) > 1                       # the replays really advance the weights
