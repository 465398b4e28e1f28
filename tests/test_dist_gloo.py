"""The N>1 path on CPU: two gloo ranks (world_size 2).  Checks that clip sharding + the bucketed
gradient all-reduce reproduce the single-process gradients on the concatenated batch, that the
weighted form reproduces a global masked mean, and the bench timing reduction (max over ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv3d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Flatten(), torch.nn.Linear(4 * 2 * 4 * 4, 5))


class _DirectLinear(torch.autograd.Function):
    """Mimics the conv units under backbone.wgrad_into_grad(): the weight gradient is ADDED to weight.grad by the op itself,
    announced through backbone.GRAD_READY, and autograd gets None for it (its post-accumulate hook still fires for that
    parameter -- with an undefined gradient -- which the reducer must not count as a second gradient)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        from step_amd import backbone
        x, w = ctx.saved_tensors
        w.grad.add_(g.t() @ x)
        if backbone.GRAD_READY is not None:
            backbone.GRAD_READY(w, None)
        return g @ w, None


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from step_amd import dist as D
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(1)
    clips = torch.randn(6, 3, 2, 4, 4)
    target = torch.randn(6, 5)
    mask = torch.tensor([1., 1, 0, 1, 0, 1])
    model = _model()
    if rank == 1:                                  # replicas must be made identical by the broadcast
        for p in model.parameters():
            p.data.add_(1.0)
    D.broadcast_parameters([model])
    idx = D.shard_clips(6, rank, world)
    assert idx == list(range(rank, 6, world))
    # (a) plain mean loss: equal shard sizes -> average of rank means == global mean
    model.zero_grad()
    ((model(clips[idx]) - target[idx]) ** 2).mean().backward()
    nb = D.allreduce_gradients(list(model.parameters()), bucket_bytes=256)     # tiny buckets: several all-reduces
    g_plain = [p.grad.clone() for p in model.parameters()]
    # (b) masked mean: weight each rank by its mask count
    model.zero_grad()
    m = mask[idx]
    (((model(clips[idx]) - target[idx]) ** 2).mean(1) * m).sum().div(m.sum()).backward()
    D.allreduce_gradients(list(model.parameters()), weight=float(m.sum()))
    g_masked = [p.grad.clone() for p in model.parameters()]
    # (c) the flat-arena exchange of step_amd.optim.FlatAdam: ONE all-reduce (SUM) + the factor for the optimizer pass
    model.zero_grad()
    ((model(clips[idx]) - target[idx]) ** 2).mean().backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    f_plain = D.allreduce_flat(flat)
    flat_plain = flat * f_plain
    model.zero_grad()
    (((model(clips[idx]) - target[idx]) ** 2).mean(1) * m).sum().div(m.sum()).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    f_w = D.allreduce_flat(flat, weight=float(m.sum()), chunk_bytes=64)     # tiny chunks: several collectives
    flat_w = flat * f_w
    # (d) the OVERLAPPED exchange (BucketedReducer): buckets of FlatAdam's arena all-reduced from autograd's
    # post-accumulate hooks while backward is still running, in a fixed issue order; a parameter that gets no gradient on
    # ONE rank only (rank 1 skips the extra head) must not change the sequence of collectives.  The optimizer's arenas
    # live on the CPU here through the test-only interpreter patch (FlatAdam itself refuses host tensors).
    from tests.emul.patch import emulated_kernels
    from step_amd.optim import FlatAdam
    with emulated_kernels():
        extra = torch.nn.Linear(5, 3)
        torch.manual_seed(3)
        torch.nn.init.normal_(extra.weight)
        params = list(model.parameters()) + list(extra.parameters())
        opt = FlatAdam(params, lr=1e-3)
        red = D.BucketedReducer(opt, bucket_bytes=256)
        nbuckets = len(red.buckets)
        outs = []
        for use_weight in (False, True):
            opt.zero_grad()
            red.begin(weight=float(m.sum()) if use_weight else None)
            y = model(clips[idx])
            if use_weight:
                loss = (((y - target[idx]) ** 2).mean(1) * m).sum().div(m.sum())
            else:
                loss = ((y - target[idx]) ** 2).mean()
            if rank == 0:
                loss = loss + 0.5 * ((_DirectLinear.apply(y.detach(), extra.weight) + extra.bias) ** 2).mean()
            loss.backward()
            during = red.issued_during_backward
            f = red.finish()
            outs.append(((opt.flat_grad * f).numpy().copy(), during, f))
        # a second backward between begin() and finish() (gradient accumulation, retain_graph) must fail loudly: the later gradient
        # would be added locally after its bucket has left (every rank runs this, so the collectives stay matched)
        opt.zero_grad()
        red.begin()
        y = model(clips[idx])
        l2 = ((y - target[idx]) ** 2).mean()
        l2.backward(retain_graph=True)
        try:
            l2.backward()
            second = "accepted"
        except RuntimeError as e:
            second = "refused" if "one backward per begin()" in str(e) else repr(e)
        red.finish()
        assert second == "refused", second
        red.close()
        offs = [(o, n) for _, _, o, n in opt._entries]
    el = D.timed_steps(lambda: None if rank == 0 else __import__("time").sleep(0.05), 2, sync=lambda: None)
    if rank == 0:
        # numpy arrays travel by value; torch tensors would be shared through file descriptors that may be gone
        # by the time the parent unpickles them
        q.put((nb, [g.numpy() for g in g_plain], [g.numpy() for g in g_masked], el, flat_plain.numpy(), flat_w.numpy(), f_plain,
               outs, offs, nbuckets))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_allreduce_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    nb, g_plain, g_masked, el, flat_plain, flat_w, f_plain, outs, offs, nbuckets = q.get()
    g_plain = [torch.from_numpy(g) for g in g_plain]
    g_masked = [torch.from_numpy(g) for g in g_masked]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    torch.manual_seed(1)
    clips = torch.randn(6, 3, 2, 4, 4)
    target = torch.randn(6, 5)
    mask = torch.tensor([1., 1, 0, 1, 0, 1])
    model = _model()
    model.zero_grad()
    ((model(clips) - target) ** 2).mean().backward()
    for a, p in zip(g_plain, model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-6)
    model.zero_grad()
    (((model(clips) - target) ** 2).mean(1) * mask).sum().div(mask.sum()).backward()
    for a, p in zip(g_masked, model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-6)
    assert f_plain == 0.5
    assert np.allclose(flat_plain, torch.cat([g.reshape(-1) for g in g_plain]).numpy(), rtol=1e-5, atol=1e-6)
    assert np.allclose(flat_w, torch.cat([g.reshape(-1) for g in g_masked]).numpy(), rtol=1e-5, atol=1e-6)
    # (d) overlapped buckets == the single-shot exchange on the model's own parameters (the extra head trains on rank 0 only)
    assert nbuckets >= 3
    for (arena, during, f), want in zip(outs, (g_plain, g_masked)):
        for (o, n), g in zip(offs, want):
            assert np.allclose(arena[o:o + n], g.reshape(-1).numpy(), rtol=1e-5, atol=1e-6)
        assert during >= 1                           # at least one bucket left while backward was still running
    assert outs[0][2] == 0.5 and abs(outs[1][2] - 1.0 / float(mask.sum())) < 1e-7
    assert nb >= 2                                   # bucketing really split the gradient
    assert el >= 0.1                                 # max over ranks: rank 1 slept 2 x 50 ms
