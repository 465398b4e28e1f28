"""step_amd/dist.py -- one-process-per-GPU data parallelism for the STEP hot path.

The reference's only parallelism is nn.DataParallel (one process, per-iteration parameter broadcast,
scatter/gather, reduce-add to GPU 0) plus manual placement of head i on GPU (i+1) % G
(train.py:142-148).  Here every rank owns a full replica and its own slice of the clip batch -- every
tensor on the path is per clip, BN is frozen, so:
  * inference (BASELINE configs C2, C3, C5): NO collective on the data path; ranks are replicas and
    clips/s adds up (bench.py);
  * training (C4): exactly one exchange per step -- the gradient average -- done as a few large
    flattened all-reduces (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm).  xGMI is point-to-point,
    7 links per GPU, so a ring all-reduce is per-link bound: buckets are kept large (default 64 MiB) to
    amortise latency, and the 177.7 MB of fp32 gradients of the full model is 3 buckets.
The same code runs over gloo on CPU (tests/test_dist_gloo.py).
"""
import os
import time

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise the default process group from the torchrun environment (RANK/WORLD_SIZE/MASTER_*)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kw = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_clips(n_clips, rank, world):
    """Indices of the clips rank `rank` processes: r, r+world, r+2*world, ...."""
    return list(range(rank, n_clips, world))


def broadcast_parameters(modules, src=0):
    """Make every replica start from rank `src`'s parameters and buffers (once, at start-up)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


def allreduce_flat(flat, weight=None, chunk_bytes=512 << 20):
    """SUM-all-reduce the contiguous gradient arena of step_amd.optim.FlatAdam in place -- one collective (a few for
    arenas beyond chunk_bytes), no bucket copies -- and return the factor that turns the sum into the average, to be
    handed to FlatAdam.step(grad_scale=...) so the division rides in the optimizer pass.

    weight: optional per-rank scalar as in allreduce_gradients(); the local arena is pre-multiplied by it and the
    returned factor is 1 / sum_r(weight_r)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 1.0
    factor = 1.0 / dist.get_world_size()
    if weight is not None:
        wsum = torch.tensor([float(weight)], device=flat.device, dtype=torch.float32)
        dist.all_reduce(wsum)
        flat.mul_(float(weight))
        factor = 1.0 / float(wsum.item())
    step = max(1, chunk_bytes // flat.element_size())
    for o in range(0, flat.numel(), step):
        dist.all_reduce(flat[o:o + step])
    return factor


class BucketedReducer:
    """The gradient exchange of the training step, overlapped with the backward pass (replaces the
    reduce-add of nn.DataParallel, train.py:142-148).

    FlatAdam's gradient arena is cut into contiguous buckets of ~bucket_bytes.  Backward produces gradients roughly from
    the END of the arena (heads) to its FRONT (stem), so buckets are exchanged in descending arena order: as soon as
    every tensor of the next bucket in that order has its gradient (autograd's post-accumulate hook, or the conv units'
    direct side-stream accumulation under backbone.wgrad_into_grad()), its in-place SUM all-reduce is issued on a
    communication stream that waits for the producing streams only -- the rest of backward keeps running on the main
    stream.  xGMI rings are per-link bound: a bucket is one large message (default 32 MiB, 6 for the 178 MB arena), not
    one message per tensor.

    The ISSUE ORDER is fixed (bucket B-1, B-2, ..., 0) whatever the readiness order, and finish() issues whatever backward
    did not reach (parameters without a gradient this step), so every rank issues the same sequence of collectives --
    RCCL matches collectives by order, not by address.

        red = BucketedReducer(opt)            # once
        red.begin(weight=None)                # before backward
        with wgrad_into_grad(): loss.backward()
        scale = red.finish()                  # joins the communication stream
        opt.step(grad_scale=scale, zero_grad=True)
    """

    def __init__(self, opt, bucket_bytes=32 << 20, single_rank=False):
        # single_rank: issue the collectives even in a ONE-rank process group (they are identities there) -- lets the whole
        # RCCL path (communicator on this device, communication stream, bucket order, stream waits) run on a one-GPU box
        self.opt = opt
        self.flat = opt.flat_grad
        entries = opt._entries                                  # (group, param, offset, numel), ascending offsets
        cap = max(1, bucket_bytes // 4)
        self.buckets = []                                        # [lo, hi, n_tensors]
        self.bucket_of = {}
        lo, cnt = 0, 0
        for k, (_, p, o, n) in enumerate(entries):
            end = entries[k + 1][2] if k + 1 < len(entries) else opt.numel
            self.bucket_of[id(p)] = len(self.buckets)
            cnt += 1
            if end - lo >= cap or k + 1 == len(entries):
                self.buckets.append((lo, end, cnt))
                lo, cnt = end, 0
        self.active = dist.is_initialized() and (dist.get_world_size() > 1 or bool(single_rank))
        self.cuda = self.flat.is_cuda
        self.comm = torch.cuda.Stream(self.flat.device) if self.cuda else None
        self._hooks = [p.register_post_accumulate_grad_hook(self._autograd_ready) for _, p, _, _ in entries]
        self._armed = False
        self.issued_during_backward = 0

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        from . import backbone
        if backbone.GRAD_READY is self.ready:
            backbone.GRAD_READY = None
            backbone.GRAD_DEFER = None

    def begin(self, weight=None):
        """Arm for one backward pass.  weight: optional per-rank scalar as in allreduce_flat()."""
        from . import backbone
        self.pending = [c for _, _, c in self.buckets]
        self.seen = set()
        self.deferred = set()                                    # parameters whose gradient a conv unit will announce itself, later (see defer())
        self._hooked = {}                                        # autograd hook firings per parameter in this armed pass
        self.next = len(self.buckets) - 1                        # next bucket to issue (descending)
        self.works = []
        self.side = []                                           # producer streams to wait for, per bucket
        self.side_streams = [set() for _ in self.buckets]
        self.weight = None
        self.factor = 1.0
        self.issued_during_backward = 0
        if self.active:
            self.factor = 1.0 / dist.get_world_size()
            if weight is not None:
                wsum = torch.tensor([float(weight)], device=self.flat.device, dtype=torch.float32)
                dist.all_reduce(wsum)
                self.weight = float(weight)
                self.factor = 1.0 / float(wsum.item())
        backbone.GRAD_READY = self.ready
        backbone.GRAD_DEFER = self.defer
        self._armed = True

    def _autograd_ready(self, p):
        # (autograd runs this hook for EVERY leaf it reaches, also when the gradient that arrives is undefined -- which is
        # what the conv units return for a weight whose gradient they accumulated themselves and announced already)
        if not self._armed:
            return
        n = self._hooked.get(id(p), 0) + 1
        self._hooked[id(p)] = n
        if id(p) in self.deferred:
            return                                               # the unit that accumulates this gradient announces it when its last kernel has been launched
        if n > 1:
            # a second backward() (gradient accumulation, retain_graph) between begin() and finish(): the later gradient would be
            # added locally after the parameter's bucket may already have been all-reduced, and the ranks would diverge silently
            raise RuntimeError("BucketedReducer: a second backward pass reached the same parameter before finish() -- "
                               "one backward per begin() (accumulate micro-batches with the exchange off, then begin() for the last one)")
        if id(p) not in self.seen:
            self.ready(p, None)

    def defer(self, p):
        """A conv unit accumulates `p`'s gradient itself and its last kernel (the grouped fixed-order sum of partial tiles,
        backbone._flush_wgrads) is launched LATER than the unit's autograd node returns: autograd's hook, which fires for the leaf even
        though no gradient arrives through it, must not count the parameter -- the unit calls ready() when the sum has been launched."""
        if self._armed:
            self.deferred.add(id(p))

    def ready(self, p, stream=None):
        """`p`'s gradient for this step is complete (as far as the host is concerned: the producing kernel has been
        launched on `stream`, None = the current stream)."""
        if not self._armed:
            return
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        if id(p) in self.seen:
            raise RuntimeError("BucketedReducer: a second gradient for the same parameter in one backward pass "
                               "(its bucket may already be in flight)")
        self.seen.add(id(p))
        if stream is not None:
            self.side_streams[b].add(stream)
        self.pending[b] -= 1
        while self.next >= 0 and self.pending[self.next] == 0:
            self._issue(self.next)
            self.issued_during_backward += 1
            self.next -= 1

    def _issue(self, b):
        lo, hi, _ = self.buckets[b]
        if not self.active:
            return
        seg = self.flat[lo:hi]
        if self.cuda:
            main = torch.cuda.current_stream(self.flat.device)
            self.comm.wait_stream(main)                          # every gradient kernel issued so far on the main stream
            for s_ in self.side_streams[b]:
                self.comm.wait_stream(s_)                        # ... and the direct weight-gradient accumulations
            with torch.cuda.stream(self.comm):
                if self.weight is not None:
                    seg.mul_(self.weight)
                self.works.append(dist.all_reduce(seg, async_op=True))
        else:
            if self.weight is not None:
                seg.mul_(self.weight)
            self.works.append(dist.all_reduce(seg, async_op=True))

    def finish(self):
        """Issue the buckets backward did not complete (same order on every rank), wait for all of them, return the
        factor that turns the sums into the average (for FlatAdam.step(grad_scale=...))."""
        from . import backbone
        if self.cuda:
            # queued weight-gradient sums are launched (and announce their parameters: buckets may still leave here); late
            # side-stream accumulations are ordered before the main stream
            backbone.wgrad_sync()
        self._armed = False
        backbone.GRAD_READY = None
        backbone.GRAD_DEFER = None
        while self.next >= 0:
            self._issue(self.next)
            self.next -= 1
        for w in self.works:
            w.wait()                                             # nccl: the current stream waits for the collective
        if self.cuda and self.active:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        self.works = []
        return self.factor


def allreduce_gradients(params, bucket_bytes=64 << 20, average=True, weight=None):
    """Average (or sum) the .grad of `params` over all ranks with a few large flattened all-reduces.

    weight: optional per-rank scalar (e.g. the number of selected tubes on this rank).  When given, the
    result is sum_r(weight_r * grad_r) / sum_r(weight_r) -- the single-process value of a loss that is a
    mean over ALL ranks' samples (the reference's masked means are per batch, two_branch.py:312,333)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    params = [p for p in params if p.requires_grad]
    dev = params[0].device if params else torch.device("cpu")
    scale = None
    if weight is not None:
        wsum = torch.tensor([float(weight)], device=dev, dtype=torch.float32)
        dist.all_reduce(wsum)
        scale = float(weight) / float(wsum.item())
    nb = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, nb
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        if scale is not None:
            flat.mul_(scale)
        dist.all_reduce(flat)
        if scale is None and average:
            flat.div_(world)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        nb += 1
        bucket, size = [], 0

    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        g = p.grad
        if bucket and (g.dtype != bucket[0].dtype or size + g.numel() * g.element_size() > bucket_bytes):
            flush()
        bucket.append(g)
        size += g.numel() * g.element_size()
    flush()
    return nb


def timed_steps(step_fn, steps, sync=None):
    """The bench contract: barrier + device sync, `steps` calls, device sync + barrier, MAX over ranks."""
    sync = sync or (torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None))
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if dist.is_initialized():
        dist.barrier()
    sync()
    el = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el
