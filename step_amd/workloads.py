"""step_amd/workloads.py -- synthetic BASELINE workloads C3 (full two_branch inference) and C4 (one training step) built
from the product modules; used by bench.py --config c3|c4 and tools/c3_bench.py.  Random-init weights of the real
architectures, synthetic clips and the reference's default anchor grid (there are no datasets or checkpoints offline)."""
from types import SimpleNamespace as NS

import numpy as np
import torch

from . import BaseNet, ContextNet, ROINet, TwoBranchNet
from . import dist as sdist
from .driver import GraphedInference, inference, inference_flat, postprocess
from .backbone import wgrad_into_grad
from .optim import FlatAdam
from .selection import train_select
from .driver import _flat_tubes
from .tube_math import generate_anchors


def step_cfg(**kw):
    """The attributes of the reference's argparse namespace the modules read (config.py; scripts/train_step.sh)."""
    base = dict(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False, T=3, num_classes=60,
                fc_dim=256, dropout=0.0, pool_size=7, no_context=False, max_iter=3, NUM_CHUNKS={1: 1, 2: 1, 3: 3, 4: 3},
                temporal_mode="predict", image_size=(400, 400), pool_mode="align",
                # training sample selection (config.py:64-76 with the values of scripts/train_step.sh:43-48)
                topk=-1, cls_thresh=[0.2, 0.35, 0.5], reg_thresh=[0.2, 0.35, 0.5], max_pos_num=5, neg_ratio=2,
                selection_sampling="softmax", lambda_reg=5.0, lambda_neighbor=1.0)
    base.update(kw)
    return NS(**base)


def build_nets(dev, seed=123, heads=3):
    args = step_cfg()
    torch.manual_seed(seed)                                       # config.py:38 man_seed
    base = BaseNet(args).to(dev).eval()
    ctx = ContextNet(args).to(dev).eval()
    nets = {"roi_net": ROINet("align", 7)}
    for i in range(heads):
        d = TwoBranchNet(args).to(dev).eval()
        d.set_device(dev)
        nets["det_net%d" % i] = d
    with torch.no_grad():                                         # keep box deltas small, like a trained regressor
        for i in range(heads):
            for nme in ("local_reg", "neighbor_reg1", "neighbor_reg2"):
                getattr(nets["det_net%d" % i], nme).weight.mul_(0.05)
    return args, base, ctx, nets


class C3Inference:
    """B clips [36,3,400,400], `tubes` initial tubes per clip, 3 refinement steps, batched per-class NMS (test.py:140-218)."""

    def __init__(self, dev, dtype, batch=4, tubes=11, seed=123, graph=True, share=None):
        # share = another C3Inference: the same networks (weights, packed-weight caches) on other clips -- a second batch in flight
        if share is not None:
            self.args, self.base, self.ctx, self.nets = share.args, share.base, share.ctx, share.nets
        else:
            self.args, self.base, self.ctx, self.nets = build_nets(dev, seed)
        g = torch.Generator().manual_seed(seed + (1000 if share is not None else 0))
        self.x = (torch.rand(batch, 36, 3, 400, 400, generator=g) * 2 - 1).to(dev).to(dtype)
        anchors = generate_anchors()[:tubes] * 400.0
        self.tubes = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32) for _ in range(batch)]
        self.batch = batch
        self.graphed = GraphedInference(self.args, self.base, self.ctx, self.nets, self.x, self.tubes) if graph else None
        if self.graphed is not None:
            self.x = self.graphed.images                      # the clips are RESIDENT in the captured step's input buffer (as bench.py's C2 loop has them)

    def eager(self):
        with torch.no_grad():
            cf = self.base(self.x)
            cx = self.ctx(cf)
            hist, _ = inference(self.args, cf, cx, self.nets, 3, self.tubes)
            return postprocess(self.args, hist)

    def step(self):
        if self.graphed is None:
            return self.eager()
        return self.finish(self.launch())

    def launch(self):
        """Enqueue the captured backbone + context + three refinement steps on the CURRENT stream; returns the history (static buffers
        the next launch() of this object overwrites).  launch() / finish() apart let a caller keep a second batch in flight."""
        with torch.no_grad():
            hist, _, _ = self.graphed(self.x)
        return hist

    def finish(self, hist):
        """Post-processing of a launched batch (thresholds, per-class NMS, the one device-to-host copy): synchronises the current stream only."""
        with torch.no_grad():
            return postprocess(self.args, hist)


def ava_clips(seed, batch):
    """`batch` synthetic AVA-shaped clips [36,3,400,400] in U(-1,1) (what ConvertFromInts(scale=2) yields), a pure function of
    the seed: rank r of a data-parallel job draws ava_clips(123 + r, ...)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(batch, 36, 3, 400, 400, generator=g) * 2 - 1


class C4TrainStep:
    """One optimisation step on `batch` AVA-shaped clips per rank (fp32): backbone + ContextNet + the max_iter = 3 heads on
    tubes of 3, 3 and 9 frames (NUM_CHUNKS 1, 1, 3), the three losses of train.py:318-331 summed over the steps
    (lambda_reg 5, lambda_nbr 1), ONE flat gradient all-reduce over ranks (step_amd.dist.allreduce_flat), ONE fused Adam
    launch that also applies 1/world and clears the gradients (step_amd.optim.FlatAdam).  The reference's proposal
    selection between steps (utils/utils.py:135-423, host Python) is not part of the hot path: every step trains on the
    same `tubes_per_clip` anchor tubes, extended to the step's length."""

    def __init__(self, dev, batch=1, tubes_per_clip=5, seed=123, max_iter=3, dtype=torch.float32, capturable=False, force_exchange=False):
        # replicas: the same weights on every rank (same init seed, then rank 0's copy is broadcast once, as DDP does);
        # `seed` only varies the rank's clips
        self.args, self.base, self.ctx, self.nets = build_nets(dev, 123, heads=max_iter)
        self.heads = [self.nets["det_net%d" % i] for i in range(max_iter)]
        self.mods = [self.base, self.ctx] + self.heads
        sdist.broadcast_parameters(self.mods)
        for m in self.mods:
            m.train()
        self.params = [p for m in self.mods for p in m.parameters() if p.requires_grad]
        self.opt = FlatAdam(self.params, lr=1e-5, capturable=capturable)
        self.graph = None
        self.graph_mode = None                                   # "one" | "split" after capture()
        self._g_update = None
        # the gradient exchange runs bucket by bucket on a communication stream WHILE backward is still producing the
        # earlier layers' gradients (step_amd.dist.BucketedReducer); single-process runs issue nothing
        # (force_exchange: in a ONE-rank process group the collectives are issued all the same -- the multi-rank program on a one-GPU box)
        self.reducer = sdist.BucketedReducer(self.opt, single_rank=bool(force_exchange))
        # fp32 master weights either way; a 16-bit clip makes every activation / data gradient 16-bit (fp32 accumulate),
        # weight gradients stay fp32
        self.x = ava_clips(seed, batch).to(dev).to(dtype)
        anchors = torch.from_numpy(generate_anchors()[:tubes_per_clip] * 400.0).to(dev)                  # [K,4]
        K = tubes_per_clip
        self.steps = []
        for it in range(max_iter):
            Tl = 3 * self.args.NUM_CHUNKS[it + 1]
            t0 = (9 - Tl) // 2                                     # centred window of the 9 feature frames (utils.py:41-43)
            flats = []
            for b in range(batch):
                fr = (b * 9 + t0 + torch.arange(Tl, device=dev, dtype=torch.float32)).view(1, Tl, 1).expand(K, Tl, 1)
                flats.append(torch.cat([fr, anchors.view(K, 1, 4).expand(K, Tl, 4)], 2))
            self.steps.append((Tl, t0, torch.cat(flats, 0).contiguous()))
        t = torch.zeros(batch * K, 3, 66, device=dev)
        t[:, :, :4] = anchors.repeat(batch, 1).view(batch * K, 1, 4) + 4.0
        t[:, :, 4] = 1
        t[:, :, 5] = 1
        t[:, :, 6 + 7] = 1
        self.targets = t
        self.clip_of = torch.arange(batch, device=dev).repeat_interleave(K)
        self.batch, self.K = batch, K
        self.loss = None

    @staticmethod
    def _ctx_per_tube(cx, k):
        """The clip's context feature for each of its k tubes (utils.py:55-57 indexes it per tube): with the SAME number of tubes per clip this
        is a broadcast, whose backward is a fixed-order sum over the k copies -- `cx[clip_of]` is an index_select whose backward is an
        atomic index_add (24 us x 6 per step in the round-6 trace, and the one place of the step whose summation order the hardware picks)."""
        B = cx.shape[0]
        return cx.unsqueeze(1).expand(B, k, *cx.shape[1:]).reshape(B * k, *cx.shape[1:])

    def forward_backward(self, exchange=False):
        """Losses of the three steps and their gradients (into FlatAdam's gradient arena); no update.  exchange=True overlaps
        the bucketed gradient all-reduce with the backward pass and leaves the averaging factor in self.scale."""
        cf = self.base(self.x)                                    # [B,9,832,25,25]
        cx = self.ctx(cf)                                         # [B,1024,9,1,1]
        loss = 0.0
        for head, (Tl, t0, flat) in zip(self.heads, self.steps):
            pooled = self.nets["roi_net"](cf, flat)               # the frame-index column addresses frame b*9 + t of cf
            pooled = pooled.reshape(self.batch * self.K, Tl, *pooled.shape[1:])
            o = head(pooled, context_feat=self._ctx_per_tube(cx, self.K)[:, :, t0:t0 + Tl], tubes=flat, targets=self.targets)
            loss = loss + o[4].mean() + 5 * o[5].mean() + o[6].mean()
        if exchange:
            self.reducer.begin()
        with wgrad_into_grad():                                   # weight gradients go straight into FlatAdam's arena
            loss.backward()
        self.scale = self.reducer.finish() if exchange else 1.0
        return loss

    def _eager_step(self):
        loss = self.forward_backward(exchange=True)
        self.opt.step(grad_scale=self.scale, zero_grad=True)     # gradients are clean for the next backward
        self.loss = loss.detach()
        return self.loss

    def step(self):
        if self.graph is not None:
            self.opt._refresh_tables()                           # lr / weight_decay of param_groups -> the device tables the captured Adam reads (schedulers keep working)
            self.graph.replay()                                  # ~800 launches (+ the bucket all-reduces of a multi-rank step), one submission
            if self._g_update is not None:                       # "split" form: forward / backward replayed, the exchange eager, the update replayed
                sdist.allreduce_flat(self.opt.flat_grad)
                self._g_update.replay()
            # the replay re-packed the weights at its start and Adam changed them at its end: any eager use of the modules between
            # replays (validation) must see its packed-weight caches as stale
            torch.autograd.graph.increment_version(self.params)
            return self.loss
        return self._eager_step()

    def _warm(self, warmup):
        dev = self.x.device
        cur = torch.cuda.current_stream(dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager_step()
        cur.wait_stream(s)
        torch.cuda.synchronize(dev)

    def capture(self, warmup=3, mode="auto"):
        """Capture the WHOLE step (forward, backward with the side-stream weight gradients, the gradient exchange, weight re-pack, Adam)
        in one HIP graph; step() then replays it.  The shapes of this step are static (fixed tubes per clip), every scalar that
        changes between steps lives on the device (FlatAdam(capturable=True): step counter), and nothing on the path synchronises with
        the host -- with 16-bit activations the eager step is bound by the host issuing ~800 launches, not by the GPU.

        With a process group (the reference's step is one optimizer.step() per iteration over all devices, train.py:142-148,257-348) the
        N > 1 step is the SAME program as the N = 1 step: mode "one" records the bucket all-reduces on the communication stream
        inside the graph (RCCL collectives are stream-ordered and capturable; ProcessGroupNCCL issues them on its own stream, forked
        from and joined to the capturing stream by events, and keeps captured work off its watchdog) -- the overlap of the exchange with
        backward is part of the replayed graph.  Mode "split" is the fallback where the collectives cannot be recorded (a gloo group;
        an RCCL build that refuses capture): graph 1 = forward + backward into the gradient arena, the exchange eager as ONE flat
        all-reduce, graph 2 = re-pack + Adam.  "auto" = "one" on nccl, falling back to "split" if the capture raises.
        Runs `warmup` eager steps first (caches, pack tables, workspaces, the communicator); the captured step itself is recorded,
        not executed."""
        if not self.opt.capturable:
            raise RuntimeError("C4TrainStep.capture: build the workload with capturable=True (device-side Adam step counter)")
        import gc
        dd = torch.distributed
        grouped = self.reducer.active
        backend = dd.get_backend() if (dd.is_available() and dd.is_initialized()) else None
        if mode == "auto":
            # with a process group: "split" -- nothing of the group is recorded, so its watchdog thread cannot collide with the capture
            # (the "one" form's guard is a pause, and the failure is a process abort no fallback can catch).  "one" with
            # collectives stays available on request; it replays bit-identically (tests/test_gpu_ddp.py) and measured 13.55 ms against
            # 13.47 ms split on the one-GPU box -- nothing is lost by the safe default until the form has run on real multi-GPU hardware.
            mode = "one" if not grouped else "split"
        if mode not in ("one", "split"):
            raise ValueError("C4TrainStep.capture: mode is 'auto', 'one' or 'split'")
        if mode == "one" and grouped and backend != "nccl":
            raise RuntimeError("C4TrainStep.capture(mode='one'): only RCCL ('nccl') collectives can be recorded in a HIP graph; use mode='split'")
        gc.collect()                                             # (torch.cuda.graph collects too: dead nets must not drop out of the re-pack table mid-capture)
        # at least TWO eager steps: the second one is the first that re-packs every weight in one launch (backbone._repack_all) and builds
        # that launch's device tables with host -> device copies, which a capture cannot record (seen in round 6 with warmup = 1:
        # hipErrorStreamCaptureUnsupported in both capture modes)
        self._warm(max(int(warmup), 2))
        dev = self.x.device
        if grouped and backend == "nccl" and mode == "one":
            # OPT-IN form.  The process group's watchdog thread polls the completion events of the collectives the warm-up steps issued
            # (every ~100 ms); if it still holds some when the capture begins it queries them while the group's internal stream is
            # capturing, HIP answers hipErrorCapturedEvent and the watchdog ABORTS the process (no exception to catch).  Round 5 guarded
            # this with a pause (1 abort in 8 runs without, 0 in 6 with); round 6 tried to replace the pause by a deterministic drain read
            # from the group's flight recorder -- the recorder's own event queries made it worse (abort inside a 30-capture loop,
            # gpurun_out of call c2) -- and no other API exposes the watchdog's list.  So the pause stays for callers who ask for "one"
            # explicitly, and "auto" no longer picks this form with a live process group (see above).
            import os
            import time
            torch.cuda.synchronize(dev)
            time.sleep(float(os.environ.get("STEP_PG_DRAIN_S", "1.0")))
        # with a live process group its watchdog / heartbeat threads may touch the runtime while this thread records: only THIS thread's
        # calls are checked against the capture
        kw = {"capture_error_mode": "thread_local"} if grouped else {}
        if mode == "one":
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **kw):
                    self._eager_step()
                self.graph, self.graph_mode, self._g_update = g, "one", None
            except RuntimeError as e:
                if not grouped:
                    raise
                import warnings
                warnings.warn("C4TrainStep.capture: recording the gradient exchange failed (%s); falling back to the split form" % (str(e).splitlines()[0],))
                torch.cuda.synchronize(dev)
                self.reducer._armed = False
                from . import backbone as _bb
                _bb.GRAD_READY = _bb.GRAD_DEFER = None
                _bb.wgrad_sync()
                self.opt.zero_grad()
                mode = "split"
        if mode == "split":
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, **kw):
                loss = self.forward_backward(exchange=False)
                self.loss = loss.detach()
            world = dd.get_world_size() if (dd.is_available() and dd.is_initialized()) else 1
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, **kw):
                self.opt.step(grad_scale=1.0 / world, zero_grad=True)
            self.graph, self._g_update, self.graph_mode = g1, g2, "split"
        # host-side caches now carry the version stamps of a step whose kernels only run on replay: make them stale again for
        # any eager use of the modules after this point
        torch.autograd.graph.increment_version(self.params)
        return self


class C4SelectTrainStep(C4TrainStep):
    """The reference's whole training iteration (train.py:257-348), proposal selection included: backbone + ContextNet with
    gradients; an eval-mode, no-grad inference() over the first max_iter-1 steps to get the refined tubes; then for every
    step train_select() (step_amd.selection: the reference's sampling, same RNG streams) picks positives / negatives among
    them, ROIAlign pools the selected tubes, the step's head returns the three losses; one backward, one flat gradient
    all-reduce, one fused Adam launch.  `targets`: 2 ground-truth tubes per clip with 3 positive classes each.

    Three forms of the same iteration:
      step()          eager, ragged: every head sees exactly the selected tubes (shapes change from iteration to iteration)
      step_padded()   eager, STATIC shapes: every clip's selection is padded to `budget` slots (max_pos_num x (1 + neg_ratio) = 15); padded
                      slots carry a copy of the clip's first initial tube, all-zero targets and row weight 0 -- they receive exactly zero loss
                      and zero gradient (the classification loss is the masked mean over the real rows, the regression losses are masked by
                      their flags already)
      capture()       the padded form replayed as HIP graphs: graph F = backbone + ContextNet forward (with gradients)
                      + the no-grad inference; HOST: train_select on the inference's static outputs (one launch + one small copy per step,
                      then the draws from `random` / `numpy.random` in the reference's order) writes the selection into static device
                      buffers; graph B = ROIAlign + the three heads + losses + backward + re-pack + Adam (with a process group: B stops after
                      backward, ONE eager flat all-reduce, graph U = re-pack + Adam -- the split form, nothing of the group is recorded).
                      Same kernels on the same buffers as step_padded(): bit-identical trajectory (tests/test_gpu_graph_step.py)."""

    def __init__(self, dev, batch=1, seed=123, dtype=torch.float32, tubes_per_clip=34, capturable=False, force_exchange=False, budget=None):
        super().__init__(dev, batch=batch, tubes_per_clip=5, seed=seed, max_iter=3, dtype=dtype, capturable=capturable, force_exchange=force_exchange)
        rs = np.random.RandomState(seed)
        anchors = (generate_anchors()[:tubes_per_clip] * 400.0).astype(np.float32)
        self.init_tubes = [np.tile(anchors[:, None, :], (1, 3, 1)) for _ in range(batch)]
        self.gt = []
        for _ in range(batch):
            t = np.zeros((2, 3, 4 + 60), np.float32)
            for g_ in range(2):
                box = anchors[rs.randint(0, len(anchors))] + rs.uniform(-20, 20, 4).astype(np.float32)
                for c in range(3):
                    t[g_, c, :4] = box + rs.uniform(-5, 5, 4).astype(np.float32)
                t[g_, :, 4 + rs.randint(0, 60, 3)] = 1
            self.gt.append(t)
        self.selected = []
        a = self.args
        self.budget = int(budget or a.max_pos_num * (1 + a.neg_ratio))
        # static state of the padded form
        K = batch * self.budget
        self.flat0, self.nums0 = _flat_tubes(self.init_tubes, dev)
        self.clip_of0 = torch.as_tensor(np.repeat(np.arange(batch), self.nums0), device=dev)
        self.clip_of_pad = torch.arange(batch, device=dev).repeat_interleave(self.budget)
        self.s_flat, self.s_tgt, self.s_mask, self.s_inv = [], [], [], []
        self.h_flat, self.h_tgt, self.h_mask, self.h_inv = [], [], [], []
        for i in range(1, a.max_iter + 1):
            Tl = a.NUM_CHUNKS[i] * a.T
            for dst, host, shape in ((self.s_flat, self.h_flat, (K, Tl, 5)), (self.s_tgt, self.h_tgt, (K, 3, 6 + a.num_classes)),
                                     (self.s_mask, self.h_mask, (K, 1)), (self.s_inv, self.h_inv, (1,))):
                dst.append(torch.zeros(shape, device=dev))
                host.append(torch.zeros(shape).pin_memory())
        self._gF = self._gB = self._gU = None
        self._front = None

    # ---- the ragged eager iteration (the reference's program, shapes follow the selection)
    def step(self):
        if self._gF is not None:
            return self._replay()
        a = self.args
        cf = self.base(self.x)
        cx = self.ctx(cf)
        for m in self.mods:
            m.eval()
        with torch.no_grad():
            history, _ = inference(a, cf.detach(), cx.detach(), self.nets, a.max_iter - 1, self.init_tubes)
        for m in self.mods:
            m.train()
        loss = 0.0
        self.selected = []
        for i in range(1, a.max_iter + 1):
            chunks, max_chunks = a.NUM_CHUNKS[i], a.NUM_CHUNKS[a.max_iter]
            t0 = int((max_chunks - chunks) / 2) * a.T
            Tl = chunks * a.T
            sel, tgt = train_select(i, history[i - 2] if i > 1 else None, self.gt, self.init_tubes, a)
            self.selected.append([len(s_) for s_ in sel])
            flat, nums = _flat_tubes(sel, cf.device)
            targets = torch.from_numpy(np.concatenate(tgt, axis=0)).to(cf.device)
            clip_of = torch.as_tensor(np.repeat(np.arange(len(nums)), nums), device=cf.device)
            pooled = self.nets["roi_net"](cf[:, t0:t0 + Tl], flat)
            pooled = pooled.reshape(flat.shape[0], Tl, *pooled.shape[1:])
            o = self.heads[i - 1](pooled, context_feat=cx[clip_of][:, :, t0:t0 + Tl], tubes=flat, targets=targets)
            loss = loss + o[4].mean() + a.lambda_reg * o[5].mean() + a.lambda_neighbor * o[6].mean()
        self.reducer.begin()
        with wgrad_into_grad():
            loss.backward()
        scale = self.reducer.finish()
        self.opt.step(grad_scale=scale, zero_grad=True)
        self.loss = loss.detach()
        return self.loss

    # ---- the padded iteration in its three parts
    def _front_part(self):
        """backbone + ContextNet (with gradients) and the no-grad, eval-mode inference over the first max_iter - 1 steps (train.py:266-272)"""
        a = self.args
        cf = self.base(self.x)
        cx = self.ctx(cf)
        for m in self.mods:
            m.eval()
        with torch.no_grad():
            history, _ = inference_flat(a, cf.detach(), cx.detach(), self.nets, a.max_iter - 1, self.flat0, self.nums0, self.clip_of0, want_classes=False)
        for m in self.mods:
            m.train()
        return cf, cx, history

    def _select_part(self, history):
        """HOST: train_select per step on the inference's outputs, padded to `budget` slots per clip, into the static device buffers"""
        a = self.args
        self.selected = []
        B, Bu = self.batch, self.budget
        for i in range(1, a.max_iter + 1):
            Tl = a.NUM_CHUNKS[i] * a.T
            sel, tgt = train_select(i, history[i - 2] if i > 1 else None, self.gt, self.init_tubes, a)
            self.selected.append([len(s_) for s_ in sel])
            hf, ht, hm = self.h_flat[i - 1].numpy(), self.h_tgt[i - 1].numpy(), self.h_mask[i - 1].numpy()
            ht[...] = 0
            hm[...] = 0
            n_real = 0
            for b in range(B):
                n = len(sel[b])
                if n > Bu:
                    raise RuntimeError("C4SelectTrainStep: %d tubes selected for one clip, budget %d" % (n, Bu))
                rows = slice(b * Bu, b * Bu + n)
                pad = slice(b * Bu + n, (b + 1) * Bu)
                hf[rows, :, 1:] = sel[b]
                fill = np.asarray(self.init_tubes[b][0], np.float32)                  # a valid box for the padded slots' ROIAlign
                if fill.shape[0] != Tl:                                               # (the last step's tubes are 3 chunks long)
                    fill = np.tile(fill, (Tl // fill.shape[0] + 1, 1))[:Tl]
                hf[pad, :, 1:] = fill
                hf[b * Bu:(b + 1) * Bu, :, 0] = b * Tl + np.arange(Tl, dtype=np.float32)   # frame index inside cf[:, t0:t0+Tl] flattened over clips
                ht[rows] = tgt[b]
                hm[rows] = 1
                n_real += n
            self.h_inv[i - 1][0] = 1.0 / (max(n_real, 1) * a.num_classes)
            self.s_flat[i - 1].copy_(self.h_flat[i - 1], non_blocking=True)
            self.s_tgt[i - 1].copy_(self.h_tgt[i - 1], non_blocking=True)
            self.s_mask[i - 1].copy_(self.h_mask[i - 1], non_blocking=True)
            self.s_inv[i - 1].copy_(self.h_inv[i - 1], non_blocking=True)

    def _loss_part(self, cf, cx):
        a = self.args
        loss = 0.0
        K = self.batch * self.budget
        for i in range(1, a.max_iter + 1):
            chunks, max_chunks = a.NUM_CHUNKS[i], a.NUM_CHUNKS[a.max_iter]
            t0 = int((max_chunks - chunks) / 2) * a.T
            Tl = chunks * a.T
            flat = self.s_flat[i - 1]
            pooled = self.nets["roi_net"](cf[:, t0:t0 + Tl], flat)
            pooled = pooled.reshape(K, Tl, *pooled.shape[1:])
            o = self.heads[i - 1](pooled, context_feat=self._ctx_per_tube(cx, self.budget)[:, :, t0:t0 + Tl], tubes=flat, targets=self.s_tgt[i - 1])
            # o[4]: the element-wise classification loss [K * classes]; the reference's .mean() runs over the REAL rows only
            lcls = (o[4].view(K, -1) * self.s_mask[i - 1]).sum() * self.s_inv[i - 1][0]
            loss = loss + lcls + a.lambda_reg * o[5].mean() + a.lambda_neighbor * o[6].mean()
        return loss

    def _back_part(self, cf, cx, update=True):
        loss = self._loss_part(cf, cx)
        exchange = update                                        # (split form: the exchange is eager, between the graphs)
        if exchange:
            self.reducer.begin()
        with wgrad_into_grad():
            loss.backward()
        if update:
            scale = self.reducer.finish()
            self.opt.step(grad_scale=scale, zero_grad=True)
        self.loss = loss.detach()
        return self.loss

    def step_padded(self):
        cf, cx, hist = self._front_part()
        self._select_part(hist)
        return self._back_part(cf, cx)

    def _eager_step(self):                                       # (what bench.py instruments for the per-kernel roofline)
        return self.step_padded()

    def capture(self, warmup=2, mode="auto"):
        """See the class docstring.  With a process group the update is a graph of its own behind ONE eager flat all-reduce (the split
        form of C4TrainStep.capture: nothing of the group is recorded)."""
        if not self.opt.capturable:
            raise RuntimeError("C4SelectTrainStep.capture: build the workload with capturable=True")
        import gc
        dd = torch.distributed
        grouped = self.reducer.active
        gc.collect()
        dev = self.x.device
        cur = torch.cuda.current_stream(dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(max(int(warmup), 2)):
                self.step_padded()
        cur.wait_stream(s)
        torch.cuda.synchronize(dev)
        kw = {"capture_error_mode": "thread_local"} if grouped else {}
        gF = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gF, **kw):
            self._front = self._front_part()
        pool = gF.pool()
        cf, cx, hist = self._front
        gF.replay()                                              # (a capture records, it does not run: give the selection real predictions to read)
        torch.cuda.synchronize(dev)
        import random
        rs_py, rs_np = random.getstate(), np.random.get_state()  # this selection trains nothing: it must not consume draws of the reference's RNG streams
        self._select_part(hist)
        random.setstate(rs_py)
        np.random.set_state(rs_np)
        torch.cuda.synchronize(dev)
        gB = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gB, pool=pool, **kw):
            self._back_part(cf, cx, update=not grouped)
        gU = None
        if grouped:
            world = dd.get_world_size() if (dd.is_available() and dd.is_initialized()) else 1
            gU = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gU, pool=pool, **kw):
                self.opt.step(grad_scale=1.0 / world, zero_grad=True)
        self._gF, self._gB, self._gU = gF, gB, gU
        self.graph, self.graph_mode = gF, ("select-split" if grouped else "select")
        torch.autograd.graph.increment_version(self.params)
        return self

    def _replay(self):
        self.opt._refresh_tables()
        self._gF.replay()
        self._select_part(self._front[2])                        # host: reads the inference's static outputs (one small copy per step)
        self._gB.replay()
        if self._gU is not None:
            sdist.allreduce_flat(self.opt.flat_grad)
            self._gU.replay()
        torch.autograd.graph.increment_version(self.params)
        return self.loss
