"""step_amd/optim.py -- FlatAdam: the optimizer of the training step.

The reference builds `optim.Adam(params, lr=args.det_lr)` (train.py:126) over the single-tensor parameter groups of
utils/solver.py:12-93 (each with its own lr / weight_decay) and calls `optimizer.step()` once per iteration
(train.py:348); its schedulers (solver.py:96-180) rewrite `group['lr']` between steps.  Same constructor, same
`param_groups` / `zero_grad` / `step` / `state_dict` surface here, but MI355X-first underneath:

* every trainable parameter is re-homed into ONE flat fp32 arena (`p.data` becomes a view of it), its gradient into
  a second arena (`p.grad` is a view; autograd accumulates in place), the two Adam moments into two more;
* `step()` is ONE launch of `step_adam_flat` (include/step_amd.h) over the arenas -- pure HBM streaming, 28 B/element;
* the gradient arena is one contiguous buffer, so the data-parallel exchange is a single large RCCL all-reduce
  (`step_amd.dist.allreduce_flat`, no bucket copies: xGMI rings are per-link bound, few large messages win), and the
  1/world_size of the average and the gradient clear are folded into the Adam pass (`grad_scale`, `zero_grad`).
"""
import torch

from . import _capi, _lib

_ALIGN = 64          # elements (256 B): every tensor starts on its own cache line; segment ends stay multiples of 4


class LossScaler:
    """Dynamic loss scaling of mixed-precision (fp16) training -- what `amp.initialize(..., opt_level="O1")` + `with amp.scale_loss(loss,
    optimizer) as scaled_loss: scaled_loss.backward()` do in the reference (train.py:136-139,342-345) -- with the whole state on the device:
    `state` = {scale, growth_tracker, found_inf, -}.  Defaults are apex's DynamicLossScaler (2^16, x2 after 2000 clean steps, /2 on overflow),
    the same rule as torch.amp.GradScaler.  Use:

        scaler = LossScaler(device)
        scaler.scale_loss(loss).backward()          # a device multiply: nothing here reads the scale on the host
        opt.step(scaler=scaler, zero_grad=True)     # overflow scan + unscale + skip-or-step + scale update: step_adam_flat_amp

    The optimizer must be FlatAdam(capturable=True): a skipped step must not advance the step count, and only the device knows."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], dtype=torch.float32, device=device)

    def scale_loss(self, loss):
        return loss * self.state[0].to(loss.dtype)

    @property
    def scale(self):
        return float(self.state[0].item())

    def state_dict(self):
        s = self.state.tolist()
        return {"scale": s[0], "growth_tracker": int(s[1]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor, self.growth_interval = float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"])
        self.state.copy_(torch.tensor([float(sd["scale"]), float(sd["growth_tracker"]), 0.0, 0.0]))


class FlatAdam(torch.optim.Optimizer):
    """A torch.optim.Optimizer (the reference's schedulers subclass torch's _LRScheduler, which insists on one:
    utils/solver.py:96,141) whose whole state lives in four flat arenas."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        # the base class normalises `params` into self.param_groups (fills lr / betas / eps / weight_decay defaults,
        # rejects duplicates) exactly as it does for torch.optim.Adam
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        b, e = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        if any(tuple(g["betas"]) != tuple(b) or g["eps"] != e for g in self.param_groups):
            raise ValueError("FlatAdam: betas / eps must be the same for every group (one launch)")
        self._entries = []                                       # (group index, parameter, offset, numel)
        off = 0
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                if p.dtype != torch.float32:
                    raise RuntimeError("FlatAdam: fp32 master parameters expected, got %s" % p.dtype)
                self._entries.append((gi, p, off, p.numel()))
                off += -(-p.numel() // _ALIGN) * _ALIGN
        if not self._entries:
            raise ValueError("FlatAdam: no trainable parameter")
        if len(self._entries) > 4096:
            raise RuntimeError("FlatAdam: more than 4096 tensors")
        dev = self._entries[0][1].device
        if any(p.device != dev for _, p, _, _ in self._entries):
            raise RuntimeError("FlatAdam: all parameters must live on one device (one process per GPU)")
        _lib.dptr(self._entries[0][1].data)                      # refuses non-device tensors: there is no CPU fallback
        self.device, self.numel = dev, off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for _, p, o, n in self._entries:
                self.flat_param[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                if p.grad is not None:
                    self.flat_grad[o:o + n].copy_(p.grad.reshape(-1))
                p.grad = self.flat_grad[o:o + n].view(p.shape)
        ends = [o + -(-n // _ALIGN) * _ALIGN for _, _, o, n in self._entries]
        self._seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
        self._seg_lr = torch.zeros(len(ends), dtype=torch.float32, device=dev)
        self._seg_wd = torch.zeros(len(ends), dtype=torch.float32, device=dev)
        self._tables = None
        # capturable (as torch.optim.Adam's flag): the step counter lives on the device and step() is free of host scalars, so a
        # whole training step can be captured in a HIP graph and replayed (step_adam_flat_dev)
        self.capturable = bool(capturable)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self._bias_corr = torch.zeros(2, dtype=torch.float32, device=dev)
        self._step_host = 0

    @property
    def step_count(self):
        return int(self._step_dev.item()) if self.capturable else self._step_host

    @step_count.setter
    def step_count(self, v):
        self._step_host = int(v)
        self._step_dev.fill_(int(v))

    # -- torch.optim.Optimizer surface ---------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """optimizer.zero_grad() (train.py:287).  Gradients stay views of the arena (set_to_none is ignored)."""
        self.flat_grad.zero_()
        for _, p, o, n in self._entries:
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def _gather_stray_grads(self):
        # a caller that replaced p.grad (set_to_none, grad = tensor): fold it back into the arena.
        # Known divergence from torch.optim.Adam (documented, not emulated): torch SKIPS a parameter whose .grad is None -- no
        # moment decay, no step count -- while the one launch here treats it as a zero gradient at the shared step count (its
        # moments decay and the parameter keeps moving along exp_avg).  The reference's loop never produces that case: every
        # parameter of get_params() receives a gradient in every iteration (train.py:318-348), and zero_grad() here keeps the
        # gradients as views of the arena instead of dropping them.
        base = self.flat_grad.data_ptr()
        for _, p, o, n in self._entries:
            g = p.grad
            if g is None:
                self.flat_grad[o:o + n].zero_()
                p.grad = self.flat_grad[o:o + n].view(p.shape)
            elif g.data_ptr() != base + 4 * o:
                self.flat_grad[o:o + n].copy_(g.reshape(-1))
                p.grad = self.flat_grad[o:o + n].view(p.shape)

    def _refresh_tables(self):
        lr = [float(self.param_groups[gi]["lr"]) for gi, _, _, _ in self._entries]
        wd = [float(self.param_groups[gi]["weight_decay"]) for gi, _, _, _ in self._entries]
        if self._tables != (lr, wd):                             # the schedulers rewrite group['lr'] every iteration
            self._seg_lr.copy_(torch.tensor(lr, dtype=torch.float32))
            self._seg_wd.copy_(torch.tensor(wd, dtype=torch.float32))
            self._tables = (lr, wd)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, zero_grad=False, scaler=None):
        """optimizer.step() (train.py:348): one kernel launch.  grad_scale multiplies every gradient on the way in
        (1/world_size after a SUM all-reduce, 1/loss_scale); zero_grad=True clears the gradient arena in the same pass.
        scaler = a LossScaler whose scale the loss was multiplied by: the gradients are scanned for inf / nan, unscaled, and the
        step is skipped on overflow (apex O1 / GradScaler semantics), all on the device (step_adam_flat_amp)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from .backbone import wgrad_sync
        wgrad_sync()                                             # weight gradients still in flight on the side stream
        self._gather_stray_grads()
        self._refresh_tables()
        g0 = self.param_groups[0]
        L = _lib.lib()
        if scaler is not None:
            if not self.capturable:
                raise RuntimeError("FlatAdam.step(scaler=...): build the optimizer with capturable=True (a skipped step must not count, "
                                   "and only the device knows whether it was skipped)")
            _capi.check(L.step_adam_flat_amp(_lib.dptr(self.flat_param), _lib.dptr(self.flat_grad), _lib.dptr(self.exp_avg),
                                             _lib.dptr(self.exp_avg_sq), self.numel, _lib.dptr(self._seg_end), _lib.dptr(self._seg_lr),
                                             _lib.dptr(self._seg_wd), len(self._entries), float(g0["betas"][0]), float(g0["betas"][1]),
                                             float(g0["eps"]), _lib.dptr(self._step_dev), _lib.dptr(self._bias_corr), float(grad_scale),
                                             int(bool(zero_grad)), _lib.dptr(scaler.state), scaler.growth_factor, scaler.backoff_factor,
                                             scaler.growth_interval, _lib.stream_ptr(self.device)), "step_adam_flat_amp")
        elif self.capturable:
            _capi.check(L.step_adam_flat_dev(_lib.dptr(self.flat_param), _lib.dptr(self.flat_grad), _lib.dptr(self.exp_avg),
                                             _lib.dptr(self.exp_avg_sq), self.numel, _lib.dptr(self._seg_end), _lib.dptr(self._seg_lr),
                                             _lib.dptr(self._seg_wd), len(self._entries), float(g0["betas"][0]), float(g0["betas"][1]),
                                             float(g0["eps"]), _lib.dptr(self._step_dev), _lib.dptr(self._bias_corr), float(grad_scale),
                                             int(bool(zero_grad)), _lib.stream_ptr(self.device)), "step_adam_flat_dev")
        else:
            self._step_host += 1
            _capi.check(L.step_adam_flat(_lib.dptr(self.flat_param), _lib.dptr(self.flat_grad), _lib.dptr(self.exp_avg),
                                         _lib.dptr(self.exp_avg_sq), self.numel, _lib.dptr(self._seg_end), _lib.dptr(self._seg_lr),
                                         _lib.dptr(self._seg_wd), len(self._entries), float(g0["betas"][0]), float(g0["betas"][1]),
                                         float(g0["eps"]), self._step_host, float(grad_scale), int(bool(zero_grad)),
                                         _lib.stream_ptr(self.device)), "step_adam_flat")
        # the kernel wrote through raw pointers: bump the autograd version counters (the packed-weight caches of
        # backbone.py / heads.py are keyed on them)
        torch.autograd.graph.increment_version([p for _, p, _, _ in self._entries])
        return loss

    def state_dict(self):
        """Same structure as torch.optim.Adam.state_dict() (checkpoints: train.py:382,437)."""
        state, idx = {}, {}
        k = 0
        packed_groups = []
        for g in self.param_groups:
            ids = []
            for p in g["params"]:
                idx[id(p)] = k
                ids.append(k)
                k += 1
            pg = {kk: vv for kk, vv in g.items() if kk != "params"}
            pg["params"] = ids
            packed_groups.append(pg)
        if self.step_count:
            for _, p, o, n in self._entries:
                state[idx[id(p)]] = {"step": torch.tensor(float(self.step_count)),
                                     "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                                     "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        return {"state": state, "param_groups": packed_groups}

    def load_state_dict(self, sd):
        """optimizer.load_state_dict(checkpoint['optimizer']) (train.py:205); accepts torch.optim.Adam's own dicts."""
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(groups, self.param_groups)):
            raise ValueError("FlatAdam.load_state_dict: parameter groups do not match")
        idx = {}
        for a, b in zip(groups, self.param_groups):
            for k, p in zip(a["params"], b["params"]):
                idx[id(p)] = k
            for kk, vv in a.items():
                if kk != "params":
                    b[kk] = vv
        steps = set()
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for _, p, o, n in self._entries:
                st = sd["state"].get(idx[id(p)])
                if st is None:
                    steps.add(0)
                    continue
                steps.add(int(float(st["step"])))
                self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        if len(steps) != 1:
            raise ValueError("FlatAdam.load_state_dict: parameters are at different step counts %s" % sorted(steps))
        self.step_count = steps.pop()
        self._tables = None
