"""step_amd/backbone.py -- the Inception-I3D building blocks and BaseNet on the gfx950 kernels.

Host-side mirror of the reference interface (same constructor arguments, forward signatures,
parameter / buffer names and state_dict keys):
    Unit3D, MaxPoolTF, Mixed           <->  models/i3dpt.py:43-163   (Unit3Dpy, MaxPool3dTFPadding, Mixed)
    BaseNet, build_base_i3d            <->  models/networks.py:50-142

The modules own ordinary nn.Parameters / buffers (inside nn.Conv3d / nn.BatchNorm3d containers so
that checkpoints, `weights_init`, utils/solver.get_params and the reference's "classname contains
BatchNorm" logic keep working), but forward() never calls torch's conv/bn/pool: it calls the HIP
kernels through step_amd.ops on CHANNELS-LAST activations ([N,T,H,W,C]).  A block's four branches
write straight into channel slices of one output buffer (no torch.cat), eval-mode BN is folded into
a per-channel scale/shift applied in the conv epilogue, TF-SAME padding is a load predicate.

Training: forward is always the HIP path.  When gradients are required the conv unit records a torch
autograd node (_ConvUnitFn) whose backward is HIP as well: the DATA gradient runs on the same conv kernels
(taps flipped, channel roles swapped), the WEIGHT gradient on step_conv_wgrad / step_stem_wgrad (fp32 MFMA over
the pixel axis), the pools on step_maxpool3d_tf_backward.  No torch / MIOpen convolution or pooling kernel is
called on either pass; torch does the element-wise mask / scale arithmetic around them.
"""
import os
import weakref

import torch
import torch.nn as nn

from . import ops

BN_EPS = 1e-5

# (models/i3dpt.py:213-231) in_channels, [b0, b1a, b1b, b2a, b2b, b3]
MIXED_CFG = {
    "3b": (192, (64, 96, 128, 16, 32, 32)), "3c": (256, (128, 128, 192, 32, 96, 64)),
    "4b": (480, (192, 96, 208, 16, 48, 64)), "4c": (512, (160, 112, 224, 24, 64, 64)),
    "4d": (512, (128, 128, 256, 24, 64, 64)), "4e": (512, (112, 144, 288, 32, 64, 64)),
    "4f": (528, (256, 160, 320, 32, 128, 128)), "5b": (832, (256, 160, 320, 32, 128, 128)),
    "5c": (832, (384, 192, 384, 48, 128, 128)),
}


def _ver(*tensors):
    return tuple((t.data_ptr(), t._version, t.device) for t in tensors if t is not None)


class _ConvUnitFn(torch.autograd.Function):
    """Autograd node around the fused HIP conv unit  y = act(conv(x, w) * scale + shift (+ res)).
    forward = the HIP kernel (always).  backward: data gradient = the HIP kernel again; weight gradient = the HIP
    wgrad kernel (step_conv_wgrad).  `w_eff` is the EFFECTIVE weight
    [Cout, Cin_eff, kd, kh, kw] (after the unit's channel slice / permutation), produced by
    differentiable view ops so autograd routes its gradient back to the parameter."""

    @staticmethod
    def forward(ctx, x, w_eff, scale, shift, res, unit, relu, raw=False):
        # raw: the conv alone, no affine / ReLU (the unit's BatchNorm runs on batch statistics as its own autograd node, _BatchNormTrainFn)
        y = unit._launch(x.detach(), relu, None if res is None else res.detach(), None, raw=raw)
        ctx.save_for_backward(x, w_eff, scale, shift, y, res)
        ctx.relu = relu
        ctx.unit = unit
        return y

    @staticmethod
    def _dgrad(gin, w_eff, dtype, k, unit=None):
        # data gradient = the SAME fused HIP conv on the output gradient with the taps flipped and the channel roles
        # swapped (stride 1, SAME padding): no torch / MIOpen kernel on this leg; the flipped / transposed weight is
        # never materialised (step_conv_pack_weight_dgrad packs it straight from w)
        vec = 16 // gin.element_size()                           # the kernels move 16-byte channel vectors
        padc = (-gin.shape[-1]) % vec
        if padc:                                                 # e.g. the 60-class / 12-column Linear layers in 16-bit
            gin = torch.nn.functional.pad(gin, (0, padc))
        wp = unit.packed_dgrad(w_eff, dtype, gin.shape[-1]) if unit is not None else ops.pack_conv_weight_dgrad(w_eff, dtype, gin.shape[-1])
        return ops.conv_forward(gin, wp, w_eff.shape[1], k, None, None, False, None, None)

    @staticmethod
    def backward(ctx, gy):
        x, w_eff, scale, shift, y, res = ctx.saved_tensors
        k = tuple(w_eff.shape[2:])
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_res = res is not None and ctx.needs_input_grad[4]
        if not (ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or (need_res and (scale is not None or res.dtype != x.dtype))):
            # frozen affine, no residual (every backbone / Inception unit): only the conv's own gradients are wanted, so the
            # cast / ReLU mask / scale / cast chain is ONE HIP pass producing the wgrad operand (fp32) and the dgrad operand
            # (a residual input without an affine -- the Bottlenecks' third conv -- receives the same masked gradient as the conv)
            # 16-bit activations: the weight gradient runs on the 16-bit matrix instructions too (step_conv_wgrad16: 16x the
            # fp32 instruction's rate), reading the SAME 16-bit activation gradient as the data-gradient conv -- what
            # mixed-precision training back-propagates; no fp32 copy of the gradient is written
            w16 = WGRAD16 and need_w and x.dtype != torch.float32 and y.dtype == x.dtype
            fused = ops.act_grad(y, gy, scale, ctx.relu, want_f32=need_w and not w16, want_act=need_x or need_res or w16) \
                if (need_x or need_w or need_res) else (None, None)
            if fused is not None:
                g32, gact = fused
                if w16:
                    g32 = gact

                    def wgrad(x_, g_, cout_, k_, into=None):
                        return ops.conv_wgrad16(x_, g_, cout_, k_, into=into)
                else:
                    wgrad = ops.conv_wgrad
                target = _wgrad_target(ctx.unit, w_eff) if (need_w and WGRAD_INTO_GRAD and x.is_cuda and ops.PROFILE is None) else None
                if target is not None:
                    # opt-in (wgrad_into_grad()): the weight gradient is ACCUMULATED straight into the parameter's .grad (the
                    # FlatAdam arena) on a side stream that nobody waits for until wgrad_sync() -- no clear, no autograd add, and the
                    # latency-bound weight-gradient launches overlap the rest of the backward pass instead of sitting in its
                    # critical path.  The fixed-order sums of the partial tiles queue up and run eight layers per launch
                    # (_PEND_Q; the parameters are announced to the gradient exchange when their sum has been launched).
                    _unit_wgrad(ctx.unit, w_eff, x, g32, k, _PEND_Q)
                    if len(_PEND_Q) >= 8:
                        _flush_wgrads(_PEND_Q, x.device)
                    if need_res and g32 is gact:
                        # the side stream is still READING this buffer while it is handed to autograd as the residual's
                        # gradient: with a use count of one autograd would accumulate the other branch's gradient into it
                        # IN PLACE on the main stream (input_buffer.cpp) -- a write/read race that perturbed the Bottleneck
                        # conv3 weight gradients by ~1 % (found by tests/test_gpu_ddp.py).  A second reference makes autograd
                        # add out of place; the list is dropped in wgrad_sync().
                        _KEEP.append(gact)
                    gx = _ConvUnitFn._dgrad(gact, w_eff, x.dtype, k, ctx.unit) if need_x else None
                    return gx, None, None, None, (gact if need_res else None), None, None, None
                if need_x and need_w and WGRAD_SIDE_STREAM and x.is_cuda and x.dtype == torch.float32 and ops.PROFILE is None:
                    # the two gradients are independent: the weight gradient (many of them latency-bound launches that fill a
                    # fraction of the chip) runs on a side stream beside the data-gradient conv.  Measured on the C4 step:
                    # fp32 69.3 -> 65.0 ms; with 16-bit activations the data-gradient convs are too short to hide anything
                    # and the extra stream bookkeeping costs 2 % (46.1 -> 47.2 ms), so 16-bit stays on one stream
                    main = torch.cuda.current_stream(x.device)
                    side = _side_streams(x.device)[0]
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        gw = wgrad(x, g32, w_eff.shape[0], k).to(w_eff.dtype)
                    gx = _ConvUnitFn._dgrad(gact, w_eff, x.dtype, k, ctx.unit)
                    main.wait_stream(side)
                    gw.record_stream(main)
                    return gx, gw, None, None, (gact if need_res else None), None, None, None
                gx = _ConvUnitFn._dgrad(gact, w_eff, x.dtype, k, ctx.unit) if need_x else None
                gw = wgrad(x, g32, w_eff.shape[0], k).to(w_eff.dtype) if need_w else None
                return gx, gw, None, None, (gact if need_res else None), None, None, None
        g = gy.float()
        if ctx.relu:
            g = g * (y > 0).to(g.dtype)
        C = g.shape[-1]
        gres = g.to(res.dtype) if (res is not None and ctx.needs_input_grad[4]) else None
        gshift = g.reshape(-1, C).sum(0) if (shift is not None and ctx.needs_input_grad[3]) else None
        gscale = None
        if scale is not None and ctx.needs_input_grad[2]:
            pre = y.float() - (shift.view(1, 1, 1, 1, -1) if shift is not None else 0.0)
            if res is not None:
                pre = pre - res.float()
            gscale = (g * pre / scale.view(1, 1, 1, 1, -1)).reshape(-1, C).sum(0)
        gconv = g * scale.view(1, 1, 1, 1, -1) if scale is not None else g
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _ConvUnitFn._dgrad(gconv.to(x.dtype).contiguous(), w_eff, x.dtype, k, ctx.unit)
        if ctx.needs_input_grad[1]:
            # weight gradient = the HIP wgrad kernel (fp32 MFMA over the pixel axis) on the same channels-last buffers
            gw = ops.conv_wgrad(x, gconv, w_eff.shape[0], k).to(w_eff.dtype)
        return gx, gw, gscale, gshift, gres, None, None, None


class _BatchNormTrainFn(torch.autograd.Function):
    """nn.BatchNorm3d in TRAINING mode (+ the unit's ReLU) on a raw conv output z (channels-last): batch statistics, running-statistics
    update, backward through the statistics -- step_bn_train_forward / _backward (csrc/bn.hip).  models/i3dpt.py:95-110 with
    --freeze_stats False (models/networks.py:85-99)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, relu):
        if bn.momentum is None:
            raise NotImplementedError("step_amd: BatchNorm(momentum=None) (cumulative average) -- the reference builds its layers with the default 0.1")
        track = bn.track_running_stats and bn.running_mean is not None
        y, mean, invstd = ops.bn_train_forward(z.detach(), None if gamma is None else gamma.detach(), None if beta is None else beta.detach(),
                                               bn.running_mean if track else None, bn.running_var if track else None, bn.eps, bn.momentum, relu)
        if track and bn.num_batches_tracked is not None:
            bn.num_batches_tracked.add_(1)
        ctx.save_for_backward(z, y, gamma, mean, invstd)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        z, y, gamma, mean, invstd = ctx.saved_tensors
        gz, gg, gb = ops.bn_train_backward(z, y, gy, gamma, mean, invstd, ctx.relu)
        return (gz if ctx.needs_input_grad[0] else None, gg.to(gamma.dtype) if (gamma is not None and ctx.needs_input_grad[1]) else None,
                gb if ctx.needs_input_grad[2] else None, None, None)


def batchnorm_train(z, bn, relu):
    """y = relu?(BatchNorm3d_training(z)) for a channels-last raw conv output; updates bn's running statistics."""
    return _BatchNormTrainFn.apply(z, bn.weight, bn.bias, bn, relu)


class _PermuteColsFn(torch.autograd.Function):
    """w[:, perm] for a permutation `perm` of the columns (inv = its inverse): forward and backward are both gathers."""

    @staticmethod
    def forward(ctx, w, perm, inv):
        ctx.inv = inv
        return w.index_select(1, perm)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(1, ctx.inv), None, None


CAT_FUSE = True          # no-grad convs over a channel concat as ONE launch (ConvUnit.cat; False: two accumulating launches -- module switch for A/B and tests)
BATCH_PACK = True        # one-launch re-pack of every stale weight image (False: each unit packs its own; module switch for tests)
_UNITS = weakref.WeakSet()     # every live ConvUnit: an optimizer step stales all their packed images at once
_TABLES = {}                   # (dtype, device) -> (signature, device table, [(unit, cache key)]) of the last batched re-pack


def _repack_all(dtype, device):
    """Re-pack EVERY stale packed image (forward and data-gradient) of dtype / device held by a live ConvUnit with ONE launch
    (step_conv_pack_weights) into the buffers the units already own.  A training step changes every weight, and the ~290
    separate pack launches it caused cost more host time than GPU time.  Returns False when this call may not batch (the
    caller then packs its own weight as before): on a side stream the launch would not be ordered before the other streams' reads
    -- every side stream forks from the main one (wait_stream) after this point, so a launch on the main stream is."""
    if device.type == "cuda":
        cur = torch.cuda.current_stream(device)
        if any(cur == s for s in _SIDE.get((device.type, device.index if device.index is not None else torch.cuda.current_device()), ())):
            return False
    entries, owners = [], []
    for u in list(_UNITS):
        if u.version_fn is not None or not u._packed:
            continue
        w = u.weight_fn()
        if w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
            continue
        ver = _ver(w)
        wcin = w.shape[1]
        lo, hi = u.cin_slice if u.cin_slice is not None else (0, wcin)
        perm = None
        if u.perm is not None:
            perm = u._perm_dev.get(device)
            if perm is None:
                if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    import sys
                    print("step_amd._repack_all: unit %s k=%s weight %s packed keys %s has no device permutation table yet (capturing): falling back" % (
                        type(u.owner).__name__, u.k, tuple(w.shape), list(u._packed.keys())), file=sys.stderr)
                    return False                                 # (a host -> device copy: not inside a graph capture; the caller packs its own weight)
                perm = u._perm_dev[device] = u.perm.to(device=device, dtype=torch.int32).contiguous()
        for key, hit in list(u._packed.items()):             # (replicas on other devices add their images to the same dict)
            if key[0] != dtype or key[1] != device or hit[0] == ver:
                continue
            cpad = key[2] if len(key) > 2 else 0
            entries.append((w.data_ptr(), None if perm is None else perm.data_ptr(), hit[1].data_ptr(), w.shape[0], hi - lo, wcin,
                            lo, u.k, len(key) > 2, cpad))
            owners.append((u, key, ver, hit[1]))
    if not entries:
        return True
    order = sorted(range(len(entries)), key=lambda i: (entries[i][0], entries[i][8], entries[i][9], entries[i][2]))   # (the set's own order is not stable)
    entries, owners = [entries[i] for i in order], [owners[i] for i in order]
    sig = tuple(entries)      # the FULL descriptors: the caching allocator can hand the same three addresses to differently shaped weights of a rebuilt net
    cached = _TABLES.get((dtype, device))
    if cached is None or cached[0] != sig:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return False                                         # (a new table is a host -> device copy: not inside a graph capture)
        cached = _TABLES[(dtype, device)] = (sig, ops.pack_table(entries, device))
    with torch.no_grad():
        ops.pack_conv_weights(cached[1], len(entries), dtype, device)
    for u, key, ver, buf in owners:
        u._packed[key] = (ver, buf)
    return True


class ConvUnit:
    """Helper (not an nn.Module): owns the packed-weight / folded-affine caches of ONE conv and
    launches it.  `weight_fn()` returns the nn.Parameter in torch layout; `bn` an nn.BatchNorm3d or
    None; `bias_fn()` an nn.Parameter or None; `perm` an optional input-channel permutation folded in
    at pack time (effective channel j reads weight channel perm[j]); `cin_slice` restricts the conv
    to input channels [lo, hi) of the weight (a conv over a channel concat becomes two accumulating
    launches)."""

    def __init__(self, owner, weight_fn, k, bn_fn=None, bias_fn=None, perm=None, cin_slice=None, version_fn=None):
        # The getters take the OWNING MODULE as their argument instead of closing over it: copy.deepcopy() copies a
        # function by reference, so a closure would keep reading the ORIGINAL module's parameters from the copy's
        # forward; `owner` is deep-copied through the memo and becomes the copy.
        self.owner = owner
        # (the BatchNorm module too is reached through the owner: nn.DataParallel re-points a replica's children AFTER it copied
        # the module, so a stored reference would stay on the original)
        self._wf, self._bf, self._vf, self._bnf = weight_fn, bias_fn, version_fn, bn_fn
        self.bias_fn = None if bias_fn is None else self._bias
        self._src = None           # nn.DataParallel replica: the unit of the ORIGINAL module (versions and caches live there)
        self.k = tuple(k)
        self.perm, self.cin_slice = perm, cin_slice
        # version_fn: key of the packed-weight cache when weight_fn() builds a NEW tensor on every call (a torch.cat of
        # several parameters): the temporary's (data_ptr, _version) says nothing about the parameters behind it
        self.version_fn = None if version_fn is None else self._version
        self._packed = {}          # (dtype, device) -> (version, forward image); (dtype, device, cin_pad) -> data-gradient image
        self._perm_dev = {}
        self._affine = {}          # device -> (version, scale, shift)
        _UNITS.add(self)

    @property
    def bn(self):
        return None if self._bnf is None else self._bnf(self.owner)

    def replica_for(self, owner):
        """The unit of an nn.DataParallel replica of the owning module (torch.nn.parallel.replicate copies a module's __dict__
        every forward and hands the copy broadcast, non-leaf tensors of another device): same conv, getters bound to the REPLICA
        (its device's tensors), versions taken from the ORIGINAL's parameters (a fresh broadcast copy says nothing about the
        optimizer steps behind it) and caches shared with the original, keyed by device -- so a device's packed images survive
        from one forward's replica to the next.  Not in the batched re-pack registry (each replica packs its own weight)."""
        r = object.__new__(ConvUnit)
        r.__dict__ = dict(self.__dict__)
        r.owner = owner
        r._src = self._src or self
        r.bias_fn = None if r._bf is None else r._bias
        r.version_fn = None if r._vf is None else r._version
        return r

    def _wver(self, w):
        """cache version of the image packed from weight tensor w (this unit's device)"""
        if self._src is None:
            return _ver(w)
        ws = self._src.weight_fn()
        return ((ws.data_ptr(), ws._version, w.device),)

    def weight_fn(self):
        return self._wf(self.owner)

    def _bias(self):
        return self._bf(self.owner)

    def _version(self):
        if self._src is None:
            return self._vf(self.owner)
        dev = self._vf(self.owner)[0][2]                         # replica: the original's versions on this replica's device
        return tuple((a, b, dev) for a, b, _ in self._src._vf(self._src.owner))

    def __deepcopy__(self, memo):
        import copy
        new = ConvUnit(copy.deepcopy(self.owner, memo), self._wf, self.k, self._bnf, self._bf, self.perm, self.cin_slice, self._vf)
        memo[id(self)] = new
        return new                                               # (caches start empty: they hold device buffers of the original)

    @property
    def cout(self):
        return self.weight_fn().shape[0]

    def effective_weight(self):
        """[Cout, Cin_eff, kd, kh, kw] via differentiable view / gather ops."""
        w = self.weight_fn()
        w = w.reshape(w.shape[0], w.shape[1], -1)
        if self.cin_slice is not None:
            w = w[:, self.cin_slice[0]:self.cin_slice[1]]
        if self.perm is not None:
            pl = self._perm_dev.get((w.device, "long"))          # (cached: a host tensor would be copied -- and waited for -- per call)
            if pl is None:
                pl = self._perm_dev[(w.device, "long")] = self.perm.to(device=w.device, dtype=torch.long)
            if torch.is_grad_enabled() and w.requires_grad:
                # differentiable path: the gather is a PERMUTATION, so its backward is the gather with the inverse permutation -- autograd's own
                # backward of index_select is an atomic index_add_ (38 us per 60 x 12544 classifier weight and step in the round-6 trace, and a
                # summation order the hardware picks, although every destination receives exactly one value)
                inv = self._perm_dev.get((w.device, "inv"))
                if inv is None:
                    p_host = self.perm.detach().cpu().to(torch.long)
                    if p_host.numel() == w.shape[1] and bool((torch.sort(p_host).values == torch.arange(p_host.numel())).all()):
                        inv_h = torch.empty_like(p_host)
                        inv_h[p_host] = torch.arange(p_host.numel())
                        inv = inv_h.to(w.device)
                    else:
                        inv = False                              # (not a bijection of this weight's columns: autograd's index_select)
                    self._perm_dev[(w.device, "inv")] = inv
                w = _PermuteColsFn.apply(w, pl, inv) if inv is not False else w.index_select(1, pl)
            else:
                w = w.index_select(1, pl)
        return w.reshape(w.shape[0], w.shape[1], *self.k)

    def packed(self, dtype):
        if self.version_fn is not None:
            ver = self.version_fn()
            key = (dtype, ver[0][2] if ver else None)
            hit = self._packed.get(key)
            if hit is not None and hit[0] == ver:
                return hit[1]
        else:
            w = self.weight_fn()
            key = (dtype, w.device)
            ver = self._wver(w)
            hit = self._packed.get(key)
        if hit is not None and hit[0] != ver and self.version_fn is None and self._src is None and BATCH_PACK and _repack_all(dtype, w.device):
            hit = self._packed.get(key)                          # every stale image of the net, this one included, in one launch
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, ops.pack_conv_weight(self.effective_weight(), dtype))
            self._packed[key] = hit
        return hit[1]

    def packed_dgrad(self, w_eff, dtype, cin_pad):
        """Packed weight of the data-gradient conv (cached like the forward image; re-packed with it after an optimizer step)."""
        if self.version_fn is not None:
            return ops.pack_conv_weight_dgrad(w_eff, dtype, cin_pad)
        w = self.weight_fn()
        key = (dtype, w.device, cin_pad)
        ver = self._wver(w)
        hit = self._packed.get(key)
        if hit is not None and hit[0] != ver and self._src is None and BATCH_PACK and _repack_all(dtype, w.device):
            hit = self._packed.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, ops.pack_conv_weight_dgrad(w_eff, dtype, cin_pad))
            self._packed[key] = hit
        return hit[1]

    def _bn_affine(self):
        bn = self.bn
        scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
        return scale, bn.bias.float() - bn.running_mean.float() * scale

    def affine(self, differentiable=False):
        """fp32 (scale, shift) of the epilogue: folded eval-mode BN, or (None, bias), or (None, None)."""
        if self.bn is not None:
            bn = self.bn
            if bn.training:                                      # (callers test bn_training first: ConvUnit.__call__, Mixed.forward, Unit3D._forward_stem)
                raise RuntimeError("step_amd: ConvUnit.affine() folds EVAL-mode BatchNorm; a layer in training mode goes through batchnorm_train")
            if differentiable and (bn.weight.requires_grad or bn.bias.requires_grad):
                return self._bn_affine()
            dev = bn.weight.device
            bv = bn if self._src is None else self._src.bn          # (replica: the original's versions, this device's values)
            ver = tuple((a, b, dev) for a, b, _ in _ver(bv.weight, bv.bias, bv.running_mean, bv.running_var))
            hit = self._affine.get(dev)
            if hit is None or hit[0] != ver:
                with torch.no_grad():
                    scale, shift = self._bn_affine()
                hit = self._affine[dev] = (ver, scale.contiguous(), shift.contiguous())
            return hit[1], hit[2]
        if self.bias_fn is not None:
            b = self.bias_fn()
            return None, (b if b.dtype == torch.float32 else b.float())
        return None, None

    @property
    def bn_training(self):
        """The unit's BatchNorm normalises with batch statistics (nn.BatchNorm3d in training mode: --freeze_stats False)."""
        bn = self.bn
        return bn is not None and bn.training

    def _launch(self, x, relu, res, out, raw=False):
        scale, shift = (None, None) if raw else self.affine()
        if shift is not None:
            shift = shift.detach().contiguous()
        return ops.conv_forward(x, self.packed(x.dtype), self.cout, self.k, scale, shift, relu, res, out)

    def cat(self, xa, xb, relu=True, res=None):
        """This (pointwise, unsliced) conv over the channel concat [xa | xb] as ONE launch (ops.conv_forward_cat), or None: when a
        gradient is wanted, under batch-statistics BatchNorm, or where the library has no such form -- the caller then runs its two
        cin_slice units one after the other."""
        if not CAT_FUSE or self.cin_slice is not None or self.perm is not None or self.bn_training or self.k != (1, 1, 1):
            return None
        if torch.is_grad_enabled() and (xa.requires_grad or xb.requires_grad or self.weight_fn().requires_grad
                                        or (res is not None and res.requires_grad)):
            return None
        scale, shift = self.affine()
        if shift is not None:
            shift = shift.detach().contiguous()
        return ops.conv_forward_cat(xa, xb, self.packed(xa.dtype), self.cout, scale, shift, relu, res)

    def __call__(self, x, relu=True, res=None, out=None):
        w = self.weight_fn()
        if self.bn_training:
            # --freeze_stats False (models/networks.py:85-99 leaves BatchNorm in training mode): conv without an epilogue, then the
            # batch-statistics BatchNorm + ReLU as its own pass / autograd node (the statistics need the whole conv output first)
            if res is not None:
                raise NotImplementedError("step_amd: a residual input on a unit with batch-statistics BatchNorm")
            if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
                z = _ConvUnitFn.apply(x, self.effective_weight(), None, None, None, self, False, True)
            else:
                z = self._launch(x, False, None, None, raw=True)
            y = batchnorm_train(z, self.bn, relu)
            if out is not None:
                out.copy_(y)
                return out
            return y
        need_grad = torch.is_grad_enabled() and (
            x.requires_grad or w.requires_grad or (res is not None and res.requires_grad)
            or (self.bias_fn is not None and self.bias_fn().requires_grad)
            or (self.bn is not None and (self.bn.weight.requires_grad or self.bn.bias.requires_grad)))
        if not need_grad:
            return self._launch(x, relu, res, out)
        scale, shift = self.affine(differentiable=True)
        y = _ConvUnitFn.apply(x, self.effective_weight(), scale, shift, res, self, relu)
        if out is not None:
            out.copy_(y)
            return out
        return y


def _replicate_with_units(self):
    """nn.DataParallel support (train.py:142-144, test.py:82-84, demo.py:79-81 wrap base_net / context_net unconditionally).
    torch.nn.parallel.replicate shallow-copies a module's __dict__ for every device on every forward; the helper objects a
    step_amd module keeps next to its parameters (ConvUnit, _FusedPointwise) would otherwise stay bound to the ORIGINAL module
    -- device-0 parameters, device-0 packed weights.  Each replica gets helpers bound to itself (ConvUnit.replica_for).
    The fast multi-GPU mode remains one process per GPU (step_amd.dist); this makes the reference scripts run unchanged."""
    replica = nn.Module._replicate_for_data_parallel(self)
    for name, val in self.__dict__.items():
        if isinstance(val, (ConvUnit, _FusedPointwise)):
            replica.__dict__[name] = val.replica_for(replica)
    return replica


class Unit3D(nn.Module):
    """conv3d (+ frozen BN) + ReLU.  Keys: conv3d.weight[, conv3d.bias], batch3d.{weight,bias,running_*}."""
    _replicate_for_data_parallel = _replicate_with_units

    def __init__(self, in_channels, out_channels, kernel_size=(1, 1, 1), stride=(1, 1, 1), use_bias=False, use_bn=True,
                 relu=True):
        super().__init__()
        self.kernel_size, self.stride, self.relu = tuple(kernel_size), tuple(stride), relu
        self.conv3d = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, bias=use_bias)
        if use_bn:
            self.batch3d = nn.BatchNorm3d(out_channels, eps=BN_EPS)
        self.is_stem = self.kernel_size == (7, 7, 7)
        if self.is_stem:
            assert self.stride == (2, 2, 2) and in_channels == 3
            self._stem_packed = {}
            self._unit = ConvUnit(self, lambda m: m.conv3d.weight, (7, 7, 7), bn_fn=(lambda m: m.batch3d) if use_bn else None)
        else:
            assert self.stride == (1, 1, 1)
            self._unit = ConvUnit(self, lambda m: m.conv3d.weight, self.kernel_size, bn_fn=(lambda m: m.batch3d) if use_bn else None,
                                  bias_fn=(lambda m: m.conv3d.bias) if use_bias else None)

    def forward(self, x, out=None):
        if self.is_stem:
            return self._forward_stem(x, out)
        return self._unit(x, relu=self.relu, out=out)

    def stem_packed(self, dtype):
        w = self.conv3d.weight
        key = (dtype, w.device)
        ver = self._unit._wver(w)
        hit = self._stem_packed.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, ops.pack_stem_weight(w, dtype))
            self._stem_packed[key] = hit
        return hit[1]

    def _forward_stem(self, x, out):
        """x is the clip in the reference layout [N,T,3,H,W]."""
        w = self.conv3d.weight
        hit = (None, self.stem_packed(x.dtype))
        bn = getattr(self, "batch3d", None)
        bn_grad = bn is not None and (bn.weight.requires_grad or bn.bias.requires_grad)
        if self._unit.bn_training:                                     # --freeze_stats False: raw stem conv, then batch-statistics BN + ReLU
            if torch.is_grad_enabled() and w.requires_grad:
                z = _StemFn.apply(x, w, None, None, self, hit[1], False)
            else:
                z = ops.stem_forward(x, hit[1], w.shape[0], None, None, None, relu=False)
            y = batchnorm_train(z, bn, self.relu)
            if out is not None:
                out.copy_(y)
                return out
            return y
        if torch.is_grad_enabled() and (w.requires_grad or bn_grad):
            scale, shift = self._unit.affine(differentiable=True)      # --freeze_affine False: the BN affine trains too
            return _StemFn.apply(x, w, scale, shift, self, hit[1])
        scale, shift = self._unit.affine()
        return ops.stem_forward(x, hit[1], w.shape[0], scale, shift, out)


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, scale, shift, unit, packed, relu=True):
        det = lambda t: None if t is None else t.detach().contiguous()
        y = ops.stem_forward(x.detach(), packed, w.shape[0], det(scale), det(shift), None, relu=relu)
        ctx.save_for_backward(x, w, scale, shift, y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, scale, shift, y = ctx.saved_tensors
        if not ctx.relu and scale is None:                             # the raw conv (batch-statistics BN follows): gy IS the conv's gradient
            gw = None
            if ctx.needs_input_grad[1]:
                if WGRAD16 and x.dtype != torch.float32 and gy.dtype == x.dtype:
                    gw = ops.stem_wgrad16(x, gy.contiguous(), w.shape[0])
                if gw is None:
                    gw = ops.stem_wgrad(x.contiguous(), gy.float(), w.shape[0])
                gw = gw.to(w.dtype)
            return None, gw, None, None, None, None, None
        if WGRAD16 and ctx.needs_input_grad[1] and not (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]) and y.dtype != torch.float32 \
                and x.dtype == y.dtype:
            # frozen affine, 16-bit activations: ReLU mask x scale in ONE HIP pass (16-bit result), then the weight gradient on the
            # 16-bit matrix instructions reading clip and gradient once (step_stem_wgrad16) -- as the other units' step_conv_wgrad16
            fused = ops.act_grad(y, gy, scale, True, want_f32=False, want_act=True)
            if fused is not None:
                gw = ops.stem_wgrad16(x, fused[1], w.shape[0])
                if gw is not None:
                    return None, gw.to(w.dtype), None, None, None, None, None
        g = gy.float() * (y > 0).to(torch.float32)
        C = g.shape[-1]
        gw = gscale = gshift = None
        if ctx.needs_input_grad[3]:
            gshift = g.reshape(-1, C).sum(0)
        if ctx.needs_input_grad[2]:                                    # same formulas as _ConvUnitFn (pre-affine output = (y - shift) / scale)
            gscale = (g * (y.float() - shift.view(1, 1, 1, 1, -1)) / scale.view(1, 1, 1, 1, -1)).reshape(-1, C).sum(0)
        if ctx.needs_input_grad[1]:
            gw = ops.stem_wgrad(x.contiguous(), g * scale.view(1, 1, 1, 1, -1), w.shape[0]).to(w.dtype)   # HIP (the clip itself needs no gradient)
        return None, gw, gscale, gshift, None, None, None


class MaxPoolTF(nn.Module):
    """TF-SAME max pool (zero-VALUED padding, ceil_mode).  models/i3dpt.py:114-126."""

    def __init__(self, kernel_size, stride):
        super().__init__()
        self.kernel_size, self.stride = tuple(kernel_size), tuple(stride)

    def forward(self, x, out=None):
        if torch.is_grad_enabled() and x.requires_grad:
            return _MaxPoolFn.apply(x, self.kernel_size, self.stride)
        return ops.maxpool_tf(x, self.kernel_size, self.stride, out)


class _MaxPoolFn(torch.autograd.Function):
    """forward and backward = the HIP pool kernels (the gradient goes to the first maximum of a window, torch's rule)."""

    @staticmethod
    def forward(ctx, x, k, s):
        ctx.k, ctx.s = k, s
        ctx.save_for_backward(x)
        return ops.maxpool_tf(x.detach(), k, s)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return ops.maxpool_tf_backward(x, gy, ctx.k, ctx.s).to(x.dtype), None, None


class _FusedPointwise:
    """The 1x1x1 convs of an Inception block that read the block input (branch_0, branch_1.0,
    branch_2.0) as ONE launch: their weights and folded BN affines are concatenated along Cout and the
    kernel's two-destination epilogue sends the first unit's channels to `out` and the rest to `out2`
    (the input is read once instead of three times; 2 launches fewer per block).  Inference only."""

    def __init__(self, owner):
        self.owner = owner               # the Mixed block: its children are read at call time (a DataParallel replica's are re-pointed after the copy)
        self._src = None
        self._cache = {}                 # (dtype, device) -> (version, packed, scale, shift, cout, split); shared with replicas

    @staticmethod
    def _units(m):
        return (m.branch_0, m.branch_1[0], m.branch_2[0])

    def replica_for(self, owner):
        r = object.__new__(_FusedPointwise)
        r.owner, r._src, r._cache = owner, self._src or self, self._cache
        return r

    def __call__(self, x, out, out2, pool=False):
        us = self._units(self.owner)
        dev = us[0].conv3d.weight.device
        key = (x.dtype, dev)
        vs = us if self._src is None else self._units(self._src.owner)     # (replica: the original's versions)
        ver = tuple((a, b, dev) for u in vs for a, b, _ in _ver(u.conv3d.weight, u.batch3d.weight, u.batch3d.bias, u.batch3d.running_mean,
                                                                u.batch3d.running_var))
        hit = self._cache.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                w = torch.cat([u.conv3d.weight for u in us], 0)
                aff = [u._unit.affine() for u in us]
                scale = torch.cat([a[0] for a in aff]).contiguous()
                shift = torch.cat([a[1] for a in aff]).contiguous()
                hit = (ver, ops.pack_conv_weight(w, x.dtype), scale, shift, w.shape[0], us[0].conv3d.weight.shape[0])
            self._cache[key] = hit
        _, packed, scale, shift, cout, split = hit
        if pool:
            # the block's 3x3x3 / 1 max pool rides in the same grid where the library has that form (the 14x14 maps): returns the pooled x
            p = ops.pool_conv_forward(x, packed, cout, scale, shift, True, out, out2, split)
            if p is not None:
                return p
        ops.conv_forward(x, packed, cout, (1, 1, 1), scale, shift, True, None, out, out2, split)
        return None


def _unit_wgrad(unit, w_eff, x, g, k, pend=None):
    """Weight gradient of one conv unit from its input x and the gradient g of its conv output (both channels-last, g possibly a channel
    slice): inside wgrad_into_grad() accumulated straight into the parameter's .grad on the side stream (returns None), else returned.
    pend (a list): the fixed-order sum of the partial tiles is deferred and described there -- _flush_wgrads() runs the sums of a whole
    block as one launch (and only then announces the parameters to the gradient exchange)."""
    w16 = WGRAD16 and x.dtype != torch.float32 and g.dtype == x.dtype
    if not w16 and g.dtype != torch.float32:
        g = g.float()
    fn = ops.conv_wgrad16 if w16 else ops.conv_wgrad
    cout = w_eff.shape[0]
    target = _wgrad_target(unit, w_eff) if (WGRAD_INTO_GRAD and x.is_cuda and ops.PROFILE is None) else None
    defer = pend is not None and ops.PROFILE is None and (ops.WGRAD16_WS if w16 else ops.WGRAD_WS)
    if target is None:
        if defer:
            dw, item, ws = ops.conv_wgrad_partial(x, g, cout, k)
            pend.append((item, ws, None, None, x.device))
            assert dw.dtype == w_eff.dtype                       # (fp32 master weights: a cast here would read the gradient before its sum has run)
            return dw
        return fn(x, g, cout, k).to(w_eff.dtype)
    main = torch.cuda.current_stream(x.device)
    side = _side_streams(x.device)[0]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        if defer:
            _, item, ws = ops.conv_wgrad_partial(x, g, cout, k, into=target)
            pend.append((item, ws, unit, side, x.device))
            if GRAD_DEFER is not None:
                GRAD_DEFER(unit.weight_fn())
        else:
            fn(x, g, cout, k, into=target)
    x.record_stream(side)
    g.record_stream(side)
    _PENDING[0] = True
    if GRAD_READY is not None and not defer:                     # this parameter bypasses autograd's accumulate hook
        GRAD_READY(unit.weight_fn(), side)
    return None


def _flush_wgrads(pend, device):
    """The deferred sums collected by _unit_wgrad(pend=...), one launch (per eight layers), on the stream their partial tiles were
    written on; then the gradient exchange hears about the parameters."""
    if not pend:
        return
    # One launch PER STREAM the partial tiles were written on (None = the current stream, whose dw autograd reads next: its sum must be
    # ordered on that stream, not on the side stream of a neighbour that happens to have a dense fp32 .grad).
    parts = {}
    for e in pend:
        parts.setdefault((e[4], e[3]), []).append(e)
    for (dev, side), es in parts.items():
        items = [e[0] for e in es]
        if side is None:
            ops.wgrad_reduce_group(items, dev if dev is not None else device)
            continue
        with torch.cuda.stream(side):
            ops.wgrad_reduce_group(items, dev if dev is not None else device)
        for _, ws, unit, _, _ in es:
            if ws is not None:
                ws.record_stream(side)
            if GRAD_READY is not None and unit is not None:
                GRAD_READY(unit.weight_fn(), side)
    del pend[:]


class _MixedTrainFn(torch.autograd.Function):
    """One autograd node for a whole Inception block in training (eval-mode BN with a frozen affine: every shipped configuration).
    Forward: the block's six units write channel slices of one output buffer and one bottleneck buffer (no torch.cat), the two 3x3x3
    convs and branch_3's 1x1x1 conv as one grouped launch -- 5 launches.  Backward, hand-scheduled instead of six per-unit nodes plus
    autograd's concat / accumulate glue:
        1 pass   ReLU mask x BN scale over the whole concat gradient (one act_grad instead of four)
        1 launch the two 3x3x3 data gradients (grouped), into one bottleneck-gradient buffer
        1 pass   ReLU mask x scale over that buffer (instead of two)
        the pooled branch: 1x1x1 data gradient, max-pool backward (2 launches)
        the block-input gradient: branch_0's, branch_1.0's and branch_2.0's 1x1x1 data gradients ACCUMULATE into the pool's result
        through the conv's residual input -- no torch add, no clear
        6 weight gradients, reading their activation-gradient slices in place
    The same kernels and the same operands as the per-unit nodes (the sums into the block-input gradient run in a fixed order:
    pool, branch_0, branch_1.0, branch_2.0)."""

    @staticmethod
    def forward(ctx, x, w0, w1a, w1b, w2a, w2b, w3, block):
        xd = x.detach()
        out, t, p = block._forward_sliced(xd)
        ctx.block = block
        ctx.save_for_backward(x, t, p, out, w0, w1a, w1b, w2a, w2b, w3)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, t, p, out, w0, w1a, w1b, w2a, w2b, w3 = ctx.saved_tensors
        m = ctx.block
        oc = m.oc
        c0, c1, c2 = oc[0], oc[0] + oc[2], oc[0] + oc[2] + oc[4]
        u0, u1a, u1b, u2a, u2b, u3 = m._train_units()
        need_x = ctx.needs_input_grad[0]
        need_w = ctx.needs_input_grad[1:7]
        # 1. the whole concat gradient through the four final ReLUs / BN scales in one pass
        r = ops.act_grad(out, gy, m._train_scales(out.device)[0], True, want_f32=False, want_act=True)
        if r is None:                                            # (a gradient that is not channels-last, e.g. through BaseNet's output permute)
            r = ops.act_grad(out, gy.contiguous(), m._train_scales(out.device)[0], True, want_f32=False, want_act=True)
        gact = r[1]
        g0, g1, g2, g3 = gact[..., :c0], gact[..., c0:c1], gact[..., c1:c2], gact[..., c2:]
        gws = [None] * 6
        pend = []                                                # the six fixed-order sums of partial tiles run as ONE launch at the end
        if need_w[0]:
            gws[0] = _unit_wgrad(u0, w0, x, g0, (1, 1, 1), pend)
        if need_w[2]:
            gws[2] = _unit_wgrad(u1b, w1b, t[..., :oc[1]], g1, (3, 3, 3), pend)
        if need_w[4]:
            gws[4] = _unit_wgrad(u2b, w2b, t[..., oc[1]:], g2, (3, 3, 3), pend)
        if need_w[5]:
            gws[5] = _unit_wgrad(u3, w3, p, g3, (1, 1, 1), pend)
        # 2. the two 3x3x3 data gradients into one bottleneck-gradient buffer (one grouped launch where the library merges them)
        gt = torch.empty(t.shape, dtype=t.dtype, device=t.device)
        dt_ = out.dtype
        ops.conv_forward_group([
            (g1, u1b.packed_dgrad(w1b, dt_, oc[2]), oc[1], (3, 3, 3), None, None, False, gt[..., :oc[1]]),
            (g2, u2b.packed_dgrad(w2b, dt_, oc[4]), oc[3], (3, 3, 3), None, None, False, gt[..., oc[1]:])])
        # 3. through the two bottleneck ReLUs / BN scales in one pass
        _, gta = ops.act_grad(t, gt, m._train_scales(out.device)[1], True, want_f32=False, want_act=True)
        if need_w[1]:
            gws[1] = _unit_wgrad(u1a, w1a, x, gta[..., :oc[1]], (1, 1, 1), pend)
        if need_w[3]:
            gws[3] = _unit_wgrad(u2a, w2a, x, gta[..., oc[1]:], (1, 1, 1), pend)
        _flush_wgrads(pend, x.device)
        gx = None
        if need_x:
            cin = x.shape[-1]
            # 4. the pooled branch, then the three pointwise data gradients accumulate into its result (the conv's residual input)
            gp = ops.conv_forward(g3, u3.packed_dgrad(w3, dt_, oc[5]), cin, (1, 1, 1), None, None, False, None, None)
            gx = ops.maxpool_tf_backward(x, gp, (3, 3, 3), (1, 1, 1))
            if gx.dtype != x.dtype:
                gx = gx.to(x.dtype)
            # (res == out: every element is read and written by the same thread; measured equal to the out-of-place form on the GPU)
            ops.conv_forward(g0, u0.packed_dgrad(w0, dt_, oc[0]), cin, (1, 1, 1), None, None, False, gx, gx)
            ops.conv_forward(gta[..., :oc[1]], u1a.packed_dgrad(w1a, dt_, oc[1]), cin, (1, 1, 1), None, None, False, gx, gx)
            ops.conv_forward(gta[..., oc[1]:], u2a.packed_dgrad(w2a, dt_, oc[3]), cin, (1, 1, 1), None, None, False, gx, gx)
        return (gx,) + tuple(gws) + (None,)


MIXED_TRAIN_FN = True     # training: an Inception block is ONE autograd node with a hand-scheduled backward (_MixedTrainFn); False: six per-unit nodes + torch.cat


class Mixed(nn.Module):
    """Inception block; the four branches write channel slices of one buffer (order b0,b1,b2,b3)."""
    _replicate_for_data_parallel = _replicate_with_units

    def __init__(self, in_channels, oc):
        super().__init__()
        self.oc = tuple(oc)
        self.branch_0 = Unit3D(in_channels, oc[0])
        self.branch_1 = nn.Sequential(Unit3D(in_channels, oc[1]), Unit3D(oc[1], oc[2], (3, 3, 3)))
        self.branch_2 = nn.Sequential(Unit3D(in_channels, oc[3]), Unit3D(oc[3], oc[4], (3, 3, 3)))
        self.branch_3 = nn.Sequential(MaxPoolTF((3, 3, 3), (1, 1, 1)), Unit3D(in_channels, oc[5]))
        self.out_channels = oc[0] + oc[2] + oc[4] + oc[5]
        self._fused = _FusedPointwise(self)

    def _train_units(self):
        return (self.branch_0._unit, self.branch_1[0]._unit, self.branch_1[1]._unit, self.branch_2[0]._unit, self.branch_2[1]._unit,
                self.branch_3[1]._unit)

    def _train_scales(self, device):
        """(folded BN scales of [b0 | b1b | b2b | b3], of [b1a | b2a]) concatenated in buffer order, cached on the BN tensors' versions"""
        u0, u1a, u1b, u2a, u2b, u3 = self._train_units()
        ver = tuple(v for u in (u0, u1a, u1b, u2a, u2b, u3) for v in _ver(u.bn.weight, u.bn.running_var)) + (device,)
        hit = self.__dict__.get("_scales_cache")
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                so = torch.cat([u.affine()[0] for u in (u0, u1b, u2b, u3)]).contiguous()
                st = torch.cat([u.affine()[0] for u in (u1a, u2a)]).contiguous()
            hit = (ver, so, st)
            self.__dict__["_scales_cache"] = hit
        return hit[1], hit[2]

    def _forward_sliced(self, x):
        """The six units writing channel slices of one output buffer and one bottleneck buffer, every unit with its OWN packed weight
        (the batched re-pack after an optimizer step covers them): b0, b1a, b2a, the pool, then [b1b, b2b, b3] as one grouped launch.
        Returns (out, bottleneck buffer, pooled input)."""
        oc = self.oc
        N, D, H, W, _ = x.shape
        c0, c1, c2 = oc[0], oc[0] + oc[2], oc[0] + oc[2] + oc[4]
        out = torch.empty((N, D, H, W, self.out_channels), dtype=x.dtype, device=x.device)
        t = torch.empty((N, D, H, W, oc[1] + oc[3]), dtype=x.dtype, device=x.device)
        u0, u1a, u1b, u2a, u2b, u3 = self._train_units()
        u0._launch(x, True, None, out[..., :c0])
        u1a._launch(x, True, None, t[..., :oc[1]])
        u2a._launch(x, True, None, t[..., oc[1]:])
        p = ops.maxpool_tf(x, (3, 3, 3), (1, 1, 1))
        m = []
        for u, xin, o in ((u1b, t[..., :oc[1]], out[..., c0:c1]), (u2b, t[..., oc[1]:], out[..., c1:c2]), (u3, p, out[..., c2:])):
            scale, shift = u.affine()
            m.append((xin, u.packed(x.dtype), u.cout, u.k, scale, None if shift is None else shift.detach().contiguous(), True, o))
        ops.conv_forward_group(m)
        return out, t, p

    def _train_fn_ok(self, x):
        """_MixedTrainFn's contract: eval-mode BN with a frozen affine on all six units, plain fp32 parameters, channel counts on the
        kernels' 16-byte grid, not a DataParallel replica."""
        if not MIXED_TRAIN_FN or x.dim() != 5:
            return False
        vec = 16 // x.element_size()
        for u in self._train_units():
            bn = u.bn
            if bn is None or bn.training or bn.weight.requires_grad or bn.bias.requires_grad or u._src is not None:
                return False
            w = u.weight_fn()
            if w.dtype != torch.float32 or not w.is_contiguous() or w.shape[0] % vec or w.shape[1] % vec:
                return False
        return True

    def forward(self, x, out=None):
        oc = self.oc
        N, D, H, W, _ = x.shape
        grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        bn_train = any(u._unit.bn_training for u in (self.branch_0, self.branch_1[0], self.branch_1[1], self.branch_2[0], self.branch_2[1], self.branch_3[1]))
        if grad and not bn_train and self._train_fn_ok(x):
            u0, u1a, u1b, u2a, u2b, u3 = self._train_units()
            y = _MixedTrainFn.apply(x, u0.weight_fn(), u1a.weight_fn(), u1b.weight_fn(), u2a.weight_fn(), u2b.weight_fn(), u3.weight_fn(), self)
            if out is not None:
                out.copy_(y)
                return out
            return y
        if grad or bn_train:
            # autograd path: branches return fresh tensors, concatenated by torch (bookkeeping only)
            y = torch.cat([self.branch_0(x), self.branch_1[1](self.branch_1[0](x)), self.branch_2[1](self.branch_2[0](x)),
                           self.branch_3[1](self.branch_3[0](x))], dim=-1)
            if out is not None:
                out.copy_(y)
                return out
            return y
        if out is None:
            out = torch.empty((N, D, H, W, self.out_channels), dtype=x.dtype, device=x.device)
        c0, c1, c2 = oc[0], oc[0] + oc[2], oc[0] + oc[2] + oc[4]
        t = torch.empty((N, D, H, W, oc[1] + oc[3]), dtype=x.dtype, device=x.device)
        if BRANCH_STREAMS == 0 or not x.is_cuda:
            # ONE stream, three or four launches: pool, the fused 1x1x1 triple, and the two 3x3x3 convs + branch_3's 1x1x1 as ONE grid
            # (ops.conv_forward_group: the narrow branch_2 conv runs on the CUs the wide branch_1 conv leaves idle).  Measured on
            # the replayed C2 step (tools/graph_timeline.py): a cross-stream edge of the HIP graph costs ~6 us of whole-GPU idle
            # (fork + join: 11-16 us per block), kernels that each fill the chip with one workgroup per CU hardly overlap, and a
            # single-stream graph runs its kernels back to back with no gap -- 1469 us against 1477 / 1505 us with 1 / 2 side streams.
            # branch_3's 1x1x1 conv on the pooled tensor travels as a third member: on the 14x14 maps the two 3x3x3 convs leave
            # 32-88 CUs idle and the library appends its workgroups to their grid (elsewhere it is launched behind them)
            pool = self.branch_3[0]
            p = self._fused(x, out[..., :c0], t, pool=POOL_WITH_POINTWISE and pool.kernel_size == (3, 3, 3) and pool.stride == (1, 1, 1))
            if p is None:
                p = pool(x)
            u1, u2, u3 = self.branch_1[1]._unit, self.branch_2[1]._unit, self.branch_3[1]._unit
            m = []
            for u, xin, o in ((u1, t[..., :oc[1]], out[..., c0:c1]), (u2, t[..., oc[1]:], out[..., c1:c2]), (u3, p, out[..., c2:])):
                scale, shift = u.affine()
                m.append((xin, u.packed(x.dtype), u.cout, u.k, scale, None if shift is None else shift.detach().contiguous(), True, o))
            ops.conv_forward_group(m)
            return out
        # The branches are independent: run them on three HIP streams (fork/join with events -- under
        # hipGraph capture these become parallel graph branches).  Most of these launches do not fill
        # 256 CUs on the 28x28 / 14x14 maps, so they overlap instead of queueing behind each other.
        main = torch.cuda.current_stream(x.device)
        s1, s2 = _side_streams(x.device)
        if BRANCH_STREAMS == 1:
            s1 = s2                                        # both side branches on ONE side stream (one fork, one join)
        s2.wait_stream(main)
        with torch.cuda.stream(s2):                        # branch_3: pool -> 1x1x1
            p = self._branch_3(x, out[..., c2:])
        self._fused(x, out[..., :c0], t)                   # main: fused 1x1x1 convs
        s1.wait_stream(main)
        with torch.cuda.stream(s1):                        # branch_2: small 3x3x3
            self.branch_2[1](t[..., oc[1]:], out=out[..., c1:c2])
        self.branch_1[1](t[..., :oc[1]], out=out[..., c0:c1])   # main: the big 3x3x3
        main.wait_stream(s1)
        if s2 is not s1:
            main.wait_stream(s2)
        if p is not None:
            p.record_stream(main)
        return out

    def _branch_3(self, x, out):
        """max pool 3x3x3 / 1 -> 1x1x1 unit into `out` (inference path): two launches; returns the pooled scratch tensor.
        (One fused launch -- the pooled tensor never reaching memory -- was built twice and measured SLOWER both times: the seven
        branch_3 layers of a C2 step 193 us as two launches, 350 / 328 us fused; the per-slab pool pass, up to 54 dependent LDS
        reads per thread between two barriers, costs more than the saved traffic.  Removed in round 3)"""
        pool, unit = self.branch_3[0], self.branch_3[1]
        p = pool(x)
        unit(p, out=out)
        return p


WGRAD16 = True           # 16-bit activations: weight gradients on the 16-bit MFMA (step_conv_wgrad16)
WGRAD_INTO_GRAD = False        # see wgrad_into_grad()
GRAD_READY = None              # step_amd.dist.BucketedReducer.ready while a backward pass is being overlapped with the exchange
GRAD_DEFER = None              # ... its defer(): this parameter's gradient will be announced by _flush_wgrads, not by autograd's hook
_PENDING = [False]
_PEND_Q = []                   # deferred weight-gradient sums of the per-unit nodes (flushed eight at a time and in wgrad_sync())
_KEEP = []                     # tensors the side stream reads that autograd must not modify in place (until wgrad_sync())


def _wgrad_target(unit, w_eff):
    """The dense fp32 .grad of the unit's weight parameter when the effective weight IS that parameter (no channel slice /
    permutation) and a gradient buffer exists; else None (the caller returns the gradient through autograd)."""
    if unit.cin_slice is not None or unit.perm is not None:
        return None
    w = unit.weight_fn()
    g = w.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != w_eff.numel() or not w.requires_grad:
        return None
    return g


class wgrad_into_grad:
    """Context manager for a training step whose optimizer owns persistent gradient buffers (step_amd.optim.FlatAdam):
    inside it the conv units add their weight gradients into `weight.grad` directly, asynchronously on a side stream.
    Leaving the context (or calling wgrad_sync()) makes the current stream wait for them.  Not for torch.autograd.grad()
    or double backward: the weight gradient bypasses autograd."""

    def __enter__(self):
        global WGRAD_INTO_GRAD
        if not WGRAD_INTO_GRAD:
            wgrad_sync()                                         # nothing of an earlier backward may ride into this step's sums
        self.prev, WGRAD_INTO_GRAD = WGRAD_INTO_GRAD, True
        return self

    def __exit__(self, *exc):
        global WGRAD_INTO_GRAD
        WGRAD_INTO_GRAD = self.prev
        if exc and exc[0] is not None:
            # backward raised mid-way: the queued partial sums describe a gradient nobody will use -- drop them instead of adding them
            # to the arena at the next flush (their workspaces die with the list); the side stream is still joined.  The gradients of
            # this step are UNDEFINED after a failed backward (some weight gradients are in the arena, the dropped ones are not): the
            # announcement hooks are cleared so that a BucketedReducer armed for this step does not wait for parameters that will never
            # report (its next begin() re-arms it), and the caller must zero the arena (opt.zero_grad()) before the next backward.
            global GRAD_READY, GRAD_DEFER
            del _PEND_Q[:]
            GRAD_READY = GRAD_DEFER = None
        wgrad_sync()
        return False


def wgrad_sync():
    """Order every weight gradient launched on the side stream before what the current stream does next."""
    if _PEND_Q:
        _flush_wgrads(_PEND_Q, _PEND_Q[0][4])
    if _PENDING[0] and torch.cuda.is_available():
        for dev_key, streams in list(_SIDE.items()):
            torch.cuda.current_stream(torch.device(dev_key[0], dev_key[1])).wait_stream(streams[0])
        _PENDING[0] = False
    del _KEEP[:]


POOL_WITH_POINTWISE = True     # inference: an Inception block's max pool and its fused 1x1x1 triple as one launch where the library has the form
FUSE_STEM_POOL = True          # inference: maxPool3d_2a is taken on the stem's tiles while they are on the chip (ops.stem_pool_forward)
FUSE_STEM_U8 = False           # BaseNet.forward_u8: True = the stem stages uint8 frames itself (ops.stem_pool_forward_u8, bit-identical); False (default) = one conversion
                               # pass in front -- MEASURED: the table look-ups in the stem's frame staging cost more than the pass they save (fed C2 6.75 k against 6.96 k clips/s)
FUSE_CONV_POOL = True          # inference: maxPool3d_3a is taken on conv3d_2c's tiles while they are on the chip (ops.conv_forward_pre_pool; needs FUSE_POINTWISE_INPUT)
FUSE_POINTWISE_INPUT = True    # inference: a 64 -> 64 1x1x1 unit directly in front of a 3x3x3 unit runs inside that unit's launch (ops.conv_forward_pre)


def _stem_then_pool(a, b, x, u8=None):
    """Unit3D a (the 7x7x7 stem, eval-mode BN, ReLU) followed by MaxPoolTF b ((1,3,3) / (1,2,2)): one call when neither needs autograd
    and the library has the fused form (16-bit clips); None otherwise -- the caller runs them one after the other (bit-identical).
    u8 = (dtype, scale_mode, mean, std): x is uint8 frames [N,T,H,W,3] and the normalisation happens in the stem's staging
    (ops.stem_pool_forward_u8)."""
    if not (isinstance(a, Unit3D) and a.is_stem and isinstance(b, MaxPoolTF)) or not x.is_cuda or x.dtype == torch.float32:
        return None
    if b.kernel_size != (1, 3, 3) or b.stride != (1, 2, 2) or not a.relu or a._unit.bn_training or a._unit.bn is None:
        return None
    w = a.conv3d.weight
    bn = a.batch3d
    if torch.is_grad_enabled() and (w.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad or x.requires_grad):
        return None
    scale, shift = a._unit.affine()
    if u8 is not None:
        dtype, mode, mean, std = u8
        return ops.stem_pool_forward_u8(x, dtype, a.stem_packed(dtype), w.shape[0], scale, shift, mode, mean, std)
    return ops.stem_pool_forward(x, a.stem_packed(x.dtype), w.shape[0], scale, shift)


def _pointwise_then_3x3x3(a, b, x, pool=None):
    """Unit3D a (1x1x1, BN, ReLU) followed by Unit3D b (3x3x3): one launch when neither needs autograd and the library has the fused
    form for the shapes (16-bit, 64 -> 64 pointwise); None otherwise -- the caller runs the units one after the other (same result,
    bit for bit).  pool: the MaxPoolTF (1,3,3) / (1,2,2) behind b -- then the pool is taken on b's tiles too (ops.conv_forward_pre_pool)
    and the POOLED tensor is returned, or None when that form does not exist for the shape."""
    if not (isinstance(a, Unit3D) and isinstance(b, Unit3D)) or a.is_stem or b.is_stem or not x.is_cuda or x.dtype == torch.float32:
        return None
    ua, ub = a._unit, b._unit
    if ua.bn_training or ub.bn_training:
        return None
    if tuple(ua.k) != (1, 1, 1) or tuple(ub.k) != (3, 3, 3) or not a.relu or ua.cout != 64 or x.shape[-1] != 64:
        return None
    for u in (ua, ub):
        w = u.weight_fn()
        if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (u.bias_fn is not None and u.bias_fn().requires_grad)
                                        or (u.bn is not None and (u.bn.weight.requires_grad or u.bn.bias.requires_grad))):
            return None
    sa, ha = ua.affine()
    sb, hb = ub.affine()
    if sa is None or ha is None:
        return None
    ha = ha.detach().contiguous()
    if hb is not None:
        hb = hb.detach().contiguous()
    pre = (ua.packed(x.dtype), sa.detach().contiguous(), ha, ua.cout)
    if pool is not None:
        if not isinstance(pool, MaxPoolTF) or pool.kernel_size != (1, 3, 3) or pool.stride != (1, 2, 2) or not b.relu or sb is None or hb is None:
            return None
        return ops.conv_forward_pre_pool(x, ub.packed(x.dtype), ub.cout, ub.k, sb, hb, b.relu, pre)
    return ops.conv_forward_pre(x, ub.packed(x.dtype), ub.cout, ub.k, sb, hb, b.relu, pre)


BRANCH_STREAMS = 0       # Inception blocks (inference path): 0 = one stream + the grouped 3x3x3 launch (default since round 3, see Mixed.forward); 1 / 2 = side branches on 1 / 2 side streams
WGRAD_SIDE_STREAM = True # training: weight gradient beside the data gradient
_SIDE = {}


def _side_streams(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE:
        _SIDE[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _SIDE[key]


def build_base_i3d(kinetics_pretrain=None, freeze_affine=True):
    """The 13 backbone stages in an nn.Sequential whose indices (0..12) are part of the parameter-name
    contract (models/networks.py:107-142; utils/solver.py:21-37 string-matches `base_model.N`)."""
    stages = [
        Unit3D(3, 64, (7, 7, 7), (2, 2, 2)),            # 0  conv3d_1a_7x7
        MaxPoolTF((1, 3, 3), (1, 2, 2)),                # 1  maxPool3d_2a_3x3
        Unit3D(64, 64, (1, 1, 1)),                      # 2  conv3d_2b_1x1
        Unit3D(64, 192, (3, 3, 3)),                     # 3  conv3d_2c_3x3
        MaxPoolTF((1, 3, 3), (1, 2, 2)),                # 4  maxPool3d_3a_3x3
        Mixed(*MIXED_CFG["3b"]), Mixed(*MIXED_CFG["3c"]),   # 5, 6
        MaxPoolTF((3, 3, 3), (2, 2, 2)),                # 7  maxPool3d_4a_3x3
        Mixed(*MIXED_CFG["4b"]), Mixed(*MIXED_CFG["4c"]), Mixed(*MIXED_CFG["4d"]), Mixed(*MIXED_CFG["4e"]),
        Mixed(*MIXED_CFG["4f"]),                        # 8..12
    ]
    base_model = nn.Sequential(*stages)
    if kinetics_pretrain is not None:
        import os
        if not os.path.isfile(kinetics_pretrain):
            raise ValueError("Kinetics_pretrain doesn't exist: {}".format(kinetics_pretrain))
        load_i3d_checkpoint_into_backbone(base_model, torch.load(kinetics_pretrain, map_location="cpu"))
    if freeze_affine:
        freeze_bn_affine(base_model)
    return base_model


I3D_STAGE_NAMES = ["conv3d_1a_7x7", "maxPool3d_2a_3x3", "conv3d_2b_1x1", "conv3d_2c_3x3", "maxPool3d_3a_3x3", "mixed_3b",
                   "mixed_3c", "maxPool3d_4a_3x3", "mixed_4b", "mixed_4c", "mixed_4d", "mixed_4e", "mixed_4f"]


def load_i3d_checkpoint_into_backbone(base_model, sd):
    """A Kinetics I3D checkpoint is keyed by the I3D attribute names (models/i3dpt.py:186-231);
    networks.py:113-132 loads it into the full I3D and then slices 13 stages out."""
    mapped = {}
    for idx, name in enumerate(I3D_STAGE_NAMES):
        pre = name + "."
        for k, v in sd.items():
            if k.startswith(pre):
                mapped["%d.%s" % (idx, k[len(pre):])] = v
    missing = [k for k in base_model.state_dict() if k not in mapped]
    if missing:
        raise RuntimeError("I3D checkpoint is missing keys: %s ..." % missing[:4])
    base_model.load_state_dict(mapped)


def freeze_bn_affine(module):
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            for p in m.parameters():
                p.requires_grad = False


def set_bn_eval(module, half=False):
    for m in module.modules():
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.eval()
            if half:
                m.half()


def weights_init(m):
    """models/networks.py:101-105"""
    if isinstance(m, (nn.Conv2d, nn.Linear, nn.Conv3d)):
        nn.init.xavier_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0.0)


class I3D_head(nn.Module):
    """Weight container of the head's Mixed_5b / 5c pair (models/i3dpt.py:165-173)."""

    def __init__(self):
        super().__init__()
        self.maxPool3d = MaxPoolTF((1, 3, 3), (1, 2, 2))
        self.mixed_5b = Mixed(*MIXED_CFG["5b"])
        self.mixed_5c = Mixed(*MIXED_CFG["5c"])


class I3D(nn.Module):
    """The Kinetics classifier the backbone is cut from (models/i3dpt.py:175-262): same constructor, attribute names /
    state_dict keys and forward(inp [N,3,T,H,W]) -> (softmax, logits).  STEP's scripts only use it as a checkpoint
    container (networks.py:113-132); forward runs the same HIP kernels as BaseNet, then maxPool3d_5a, Mixed_5b / 5c, the
    (2,7,7) average pool (7x7 window kernel + the mean of neighbouring planes), dropout and the biased logits conv."""

    def __init__(self, num_classes, dropout_prob=0, name="inception"):
        super().__init__()
        self.name = name
        self.num_classes = num_classes
        self.conv3d_1a_7x7 = Unit3D(3, 64, (7, 7, 7), (2, 2, 2))
        self.maxPool3d_2a_3x3 = MaxPoolTF((1, 3, 3), (1, 2, 2))
        self.conv3d_2b_1x1 = Unit3D(64, 64, (1, 1, 1))
        self.conv3d_2c_3x3 = Unit3D(64, 192, (3, 3, 3))
        self.maxPool3d_3a_3x3 = MaxPoolTF((1, 3, 3), (1, 2, 2))
        self.mixed_3b = Mixed(*MIXED_CFG["3b"])
        self.mixed_3c = Mixed(*MIXED_CFG["3c"])
        self.maxPool3d_4a_3x3 = MaxPoolTF((3, 3, 3), (2, 2, 2))
        self.mixed_4b = Mixed(*MIXED_CFG["4b"])
        self.mixed_4c = Mixed(*MIXED_CFG["4c"])
        self.mixed_4d = Mixed(*MIXED_CFG["4d"])
        self.mixed_4e = Mixed(*MIXED_CFG["4e"])
        self.mixed_4f = Mixed(*MIXED_CFG["4f"])
        self.maxPool3d_5a_2x2 = MaxPoolTF((2, 2, 2), (2, 2, 2))
        self.mixed_5b = Mixed(*MIXED_CFG["5b"])
        self.mixed_5c = Mixed(*MIXED_CFG["5c"])
        self.avg_pool = nn.AvgPool3d((2, 7, 7), (1, 1, 1))       # (parameter-free holder, as in the reference; forward uses the HIP pool)
        self.dropout = nn.Dropout(dropout_prob)
        self.conv3d_0c_1x1 = Unit3D(1024, num_classes, (1, 1, 1), use_bias=True, use_bn=False, relu=False)
        self.softmax = nn.Softmax(1)

    def forward(self, inp):
        if inp.dim() != 5 or inp.shape[1] != 3:
            raise RuntimeError("I3D expects [batch, 3, T, H, W]")
        x = inp.permute(0, 2, 1, 3, 4).contiguous()             # the stem reads [N,T,3,H,W]
        for name in I3D_STAGE_NAMES:
            x = getattr(self, name)(x)
        x = self.mixed_5c(self.mixed_5b(self.maxPool3d_5a_2x2(x)))
        if x.shape[2] < 7 or x.shape[3] < 7 or x.shape[1] < 2:
            raise RuntimeError("I3D: the (2,7,7) average pool needs at least 2 x 7 x 7 features (T >= 16, H, W >= 224)")
        a = ops.avgpool_hw(x, 7, 7)                              # [N, D, H-6, W-6, 1024]
        a = (a[:, :-1].float() + a[:, 1:].float()).mul_(0.5).to(x.dtype)
        a = self.dropout(a)
        out = self.conv3d_0c_1x1(a)                              # [N, D-1, H-6, W-6, classes]
        out = out.permute(0, 4, 1, 2, 3)                         # logical NCDHW, as the reference's conv output
        out = out.squeeze(3).squeeze(3).float().mean(2)
        return self.softmax(out), out


def as_channels_last_5d(t):
    """logical [N,T,C,H,W] (any strides) -> physical [N,T,H,W,C] contiguous tensor (zero-copy when the
    tensor already is a permuted view of one, e.g. our own BaseNet / ROIAlign outputs)."""
    v = t.permute(0, 1, 3, 4, 2)
    if v.is_contiguous():
        return v
    N, T, C, H, W = t.shape
    if t.is_contiguous() and not (torch.is_grad_enabled() and t.requires_grad):   # (the HIP transpose records no autograd node)
        return ops.to_channels_last(t.reshape(N * T, C, H, W)).view(N, T, H, W, C)
    return v.contiguous()


class BaseNet(nn.Module):
    """Backbone: I3D up to mixed_4f.  forward(x [N,T,3,H,W]) -> conv_feat logical [N,T',832,H',W']
    (a permuted VIEW of the channels-last buffer, like the reference returns a permuted view,
    models/networks.py:69-83)."""

    def __init__(self, cfg):
        super().__init__()
        self.base_name = cfg.base_net
        self.kinetics_pretrain = cfg.kinetics_pretrain
        self.freeze_stats = cfg.freeze_stats
        self.freeze_affine = cfg.freeze_affine
        self.fp16 = cfg.fp16
        if self.base_name != "i3d":
            raise NotImplementedError
        self.base_model = build_base_i3d(self.kinetics_pretrain, self.freeze_affine)

    def forward_u8(self, frames, dtype=torch.bfloat16, scale=2, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
        """forward() on the decoder's uint8 frames [batch, T, H, W, 3] (what a fed node receives over PCIe: 4x fewer bytes than the fp32
        clip of data/ava.py:298-368): the reference's ConvertFromInts(scale) / SubtractMeans / DivideStds (data/augmentations.py:68-111)
        and the rounding to `dtype` happen inside the stem's frame staging where the library has that form (inference, 16-bit, W % 4 == 0),
        else as one conversion pass (ops.clip_from_u8) in front of forward().  Bit-identical to forward(clip_from_u8(frames, dtype))."""
        z = self.stem_u8(frames, dtype, scale, mean, std)
        if z is not None:
            return self.after_stem(z)
        return self.forward(ops.clip_from_u8(frames.contiguous(), dtype, scale, mean, std))

    def stem_u8(self, frames, dtype=torch.bfloat16, scale=2, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
        """The first two stages (conv3d_1a_7x7 + maxPool3d_2a) from uint8 frames: the pooled channels-last tensor [N,To,Hp,Wp,64], or None
        when the library has no such form for the call (then forward_u8 converts first).  A fed loop captures this and after_stem() as two
        graphs: the frames' staging buffer is free again as soon as THIS part has run."""
        if frames.dim() != 5 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise RuntimeError("BaseNet.forward_u8 expects uint8 frames [batch, T, H, W, 3]")
        stages = list(self.base_model)
        if FUSE_STEM_POOL and FUSE_STEM_U8 and len(stages) > 1 and frames.is_cuda and dtype != torch.float32:
            return _stem_then_pool(stages[0], stages[1], frames.contiguous(), u8=(dtype, scale, mean, std))
        return None

    def after_stem(self, z):
        """Stages 2.. on the pooled stem output of stem_u8(); returns what forward() returns."""
        return self._stages_from(z, 2)

    def forward(self, x):
        if x.dim() != 5 or x.shape[2] != 3:
            raise RuntimeError("BaseNet expects [batch, T, 3, H, W]")
        return self._stages_from(x.contiguous(), 0)

    def _stages_from(self, y, i):
        stages = list(self.base_model)
        while i < len(stages):
            st = stages[i]
            if FUSE_STEM_POOL and i == 0 and i + 1 < len(stages):
                z = _stem_then_pool(st, stages[i + 1], y)                # maxPool3d_2a on the stem's tiles
                if z is not None:
                    y = z
                    i += 2
                    continue
            if FUSE_POINTWISE_INPUT and FUSE_CONV_POOL and i + 2 < len(stages):
                z = _pointwise_then_3x3x3(st, stages[i + 1], y, pool=stages[i + 2])    # ... and maxPool3d_3a on conv3d_2c's tiles
                if z is not None:
                    y = z
                    i += 3
                    continue
            if FUSE_POINTWISE_INPUT and i + 1 < len(stages):
                z = _pointwise_then_3x3x3(st, stages[i + 1], y)          # conv3d_2b evaluated inside conv3d_2c's halo staging
                if z is not None:
                    y = z
                    i += 2
                    continue
            y = st(y)
            i += 1
        return y.permute(0, 1, 4, 2, 3)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        if mode and self.freeze_stats:
            set_bn_eval(self.base_model, half=False)   # BN is folded in fp32 whatever the activation dtype
        return self
