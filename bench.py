#!/usr/bin/env python
"""bench.py -- headline benchmark of the STEP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): the I3D backbone (BaseNet, conv3d_1a ... mixed_4f) forward,
bf16 storage / fp32 accumulate, on a batch of 8 synthetic clips [8, 32, 3, 224, 224] per GPU that is
already resident in HBM.  A "step" is one forward over one batch.  With N GPUs every rank processes its
own batch (clips shard by clip, no data-path collective, weak scaling); value = clips of all ranks /
max-over-ranks time.

One JSON line is printed by rank 0: the contract fields plus
  "roofline"     for the dominant kernel of the step (largest share of GPU time), from HIP events
                 recorded around every launch on the launch stream during an instrumented pass;
  "cpu_baseline" the torch-CPU fp32 restatement of the same backbone (oracle/i3d_ref.py) timed on
                 this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CLIPS_PER_GPU = 8
T_IN, HW_IN = 32, 224
# --config c5 (BASELINE configs[4], long-clip stress): T=64, 400x400, fp16, 4 clips per GPU (= batch 32 on 8 GPUs)
CONFIGS = {"c2": dict(clips=8, T=32, HW=224, dtype="bf16", gflop=109.29, act_mb=304.9, name="C2"),
           "c5": dict(clips=4, T=64, HW=400, dtype="f16", gflop=696.98, act_mb=1944.6, name="C5"),
           # full pipelines (step_amd/workloads.py): C3 = 3-step two_branch inference + NMS, C4 = one training step
           "c3": dict(clips=4, T=36, HW=400, dtype="bf16", gflop=392.05 + 14.46 + 158.0, act_mb=1093.9, name="C3"),
           "c4": dict(clips=1, T=36, HW=400, dtype="f32", gflop=3 * (392.05 + 14.46), act_mb=2187.7, name="C4")}
PEAK = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}          # dense MFMA TFLOP/s, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PROFILE_ROUND = "r06"                                           # the profiles/ files of the round this bench.py ships with
# algorithmic work of BaseNet at C2 per clip: 109.29 GFLOP, 304.9 MB activations + 15.0 MB weights/batch
GFLOP_PER_CLIP = 109.29
ACT_MB_PER_CLIP, W_MB = 304.9, 15.0


def cfg():
    from types import SimpleNamespace as NS
    return NS(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False)


def build_net(device, seed=123):
    import step_amd
    torch.manual_seed(seed)                                      # config.py:38 man_seed
    net = step_amd.BaseNet(cfg())
    # random-init weights of the real architecture (no checkpoints offline): He-scaled convs, mild BN stats
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv3d):
                torch.nn.init.kaiming_uniform_(m.weight, a=0.0)
            elif isinstance(m, torch.nn.BatchNorm3d):
                m.weight.uniform_(0.9, 1.1)
                m.bias.uniform_(-0.05, 0.05)
                m.running_mean.uniform_(-0.1, 0.1)
                m.running_var.uniform_(0.8, 1.2)
    return net.to(device).eval()


def cpu_baseline(net, seconds_budget=12.0):
    """The oracle's torch-CPU fp32 BaseNet on single clips [1,32,3,224,224], all host cores."""
    from oracle import i3d_ref as R
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(123)
    x = torch.rand(1, 32, 3, 224, 224, generator=g) * 2 - 1     # the CPU sample is always a C2-shaped clip
    with torch.no_grad():
        best, table = _cpu_threads(lambda: R.basenet_forward(x[:, :8], sd), ncpu)    # probe on a quarter-length clip
        R.basenet_forward(x, sd)                                 # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            R.basenet_forward(x, sd)
            n += 1
            el = time.perf_counter() - t0
            if el > seconds_budget or n >= 64:
                break
    model = _cpu_model()
    # cores = the threads the timed sample actually ran on (the fastest of the probed thread counts); host_cores = what the box has
    return {"value": round(n / el, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "host_cores": ncpu, "cpu_model": model, "kind": "port", "threads_probe_s": table,
            "sample": "%d single-clip [1,32,3,224,224] fp32 forwards of oracle/i3d_ref.basenet_forward (torch CPU) after 1 warm-up, %.1f s" % (n, el)}


def _cpu_threads(probe, ncpu):
    """torch's CPU conv3d does not scale to hundreds of threads: time `probe()` at a few thread counts and keep the fastest."""
    best, best_t, table = None, None, {}
    for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        probe()
        t0 = time.perf_counter()
        probe()
        dt = time.perf_counter() - t0
        table[th] = round(dt, 3)
        if best_t is None or dt < best_t:
            best, best_t = th, dt
        if dt > 6.0:
            break
    torch.set_num_threads(best)
    return best, table


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_pipeline_baseline(config, w, tubes, seconds_budget=15.0):
    """cpu_baseline of --config c3 / c4: the torch-CPU fp32 restatement (oracle/i3d_ref.py; test infrastructure, used here only as the
    timed CPU leg) of the same pipeline on ONE AVA-shaped clip with the workload's weights -- c3: BaseNet + ContextNet + the 3-step
    inference() + the per-class post-processing loop; c4: forward + backward of BaseNet + ContextNet + the three heads' losses under
    torch autograd (no optimizer step: the restatement has none).  Bounded sample, host cores of this box."""
    import numpy as np
    from oracle import i3d_ref as R
    from oracle import postprocess_ref as PR
    ncpu = os.cpu_count() or 1
    f = lambda m: {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    sd_b, sd_c = f(w.base), f(w.ctx)
    nets_sd = {k: f(v) for k, v in w.nets.items() if k.startswith("det_net")}
    g = torch.Generator().manual_seed(123)
    x = torch.rand(1, 36, 3, 400, 400, generator=g) * 2 - 1
    from step_amd.tube_math import generate_anchors
    anchors = generate_anchors()[:tubes] * 400.0
    tl = [np.tile(anchors[:, None, :], (1, 3, 1)).astype(np.float32)]
    if config == "c3":
        def once():
            with torch.no_grad():
                cf = R.basenet_forward(x, sd_b)
                cx = R.contextnet_forward(cf, sd_c)
                hist = R.inference(cf, cx, nets_sd, tl)
                return PR.postprocess(hist)
        what = "BaseNet + ContextNet + 3-step inference() + post-processing on 1 x [36,3,400,400] clip, %d tubes" % tubes
    else:
        K = tubes
        params = []
        for sd in [sd_b, sd_c] + list(nets_sd.values()):
            for k, v in sd.items():
                if v.dtype.is_floating_point and "batch3d" not in k and "running" not in k:
                    v.requires_grad_(True)
                    params.append(v)
        at = torch.from_numpy(anchors.astype(np.float32))
        tgt = torch.zeros(K, 3, 66)
        tgt[:, :, :4] = at.view(K, 1, 4) + 4.0
        tgt[:, :, 4] = 1; tgt[:, :, 5] = 1; tgt[:, :, 6 + 7] = 1
        import oracle

        class RoiAlignCPU(torch.autograd.Function):               # the C restatement's forward / backward pair under torch autograd
            @staticmethod
            def forward(ctx_, feat, rois):
                ctx_.shape, ctx_.rois = tuple(feat.shape), rois.numpy()
                return torch.from_numpy(oracle.roi_align_forward(feat.detach().numpy(), ctx_.rois, (7, 7), 1 / 16., 0))

            @staticmethod
            def backward(ctx_, g_):
                return torch.from_numpy(oracle.roi_align_backward(g_.contiguous().numpy(), ctx_.rois, (7, 7), 1 / 16., 0, ctx_.shape)), None

        def once():
            for p_ in params:
                p_.grad = None
            cf = R.basenet_forward(x, sd_b)
            cx = R.contextnet_forward(cf, sd_c)
            loss = 0.0
            for it in range(3):
                Tl = 3 * w.args.NUM_CHUNKS[it + 1]
                t0 = (9 - Tl) // 2
                fr = (t0 + torch.arange(Tl, dtype=torch.float32)).view(1, Tl, 1).expand(K, Tl, 1)
                flat = torch.cat([fr, at.view(K, 1, 4).expand(K, Tl, 4)], 2).contiguous()
                pooled = RoiAlignCPU.apply(cf.reshape(-1, *cf.shape[2:]).contiguous(), flat.reshape(-1, 5).contiguous())
                pooled = pooled.view(K, Tl, *pooled.shape[1:])
                o = R.twobranch_forward(pooled, cx.expand(K, -1, -1, -1, -1)[:, :, t0:t0 + Tl], nets_sd["det_net%d" % it], tubes=flat, targets=tgt)
                loss = loss + o[4].mean() + 5 * o[5].mean() + o[6].mean()
            loss.backward()
            return float(loss.detach())
        what = "forward + backward (torch autograd) of BaseNet + ContextNet + 3 heads' losses on 1 x [36,3,400,400] clip, %d tubes" % tubes
    small = x[:, :12]
    best, table = _cpu_threads(lambda: _nograd(R.basenet_forward, small, sd_b), ncpu)
    once()                                                       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        once()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 16:
            break
    return {"value": round(n / el, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "host_cores": ncpu, "cpu_model": _cpu_model(),
            "kind": "port", "threads_probe_s": table,
            "sample": "%d x (%s) of oracle/i3d_ref.py (torch CPU fp32) after 1 warm-up, %.1f s" % (n, what, el)}


def _nograd(fn, *a):
    with torch.no_grad():
        return fn(*a)


class ClockSampler:
    """Best-effort shader-clock reading while a loop runs: a thread polls the amdgpu sysfs nodes of the device (hwmon freq1_input in
    Hz, else the starred line of pp_dpm_sclk) every 20 ms.  None when the box exposes neither (nothing else is affected)."""

    def __init__(self, index=0):
        import glob
        self.paths = []
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or glob.glob(os.path.join(c, "hwmon/hwmon*/freq1_input"))]
        if cards:
            c = cards[min(index, len(cards) - 1)]
            self.paths = glob.glob(os.path.join(c, "hwmon/hwmon*/freq1_input")) + [os.path.join(c, "pp_dpm_sclk")]
        self.samples, self.smi_samples, self._stop, self._th = [], [], False, None

    def _smi(self):
        """`amd-smi metric --clock --json` (one call takes ~1 s): mean of the gfx engine clocks it lists, in GHz"""
        import re
        import subprocess
        try:
            txt = subprocess.run(["amd-smi", "metric", "--clock", "--json"], capture_output=True, text=True, timeout=10).stdout
            j = json.loads(txt)
            vals = []

            def walk(o, under_gfx=False):
                if isinstance(o, dict):
                    for k_, v_ in o.items():
                        g_ = under_gfx or k_.lower().startswith("gfx")
                        if g_ and k_ in ("clk", "cur_clk") and isinstance(v_, dict) and isinstance(v_.get("value"), (int, float)):
                            vals.append(float(v_["value"]))
                        else:
                            walk(v_, g_)
                elif isinstance(o, list):
                    for v_ in o:
                        walk(v_, under_gfx)
            walk(j[0] if isinstance(j, list) and j else j)
            vals = [v_ for v_ in vals if v_ > 0]
            return sum(vals) / len(vals) / 1e3 if vals else None
        except Exception:
            return None

    def _read(self):
        for p_ in sorted(self.paths, key=lambda q: not q.endswith("pp_dpm_sclk")):     # the DPM table's starred line first
            try:
                txt = open(p_).read()
            except OSError:
                continue
            if p_.endswith("freq1_input"):
                v = float(txt.strip()) / 1e9
                if v > 0:
                    return v
            else:
                for line in txt.splitlines():
                    if "*" in line:
                        return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()) / 1e3
        return None

    def __enter__(self):
        import threading
        def loop():
            while not self._stop:
                try:
                    v = self._read()
                except Exception:
                    v = None
                if v:
                    self.samples.append(v)
                time.sleep(0.02)
        def smi_loop():
            while not self._stop:
                v = self._smi()
                if v:
                    self.smi_samples.append(v)
                else:
                    break
        if self.paths:
            self._th = threading.Thread(target=loop, daemon=True)
            self._th.start()
        self._th2 = threading.Thread(target=smi_loop, daemon=True)
        self._th2.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=1.0)
        # the amd-smi poller too: its last call (~1 s, a subprocess that queries the driver / SMU) must not run on into whatever is timed
        # next -- round 6: the contract window right behind the sustained loop read 5.5 % under it (7 108 against 7 524 clips/s) with the
        # poller still alive
        if getattr(self, "_th2", None) is not None:
            self._th2.join(timeout=5.0)
        return False

    def median(self):
        s_ = sorted(self.samples)
        return round(s_[len(s_) // 2], 3) if s_ else None

    def smi_median(self):
        s_ = sorted(self.smi_samples)
        return round(s_[len(s_) // 2], 3) if s_ else None


class EffClock:
    """Effective shader clock DURING a loop: step_clock_sample (one wavefront per workgroup that compares s_memtime with the 100 MHz
    s_memrealtime over ~50 us of s_sleep) launched on a high-priority side stream every `every` steps of the loop it accompanies.
    Unlike the DPM state the driver reports (ClockSampler), this is the clock the CUs really ran at while the loop's kernels
    executed around the sampler wave.  median / p5 / p95 over all samples, GHz."""

    def __init__(self, dev, slots=64, wgs=8, ticks=5000, stream=None):
        import ctypes
        from step_amd import _capi, _lib
        self.L, self._capi, self._lib, self.ct = _lib.lib(), _capi, _lib, ctypes
        self.dev, self.slots, self.wgs, self.ticks = dev, slots, wgs, ticks
        self.buf = torch.zeros(slots * wgs * 2, dtype=torch.int64, device=dev)
        # `stream`: an EXISTING stream that is idle or lightly used beside the loop (bench.py hands over the second batch's stream).  Creating
        # streams here is not harmless: HIP maps streams onto a few hardware queues in creation order, and three extra priority streams in
        # front of the fed loop's copy stream made that stream share a queue with a compute stream -- the fed rate fell from ~6.9 k to 4.8 k
        # clips/s (round 6, call c12; the aliasing itself: tools/feed_probe.py, round 5)
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.n = 0

    def sample(self):
        if self.n >= self.slots:
            return
        off = self.n * self.wgs * 2 * 8
        with torch.cuda.stream(self.stream):
            self._capi.check(self.L.step_clock_sample(self.ct.c_void_p(self.buf.data_ptr() + off), self.wgs, self.ticks, self._lib.stream_ptr(self.dev)), "step_clock_sample")
        self.n += 1

    def result(self):
        torch.cuda.synchronize()
        h = self.buf.cpu().numpy().reshape(-1, 2)[: self.n * self.wgs].astype("float64")
        ok = h[:, 1] > 0
        if not ok.any():
            return None
        g = sorted(h[ok, 0] / (h[ok, 1] * 10.0))
        return {"ghz_median": round(g[len(g) // 2], 3), "ghz_p5": round(g[len(g) // 20], 3), "ghz_p95": round(g[-1 - len(g) // 20], 3), "samples": len(g)}


def sustained_mfma(dev):
    """What THIS box sustains with nothing but 16-bit matrix instructions on every CU (step_mfma_clock_probe, include/step_amd.h):
    MI355X is power-managed, so the clock under matrix load is well below the 2.4 GHz the datasheet peak assumes.  Reported beside
    `roofline` (whose `peak` stays the datasheet figure the contract names) so that `frac` can be read against the box at hand."""
    import ctypes
    from step_amd import _capi, _lib
    L = _lib.lib()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    iters = 20000
    buf = torch.zeros(3 * cus, dtype=torch.int64, device=dev)
    for _ in range(40):                                          # ~60 ms of continuous matrix load: the power manager needs tens of ms to settle; the LAST launch is read
        _capi.check(L.step_mfma_clock_probe(ctypes.c_void_p(buf.data_ptr()), cus, iters, _lib.stream_ptr(dev)), "step_mfma_clock_probe")
    torch.cuda.synchronize()
    h = buf.cpu().numpy().reshape(cus, 3).astype("float64")
    ok = h[:, 1] > 0
    if not ok.any():
        return None
    ghz = sorted(h[ok, 0] / (h[ok, 1] * 10.0))
    us = sorted(h[ok, 1] * 0.01)
    clock, t = ghz[len(ghz) // 2], us[len(us) // 2]
    tf = cus * 4 * 4 * 32768.0 * iters / (t * 1e-6) / 1e12      # 4 waves x 4 MFMAs x 32768 FLOP per iteration per workgroup
    return {"clock_ghz_median": round(clock, 3), "clock_ghz_p5_p95": [round(ghz[len(ghz) // 20], 3), round(ghz[-1 - len(ghz) // 20], 3)],
            "dense_bf16_tflops": round(tf, 1), "workgroups": cus,
            "note": "one 256-thread workgroup per CU issuing v_mfma_f32_32x32x16_bf16 back to back on pseudo-random bf16 operands, nothing "
                    "else, read after ~60 ms of that load; clock = s_memtime / s_memrealtime over the loop; the datasheet peak "
                    "(2500 TFLOP/s) assumes 2.4 GHz"}


def sustained_hbm(dev):
    """What THIS box moves with a plain streaming copy (step_hbm_stream_probe): 1 GiB read + 1 GiB written per launch, far beyond the
    256 MB last-level cache; best of 5 launches after a warm-up, HIP events on the launch stream.  The HBM-bound kernels (pools,
    pointwise convs, ROIAlign) are read against this next to the 8 TB/s of the datasheet."""
    from step_amd import _capi, _lib
    import ctypes
    L = _lib.lib()
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255)
    dst = torch.empty_like(src)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    best = None
    for k in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _capi.check(L.step_hbm_stream_probe(ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), nbytes, cus * 16, _lib.stream_ptr(dev)), "step_hbm_stream_probe")
        e1.record()
        torch.cuda.synchronize()
        if k:
            ms = e0.elapsed_time(e1)
            best = ms if best is None else min(best, ms)
    del src, dst
    return {"copy_GBs": round(2 * nbytes / (best * 1e-3) / 1e9, 1), "bytes_read_plus_written": 2 * nbytes,
            "note": "step_hbm_stream_probe: grid-stride 16 B / lane non-temporal copy, 16 workgroups per CU, best of 5"}


def roofline(net, x, dtype_name):
    """Per-launch durations of the REPLAYED step, measured live: HIP graphs of growing prefixes of the forward (launches 1..k, the
    rest skipped by step_amd.ops.PROFILE_LIMIT) are captured and replayed, and launch k's duration is the difference of the best
    replay times of prefix k and prefix k-1 -- the kernel in its real place of the sequence, at replay clocks, its inputs where the
    previous kernel left them (L2 / MALL / HBM), no warm-up twin, no eager launch gaps.  The replayed step is a single-stream chain
    (tools/graph_timeline.py: kernels start back to back), so the differences add up to the step time."""
    from step_amd import ops
    ops.PROFILE, ops.PROFILE_LIMIT = [], 1 << 30
    with torch.no_grad():
        net(x)
    torch.cuda.synchronize()
    plan = ops.PROFILE
    n = len(plan)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = [0.0]
    eff = EffClock(x.device, slots=max(n, 1))
    try:
        for k in range(1, n + 1):
            ops.PROFILE, ops.PROFILE_LIMIT = [], k
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                net(x)
            g.replay()
            torch.cuda.synchronize()
            best = float("inf")
            for r_ in range(4):
                if r_ == 3:
                    eff.sample()                   # (beside the last round of this prefix's replays)
                e0.record()
                for _ in range(5):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            times.append(best)
            del g
    finally:
        ops.PROFILE, ops.PROFILE_LIMIT = None, None
    rec = [(name, flops, nbytes, max(times[k + 1] - times[k], 1e-4)) for k, (name, flops, nbytes, _, _) in enumerate(plan)]
    # cross-check with HIP events (the contract's wording): the same step launched EAGERLY, every instrumented call bracketed by events on its
    # launch stream, median of 7 passes.  Long kernels behind a long predecessor are exact (the host is far ahead); short ones carry the host's
    # record -> launch gap.  Reported per kernel name as `events_avg_launch_ms` beside the prefix-graph figure.
    ev = {}
    try:
        for _ in range(8):
            ops.PROFILE, ops.PROFILE_LIMIT = [], None
            with torch.no_grad():
                net(x)
            torch.cuda.synchronize()
            for k, (name, _, _, q0, q1) in enumerate(ops.PROFILE):
                ev.setdefault(k, []).append(q0.elapsed_time(q1))
    finally:
        ops.PROFILE, ops.PROFILE_LIMIT = None, None
    ev_by_name = {}
    for k, (name, _, _, _, _) in enumerate(plan):
        v = sorted(ev.get(k, [0.0])[1:])
        ev_by_name.setdefault(name, []).append(v[len(v) // 2] if v else 0.0)
    agg = {}
    for name, flops, nbytes, t in rec:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 3                                   # (describe() below counts in units of three passes)
        a[1] += 3 * t
        a[2] += 3 * flops
        a[3] += 3 * nbytes
    total_ms = sum(a[1] for a in agg.values())
    def describe(name, cnt, ms, flops, nbytes):
        avg_ms = ms / cnt
        hbm_time = (nbytes / cnt) / (PEAK_HBM_GBS * 1e9)
        mfma_time = (flops / cnt) / (PEAK[dtype_name] * 1e12)
        traffic, tsrc, covers = None, None, None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if name in tj.get("kernels", {}):
                    traffic = tj["kernels"][name].get("hbm_bytes_per_launch")
                    tsrc = "NOT measured in this run: profiles/traffic_latest.json (%s; taken at commit %s)" % (tj.get("source"), tj.get("commit", "?"))
                    # a one-group conv_tap layer is launched in two parts (full rounds at NB = 3, the partial last round at NB = 1,
                    # DESIGN.md 3.1): the timed call and its algorithmic bytes cover both, so does the traffic
                    tail = name.replace(", 3, 3, 3, 3, 2, 2, 8, 1>", ", 1, 3, 3, 3, 2, 2, 8, 1>") if ", 3, 3, 3, 3, 2, 2, 8, 1>" in name else None
                    if ("conv_tap_pre_kernel<" in name or "conv_tap_pre_pool_kernel<" in name or "conv_tap_pre_pool_persist_kernel<" in name) and name.endswith(", 3>(step::ConvParams)"):      # (the forms with conv3d_2b fused in / and maxPool3d_3a)
                        tail = name.replace(", 3>(step::ConvParams)", ", 1>(step::ConvParams)").replace("_persist_kernel", "_kernel")
                    if tail and tail != name and tail in tj["kernels"] and tj["kernels"][tail].get("with") == name:
                        traffic += tj["kernels"][tail].get("hbm_bytes_per_launch", 0)
                        tsrc += "; + the NB = 1 launch of the layer's last partial round"
                        covers = [name, tail]
            except Exception:
                pass
        if mfma_time >= hbm_time:      # matrix-bound kernel: algorithmic FLOP/s against the dense MFMA peak
            ach = (flops / cnt) / (avg_ms * 1e-3) / 1e12
            out = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK[dtype_name], "unit": "TFLOP/s",
                   "frac": round(ach / PEAK[dtype_name], 4)}
        else:                          # streaming kernel: algorithmic bytes/s against the HBM peak
            ach = (nbytes / cnt) / (avg_ms * 1e-3) / 1e9
            out = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": round(ach / PEAK_HBM_GBS, 4)}
        evl = ev_by_name.get(name)
        out.update({"traffic": traffic, "launches_per_step": cnt // 3, "avg_launch_ms": round(avg_ms, 4),
                    "events_avg_launch_ms": round(sum(evl) / len(evl), 4) if evl else None,
                    "share_of_gpu_time": round(ms / total_ms, 3),
                    "algorithmic_gflop_per_launch": round(flops / cnt / 1e9, 3),
                    "algorithmic_mb_per_launch": round(nbytes / cnt / 1e6, 3)})
        if tsrc:
            out["traffic_from"] = tsrc
        if covers:
            out["call_covers"] = {"launches": covers, "note": "avg_launch_ms / achieved are those of the layer CALL = both launches back to back (in a "
                                  "kernel trace: the sum of the two kernels' average durations)"}
        return out

    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = describe(ranked[0][0], *ranked[0][1])
    # the classes behind the dominant one (conv3d_2c and the stem are within a few per cent of each other: which of them leads
    # changes with the box), same accounting
    out["next_kernels"] = [{k_: v_ for k_, v_ in describe(n_, *a_).items() if k_ != "traffic_from"} for n_, a_ in ranked[1:3]]
    out["method"] = ("durations = differences of the best replay times of HIP graphs of growing prefixes of the step (each launch in its place of "
                     "the replayed sequence, no warm-up twin); 'dominant' = the kernel name with the largest summed time over the step")
    out["profile"] = ("rocprofv3 --kernel-trace --stats of `bench.py --in-flight 1`: profiles/%s_c2_kernel_stats.txt (one batch at a time, as these "
                      "durations); of the default command: profiles/%s_c2_kernel_stats_two_in_flight.txt -- there launches of the two batches in "
                      "flight share the CUs, so a trace's per-launch durations are longer than the kernels' own cost while the step is shorter" % (PROFILE_ROUND, PROFILE_ROUND))
    out["clock_during_prefix_replays"] = eff.result()
    table = sorted(((n_, a[1] / 3, a[0] // 3, a[2] / max(a[1], 1e-9) / 1e9) for n_, a in agg.items()), key=lambda r: -r[1])   # TFLOP/s = flops / ms / 1e9
    # the step launch by launch, in launch order (the driver's record must be able to say WHICH launch is slow on its box)
    seq = []
    for name, flops, nbytes, t in rec:
        short = name.split("(")[0].replace("step::", "")
        row = {"kernel": short, "us": round(t * 1e3, 1)}
        if flops / (PEAK[dtype_name] * 1e12) >= nbytes / (PEAK_HBM_GBS * 1e9):
            row["tflops"] = round(flops / (t * 1e-3) / 1e12, 1)
        else:
            row["gbs"] = round(nbytes / (t * 1e-3) / 1e9, 1)
        seq.append(row)
    return out, table, total_ms / 3, seq


def fed_loop(a, net, flights, x, dev, tdt, dist, resident_s_per_step):
    """--feed u8: the loop a fed node runs (reference: data/ava.py:298-368 hands [T,3,H,W] fp32 frames to the DataLoader every iteration,
    data/augmentations.py:68-84 converts from uint8 on the HOST; SURVEY 8e names input feeding as the scaling limiter).  Here the wire
    format is uint8 [N,T,H,W,3] (4x fewer PCIe bytes than fp32): per batch in flight a PINNED host buffer, a device staging buffer and one
    hipMemcpyAsync on a high-priority COPY stream; then either
      "u8_stem"    the backbone reads the staged uint8 frames itself (BaseNet.stem_u8 / after_stem, step_stem_pool_forward_u8: the
                   normalisation happens in the stem's frame staging) -- two captured graphs per batch, stem | rest, so that the copy of
                   batch k + nfl may start as soon as batch k's STEM has run; or
      "convert"    step_clip_from_u8 (scale 2: x*2/255-1, the reference's ConvertFromInts) writes the captured step's static 16-bit input,
                   then the resident loop's captured step; copy k + nfl waits for conversion k.
    Both are timed; `value` of the block is the faster one (the product form).  The host buffers hold synthetic frames that are not
    rewritten between steps (there is no decoder on this path); everything behind them is what a fed loop does."""
    from step_amd import ops
    nfl = len(flights)
    N, T, _, H, W = x.shape
    g = torch.Generator().manual_seed(777)
    hosts = [torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(nfl)]
    stages = [torch.empty((N, T, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(nfl)]
    xs = [x if i == 0 else flights[i][2] for i in range(nfl)]      # the resident graphs' static inputs
    graphs = [fl[1] for fl in flights]
    # every batch in flight on a stream of its own, the copies on a HIGH-PRIORITY stream: HIP maps streams onto a handful of hardware
    # queues, and a copy stream that shares a queue with a compute stream waits behind that stream's whole captured step (measured,
    # tools/feed_probe.py: 4.5 k clips/s fed with an aliased copy stream, 6.8 k with streams of their own, 7.2 k resident)
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    copy_stream = torch.cuda.Stream(priority=int(os.environ.get("STEP_FEED_PRIO", "-1")))   # (STEP_FEED_PRIO: A/B aid of tools/r06_fed_call.sh)
    ev_copied = [torch.cuda.Event() for _ in range(nfl)]
    ev_free = [torch.cuda.Event() for _ in range(nfl)]             # the staging buffer may be overwritten
    nbytes = hosts[0].numel()
    # the uint8-stem form: two graphs per batch in flight
    u8g = []
    from step_amd import backbone as _bb
    keep_u8, _bb.FUSE_STEM_U8 = _bb.FUSE_STEM_U8, True            # (opt-in form of the library; measured here beside the default)
    with torch.no_grad():
        for i in range(nfl):
            with torch.cuda.stream(streams[i]):
                z = net.stem_u8(stages[i], tdt)
                if z is None:
                    u8g = None
                    break
                net.after_stem(z)
                torch.cuda.synchronize()
                ga = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=streams[i]):
                    z = net.stem_u8(stages[i], tdt)
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, stream=streams[i]):
                    y = net.after_stem(z)
                u8g.append((ga, gb, z, y))
    _bb.FUSE_STEM_U8 = keep_u8
    torch.cuda.synchronize()

    def copy_in(i):
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ev_free[i])
            stages[i].copy_(hosts[i], non_blocking=True)
            ev_copied[i].record(copy_stream)

    def step_convert(k):
        i = k % nfl
        copy_in(i)
        with torch.cuda.stream(streams[i]):
            streams[i].wait_event(ev_copied[i])
            ops.clip_from_u8(stages[i], scale=2, out=xs[i])
            ev_free[i].record(streams[i])
            graphs[i].replay()

    def step_u8(k):
        i = k % nfl
        copy_in(i)
        with torch.cuda.stream(streams[i]):
            streams[i].wait_event(ev_copied[i])
            u8g[i][0].replay()
            ev_free[i].record(streams[i])
            u8g[i][1].replay()

    def copies_only(k):
        i = k % nfl
        with torch.cuda.stream(copy_stream):
            stages[i].copy_(hosts[i], non_blocking=True)

    def run(fn, steps, warm):
        for i in range(nfl):                                    # events start signalled
            ev_free[i].record(streams[i])
        torch.cuda.synchronize()
        for k in range(warm):
            fn(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    steps = max(a.steps, int(1.0 / max(resident_s_per_step, 1e-5)) + 1)       # ~1 s per fed loop
    el_conv = run(step_convert, steps, max(a.warmup, 2 * nfl))
    el_u8 = run(step_u8, steps, max(a.warmup, 2 * nfl)) if u8g else None
    el_c = run(copies_only, steps, 4)
    if dist is not None:
        t = torch.tensor([el_conv, el_u8 if el_u8 is not None else 0.0, el_c], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el_conv, el_c = float(t[0].item()), float(t[2].item())
        el_u8 = float(t[1].item()) if el_u8 is not None else None
    world = dist.get_world_size() if dist is not None else 1
    rate = lambda el: world * N * steps / el
    form = "u8_stem" if (el_u8 is not None and el_u8 <= el_conv) else "convert"
    el_f = el_u8 if form == "u8_stem" else el_conv
    fed_rate = rate(el_f)
    res_rate = world * N / resident_s_per_step
    h2d = nbytes * steps / el_c / 1e9
    need = nbytes / N * (res_rate / world) / 1e9              # GB/s of uint8 frames one rank consumes at the resident rate
    if fed_rate >= 0.97 * res_rate:
        limit = "compute: the fed loop runs at the resident-input rate (transfers hide under the other batch's compute)"
    elif h2d < 1.1 * need:
        limit = "the host->device link: the copy-only loop moves %.1f GB/s per rank, the resident rate would consume %.1f GB/s" % (h2d, need)
    else:
        limit = ("compute + what feeding adds to a step: the link has room (copy-only %.1f GB/s against the %.1f GB/s the fed rate consumes); the fed step carries the "
                 "copy's HBM writes under the backbone and %s" % (h2d, nbytes / N * (fed_rate / world) / 1e9,
                 "the table look-ups of the uint8 frame staging in the stem" if form == "u8_stem" else "step_clip_from_u8 on the compute stream (reads the uint8 batch, writes the 16-bit clip)"))
    return {"value": round(fed_rate, 2), "unit": "clips/s", "ms_per_step": round(el_f / steps * 1e3, 4), "steps": steps, "form": form,
            "forms": {"u8_stem": round(rate(el_u8), 2) if el_u8 is not None else None, "convert": round(rate(el_conv), 2)},
            "resident_value": round(res_rate, 2), "fed_over_resident": round(fed_rate / res_rate, 4),
            "wire_format": "uint8 [N,T,H,W,3], %.2f MB per clip (fp32 [T,3,H,W] as the reference feeds it: %.2f MB)" % (nbytes / N / 1e6, 4 * nbytes / N / 1e6),
            "h2d_GBs_in_fed_loop": round(nbytes * steps / el_f / 1e9, 2), "h2d_GBs_copy_only": round(h2d, 2),
            "u8_GBs_needed_at_resident_rate": round(need, 2), "bottleneck": limit,
            "eight_ranks": "8 ranks at this rate pull %.0f GB/s of uint8 frames from the host (fp32 frames: %.0f GB/s)" % (8 * nbytes / N * (fed_rate / world) / 1e9, 32 * nbytes / N * (fed_rate / world) / 1e9),
            "note": "pinned host buffers (synthetic frames, not rewritten between steps) -> hipMemcpyAsync on a high-priority copy stream -> u8_stem: the stem stages the uint8 "
                    "frames itself (two captured graphs per batch: stem | rest; copy k+%d waits for stem k) | convert: step_clip_from_u8 (x*2/255-1) into the captured step's input "
                    "(copy k+%d waits for conversion k); %d batches in flight, each on a stream of its own" % (nfl, nfl, nfl)}


def fp16_leg(a, net, x, dev, nfl, thr_profile, streams=None):
    """The C2 loop with fp16 storage: BASELINE names bf16, so bf16 stays the headline, but bf16 storage costs one
    8-bit-mantissa rounding per layer (6.7e-3 against the oracle over the 45 layers at full size) while fp16 -- same MFMA rate, same
    bytes -- is inside north_star's 1e-3 (tests/module_cases.py case_c2_full_size_properties asserts < 1e-3 against the oracle's full tensor).
    Same captured-step loop as `value`: nfl batches in flight, ~0.5 s of the loop first, then the contract's K steps; and the same K steps one at a
    time.  rel_err_vs_fp32 is measured here on the batch's own input: max |y16 - y32| / max |y32| against this library's fp32 path (itself
    2e-6 from the oracle at this size)."""
    from step_amd import _capi as _cp, _lib as _lb
    x16 = x.to(torch.float16)
    y32 = net(x.float())
    y16 = net(x16)
    torch.cuda.synchronize()
    rel = float((y16.float() - y32).abs().max() / y32.abs().max())
    relb = float((net(x).float() - y32).abs().max() / y32.abs().max())
    del y32
    g = torch.Generator(device="cpu").manual_seed(4242)
    fl = []
    with _cp.options(_lb.lib(), **({"throughput": 1} if thr_profile else {})):
        for i in range(nfl):
            xi = x16 if i == 0 else (torch.rand(x.shape, generator=g) * 2 - 1).to(dev).to(torch.float16)
            # (the bf16 loop's own streams: a NEW stream here, behind the three the fed loop created, shared a hardware queue with the default
            # stream on two leases -- the fp16 leg's two batches overlapped half as well as the bf16 ones, 6.8 k against 7.3 k clips/s)
            si = torch.cuda.current_stream() if i == 0 else (streams[i] if streams is not None and i < len(streams) else torch.cuda.Stream())
            with torch.cuda.stream(si):
                for _ in range(2):
                    net(xi)
                torch.cuda.synchronize()
            gi = torch.cuda.CUDAGraph()
            if i == 0:                                               # (captured on the capture's own side stream, replayed on the default stream)
                with torch.cuda.graph(gi):
                    yi = net(xi)
            else:
                with torch.cuda.stream(si), torch.cuda.graph(gi, stream=si):
                    yi = net(xi)
            fl.append((si, gi, xi, yi))
    g1 = torch.cuda.CUDAGraph()                                    # default planner profile, one batch at a time
    with torch.cuda.graph(g1):
        y1 = net(x16)
    torch.cuda.synchronize()

    def loop(steps, n):
        for k in range(steps):
            if n == 1:
                g1.replay()
            else:
                f = fl[k % n]
                with torch.cuda.stream(f[0]):
                    f[1].replay()

    def timed(steps, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(steps, n)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    pilot = timed(10, nfl) / 10
    timed(int(0.5 / max(pilot, 1e-5)) + 1, nfl)
    el = timed(a.steps, nfl)
    timed(int(0.3 / max(pilot, 1e-5)) + 1, 1)
    el1 = timed(a.steps, 1)
    n = x.shape[0]
    return {"value": round(n * a.steps / el, 2), "unit": "clips/s", "ms_per_step": round(el / a.steps * 1e3, 4), "batches_in_flight": nfl,
            "one_batch_in_flight": {"value": round(n * a.steps / el1, 2), "ms_per_step": round(el1 / a.steps * 1e3, 4)},
            "rel_err_vs_fp32": float("%.3e" % rel), "bf16_rel_err_vs_fp32": float("%.3e" % relb),
            "rel_err_vs_oracle": "asserted < 1e-3 on the oracle's full [1,8,832,14,14] tensor (golden weights and clip) in tests/module_cases.py case_c2_full_size_properties (measured 9.3e-4; bf16 6.7e-3)",
            "note": "the same loop as `value` with fp16 storage (fp32 accumulate): same MFMA rate and bytes as bf16, ~8x closer to fp32 (one 11-bit instead of one 8-bit mantissa rounding per "
                    "layer); rel_err_vs_fp32 = max |y16 - y32| / max |y32| on THIS run's random weights and clips (1.1-1.2e-3 measured: at north_star's 1e-3 bar, which the golden "
                    "configuration meets against the oracle); bf16 stays the headline because BASELINE names it"}


def spawn_ranks(n):
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS),
                    help="c2 = headline backbone forward (default); c5 = long-clip stress; c3 = full inference; c4 = training step")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--branch-streams", type=int, default=None, choices=[0, 1, 2],
                    help="tuning aid: side streams of the Inception blocks (step_amd.backbone.BRANCH_STREAMS; default: the module's)")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="c2 / c5: independent batches kept in flight on separate HIP streams (each step is still one batch; default 2)")
    ap.add_argument("--clips", type=int, default=None, help="clips per GPU (default: the config's: c2 8, c5 4, c3 4, c4 1)")
    ap.add_argument("--tubes", type=int, default=None, help="c3 / c4: tubes per clip (default: c3 11, c4 5)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="c2 / c5: length of the second, sustained loop of the same captured step reported under 'sustained' (0 = skip)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="c4 on ONE GPU: initialise a one-rank RCCL process group and issue the bucketed gradient all-reduces all the same "
                         "(identities there) -- the multi-rank program, captured in the step's HIP graph, on a one-GPU box")
    ap.add_argument("--c4-graph", default="auto", choices=["auto", "one", "split"],
                    help="c4: how the step is captured (workloads.C4TrainStep.capture): one graph incl. the RCCL collectives, or fwd/bwd + eager exchange + update")
    ap.add_argument("--select", action="store_true",
                    help="c4: the reference's WHOLE iteration (train.py:257-348): no-grad inference + train_select between the steps (workloads.C4SelectTrainStep)")
    ap.add_argument("--feed", default=None, choices=["none", "u8"],
                    help="c2 / c5: additionally time the FED loop -- pinned host uint8 frames -> H2D on a copy stream -> step_clip_from_u8 -> the "
                         "captured step, double-buffered against the batches in flight; reported under 'fed' (value stays the resident-input loop)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning aid: a planner option of the library (include/step_amd.h step_set_option), e.g. conv_group_pw=0; repeatable")
    ap.add_argument("--no-fp16-leg", action="store_true", help="c2: skip the fp16 leg (the same loop with fp16 storage, reported under 'fp16')")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    global CLIPS_PER_GPU, T_IN, HW_IN, GFLOP_PER_CLIP, ACT_MB_PER_CLIP
    c = CONFIGS[a.config]
    CLIPS_PER_GPU, T_IN, HW_IN, GFLOP_PER_CLIP, ACT_MB_PER_CLIP = a.clips or c["clips"], c["T"], c["HW"], c["gflop"], c["act_mb"]
    a.dtype = a.dtype or c["dtype"]
    if a.feed is None:
        # the headline config also times the FED loop by default on ONE GPU (the driver's command has no --feed); with several
        # ranks it stays opt-in (`--feed u8`): the scaling run's line must not depend on a loop that has never run on more than one device
        a.feed = "u8" if (a.config == "c2" and a.gpus == 1) else "none"

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run (the form the
        # driver uses itself), same arguments; the ranks rendezvous over 127.0.0.1 and rank 0 prints the JSON line
        return spawn_ranks(a.gpus)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU fallback)")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not os.environ.get("STEP_BENCH_SHARE_GPU"):
        raise SystemExit("bench.py: local rank %d but only %d visible GPU(s)" % (local_rank, ndev))
    local_dev = local_rank % ndev                                # STEP_BENCH_SHARE_GPU=1: functional test of the N > 1 path on one GPU
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STEP_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for the shared-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif a.force_exchange:
        import socket
        import torch.distributed as dist
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)

    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    if a.branch_streams is not None:
        from step_amd import backbone as _bb
        _bb.BRANCH_STREAMS = a.branch_streams
    for kv in a.opt:
        from step_amd import _capi, _lib
        k_, v_ = kv.split("=")
        _capi.set_option(_lib.lib(), k_, int(v_))
    if a.config in ("c3", "c4"):
        return pipeline_bench(a, c, dev, tdt, rank, world, dist)
    net = build_net(dev)
    g = torch.Generator(device="cpu").manual_seed(123 + rank)
    x = (torch.rand(CLIPS_PER_GPU, T_IN, 3, HW_IN, HW_IN, generator=g) * 2 - 1).to(dev).to(tdt)   # U(-1,1), resident in HBM

    with torch.no_grad():
        y = net(x)                                               # packs weights, warms the allocator
        torch.cuda.synchronize()
        assert tuple(y.shape) == (CLIPS_PER_GPU, T_IN // 4, 832, -(-HW_IN // 16), -(-HW_IN // 16)) and bool(torch.isfinite(y.float()).all())
        graph = None
        thr_profile = False
        flights = []                                             # (stream, graph) per batch in flight
        if not a.no_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    net(x)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()                       # a hipGraph of the whole forward
            with torch.cuda.graph(graph):
                y = net(x)
            # Batches are independent, so a serving loop keeps TWO in flight: every step is still one full pass over one batch of
            # CLIPS_PER_GPU clips, but step k + 1 (its own input / activation buffers, its own captured graph) is replayed on a second
            # stream while step k drains -- the last launches of a step leave most CUs idle (one-round 3x3x3 grids with 40-57 us
            # workgroups on the 14x14 maps) and the next step's stem fills them.  Same kernels, same results, +12 % clips/s
            # (tools/two_in_flight.py, round 3: three in flight +13 %, four +10 %; re-measured in round 6 on the final kernels, same box, alternating:
            # two 7 539 / 7 582 clips/s, three 7 184 / 7 096 -- a third batch now only adds contention).  --in-flight 1 times the one-batch-at-a-time
            # loop; the JSON line carries both.
            torch.cuda.synchronize()
            # The steps of the multi-batch loop are captured under the library's THROUGHPUT profile (planner option `throughput`, include/step_amd.h:
            # launch shapes for several independent batches in flight -- a block's pointwise conv on its own instead of inside the 3x3x3
            # members' grid; bit-identical results, +1.3 % at two in flight, -2.2 % one batch at a time, tools/flag_ab.py); the one-batch
            # loop (`one_batch_in_flight`) and the roofline's per-kernel durations keep the default (latency) profile.
            from step_amd import _capi as _cp, _lib as _lb
            many = max(1, a.in_flight) > 1
            forced = [kv for kv in a.opt if kv.startswith("throughput=")]
            prof_opts = {"throughput": 1} if (many and not forced) else {}
            thr_profile = bool(prof_opts) or (bool(forced) and forced[-1].endswith("=1"))
            with _cp.options(_lb.lib(), **prof_opts):
                if many:
                    # (batch 0 stays on the DEFAULT stream, the others on pool streams: measured -- with batch 0 on a pool stream of its own
                    # two batches in flight ran NO faster than one, 6.42 k against 6.57 k clips/s on one box; HIP maps streams onto a few
                    # hardware queues and two pool streams may share one.  Default + pool streams has overlapped on every box since round 3.)
                    g0 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g0):
                        y0 = net(x)
                    flights.append((torch.cuda.current_stream(), g0, x, y0))
                else:
                    flights.append((torch.cuda.current_stream(), graph))
                for i in range(1, max(1, a.in_flight)):
                    xi = (torch.rand(CLIPS_PER_GPU, T_IN, 3, HW_IN, HW_IN, generator=g) * 2 - 1).to(dev).to(tdt)
                    si = torch.cuda.Stream()
                    with torch.cuda.stream(si):
                        for _ in range(2):
                            net(xi)
                        torch.cuda.synchronize()
                        gi = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gi, stream=si):
                            yi = net(xi)
                    flights.append((si, gi, xi, yi))
            torch.cuda.synchronize()
            if many:
                graph.replay(); flights[0][1].replay()
                torch.cuda.synchronize()
                assert torch.equal(y, flights[0][3]), "the throughput profile changed the result"

        _timed_hook = [None]

        def timed(nfl):
            """W warm-up steps, then K steps between barrier + synchronize on both sides; nfl batches in flight (round-robin)."""
            def step(k):
                if _timed_hook[0] is not None:
                    _timed_hook[0](k)
                if graph is None:
                    net(x)
                elif nfl == 1:
                    graph.replay()
                else:
                    fl = flights[k % nfl]
                    with torch.cuda.stream(fl[0]):
                        fl[1].replay()
            for k in range(a.warmup):
                step(k)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(a.steps):
                step(k)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        nfl = len(flights) if graph is not None else 1
        # The SUSTAINED loop runs FIRST: the contract's K steps can be a few tens of milliseconds, shorter than the
        # power manager's settling time -- on the driver's box the 24 ms window read 8 % under the same captured steps replayed for 2 s
        # right after it (the window sat on the clock ramp).  So: >= --sustained-seconds of the same loop first (reported under
        # 'sustained', and it leaves the DPM state where a serving loop keeps it), THEN the contract's W warm-up + K timed steps
        # (`value`), then the same K steps one batch at a time.
        sus = None
        if a.sustained_seconds > 0:
            keep_steps, keep_warm = a.steps, a.warmup
            a.steps, a.warmup = 10, 5
            pilot = timed(nfl) / 10                                  # s per step, to size the loop
            if dist is not None:
                tp = torch.tensor([pilot], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(tp, op=dist.ReduceOp.MAX)
                pilot = float(tp.item())
            a.steps, a.warmup = max(keep_steps, int(a.sustained_seconds * 1.1 / pilot) + 1), 0
            ec_s = EffClock(dev, stream=flights[1][0] if nfl > 1 else None)
            every_s = max(1, a.steps // ec_s.slots)
            _timed_hook[0] = lambda k: ec_s.sample() if k % every_s == 0 else None
            with ClockSampler(local_dev) as cs:
                el_s = timed(nfl)
            _timed_hook[0] = None
            sus = (a.steps, el_s, cs.median(), len(cs.samples), cs.smi_median(), len(cs.smi_samples), ec_s.result())
            # the samplers are joined (a host-side pause of up to a few seconds): ~0.3 s of the loop again, untimed, so that the contract's
            # W + K steps start from the state the sustained loop left, not from an idle chip
            a.steps, a.warmup = int(0.3 / pilot) + 1, 0
            timed(nfl)
            a.steps, a.warmup = keep_steps, keep_warm
        el = timed(nfl)
        # the one-batch-at-a-time window: first ~0.5 s of that loop (the DPM state after two batches in flight is not the state a
        # one-batch loop settles in; the driver's round-5 box read 1.45 ms here against 1.23 ms on every other box),
        # with the EFFECTIVE clock sampled beside it, then the same K steps
        one_clock = None
        if nfl > 1:
            keep_steps, keep_warm = a.steps, a.warmup
            a.steps, a.warmup = max(keep_steps, int(0.5 / max(el / keep_steps, 1e-5)) + 1), 0
            ec = EffClock(dev, stream=flights[1][0])                # (the second batch's stream: idle in this loop)
            every = max(1, a.steps // ec.slots)
            _timed_hook[0] = lambda k: ec.sample() if k % every == 0 else None
            timed(1)
            _timed_hook[0] = None
            one_clock = ec.result()
            a.steps, a.warmup = keep_steps, keep_warm
        el_one = timed(1) if nfl > 1 else el                   # the same K steps one batch at a time (reported beside the headline)
        # (the fp16 leg first: it runs on the bf16 loop's own streams; behind the fed loop -- which creates three streams of its own -- its two
        # batches overlapped half as well, 6.75 k against 7.26 k clips/s)
        fp16 = fp16_leg(a, net, x, dev, nfl, thr_profile, streams=[fl[0] for fl in flights]) if (a.config == "c2" and a.dtype == "bf16" and graph is not None and world == 1 and not a.no_fp16_leg) else None
        fed = fed_loop(a, net, flights, x, dev, tdt, dist, el / a.steps) if (a.feed == "u8" and graph is not None) else None
    if dist is not None:
        t = torch.tensor([el, el_one, sus[1] if sus else 0.0], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, el_one = float(t[0].item()), float(t[1].item())
        if sus:
            sus = (sus[0], float(t[2].item())) + sus[2:]

    out = None
    if rank == 0:
        clips = world * CLIPS_PER_GPU * a.steps
        val = clips / el
        out = {"metric": "clips_per_sec_T%d_%d" % (T_IN, HW_IN), "value": round(val, 2), "unit": "clips/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": "%s: I3D backbone (BaseNet conv3d_1a..mixed_4f) forward, %d x [3,%d,%d,%d] clips per GPU, "
                                      "inputs resident in HBM, random-init weights" % (c["name"], CLIPS_PER_GPU, T_IN, HW_IN, HW_IN),
                          "clips_per_gpu": CLIPS_PER_GPU, "T": T_IN, "HW": HW_IN, "parallelism": "clip-sharded replicas x%d (no data-path collective)" % world,
                          "launch": "eager" if graph is None else ("hipGraph replay" if nfl == 1 else
                                                                    "hipGraph replay, %d batches in flight (one captured step per batch, %d HIP streams, round-robin%s)" % (
                                                                        nfl, nfl, "; steps captured under the library's `throughput` planner profile, bit-identical to the default" if thr_profile else "")),
                          "batches_in_flight": nfl},
               "one_batch_in_flight": {"value": round(clips / el_one, 2), "ms_per_step": round(el_one / a.steps * 1e3, 4), "clock_effective": one_clock,
                                       "note": "the same K steps replayed one after the other on one stream (the loop of rounds 1-2), after ~0.5 s of that same loop; "
                                               "clock_effective = step_clock_sample beside that 0.5 s loop (s_memtime / s_memrealtime, GHz)"},
               "planner_profile": {"value": "throughput" if thr_profile else "default", "one_batch_in_flight": "default", "roofline": "default",
                                   "note": "planner option `throughput` (include/step_amd.h) under which the steps of each loop were captured; results bit-identical (asserted in this run)"},
               "ranks": {"world_size": world, "backend": (dist.get_backend() + " (RCCL over xGMI)" if dist.get_backend() == "nccl" else dist.get_backend()) if dist is not None else None,
                         "devices_visible": ndev}}
        if sus:
            out["sustained"] = {"seconds": round(sus[1], 3), "steps": sus[0], "value": round(world * CLIPS_PER_GPU * sus[0] / sus[1], 2),
                                "ms_per_step": round(sus[1] / sus[0] * 1e3, 4), "clock_ghz": sus[4] if sus[4] else sus[2], "clock_samples": sus[5] if sus[4] else sus[3],
                                "clock_ghz_sysfs": sus[2], "clock_ghz_amd_smi": sus[4], "clock_effective": sus[6],
                                "note": "the same loop (same captured steps, same batches in flight) run for >= %.1f s right BEFORE the timed K steps (it also settles the DPM state the contract window then starts from); "
                                        "clock_ghz = median gfx clock during it AS THE DRIVER REPORTS IT (`amd-smi metric --clock`, polled back to back; failing "
                                        "that the amdgpu sysfs node) -- the DPM state, not the effective clock: inside these kernels s_memtime / s_memrealtime "
                                        "measure 1.3-1.6 GHz (tools/timeline_probe.py) and roofline.sustained_on_this_box gives the rate under bare MFMA load; "
                                        "null: not exposed on this box; `value` / `ms_per_step` above stay on the contract's K steps" % a.sustained_seconds}
        if fed is not None:
            out["fed"] = fed
        if fp16 is not None:
            out["fp16"] = fp16
        per_gpu = val / world
        out["backbone_roofline"] = {
            "hbm_frac": round(per_gpu * (ACT_MB_PER_CLIP + W_MB / CLIPS_PER_GPU) * 1e6 / (PEAK_HBM_GBS * 1e9), 4),
            "mfma_frac": round(per_gpu * GFLOP_PER_CLIP * 1e9 / (PEAK[a.dtype] * 1e12), 4),
            "note": "whole-backbone algorithmic bytes (BASELINE.md sec. 2: %.1f MB/clip + 15 MB weights/batch) and FLOPs (%.2f GFLOP/clip) " % (ACT_MB_PER_CLIP, GFLOP_PER_CLIP) +
                    "per second per GPU over the 8 TB/s HBM and dense MFMA peaks"}
    # roofline of the dominant kernel (every rank could, rank 0 reports)
    if rank == 0:
        with torch.no_grad():
            rl, table, gpu_ms, seq = roofline(net, x, a.dtype)
        out["roofline"] = rl
        out["kernel_table"] = seq
        out["clock_ghz"] = {"two_in_flight_sustained": (out.get("sustained") or {}).get("clock_effective"), "one_batch_loop": one_clock,
                            "prefix_graph_replays": rl.pop("clock_during_prefix_replays", None),
                            "note": "EFFECTIVE shader clock (step_clock_sample: s_memtime / s_memrealtime of a sampler wave beside the loop), not the DPM state"}
        if a.dtype != "f32":
            sm = sustained_mfma(dev)
            if sm:
                rl["sustained_on_this_box"] = dict(sm, frac_of_sustained=round(rl["achieved"] / sm["dense_bf16_tflops"], 4) if rl.get("bound") == "mfma" else None)
        sh = sustained_hbm(dev)
        rl.setdefault("sustained_on_this_box", {})["hbm"] = dict(sh, frac_of_sustained=round(rl["achieved"] / sh["copy_GBs"], 4) if rl.get("bound") == "hbm" else None)
        if "backbone_roofline" in out:
            out["backbone_roofline"]["hbm_frac_of_sustained_copy"] = round(out["backbone_roofline"]["hbm_frac"] * PEAK_HBM_GBS / sh["copy_GBs"], 4)
        out["kernel_time_ms_per_step"] = round(gpu_ms, 4)
        if out["config"].get("batches_in_flight", 1) > 1:
            out["kernel_time_note"] = ("sum of the per-launch durations of ONE step replayed alone (= one_batch_in_flight.ms_per_step); with %d batches "
                                       "in flight consecutive steps overlap, so ms_per_step = elapsed / steps is shorter than this sum" % out["config"]["batches_in_flight"])
        if a.verbose:
            for n_, ms, cnt, gfs in table:
                print("%9.4f ms %3d x  %8.1f TFLOP/s  %s" % (ms, cnt, gfs, n_), file=sys.stderr)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(net)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pipeline_bench(a, c, dev, tdt, rank, world, dist):
    """--config c3 / c4: the whole-pipeline workloads (not the headline metric; same timing contract)."""
    from step_amd import backbone, ops, workloads
    tubes = a.tubes or (11 if a.config == "c3" else 5)
    if a.config == "c3":
        w = workloads.C3Inference(dev, tdt, batch=CLIPS_PER_GPU, tubes=tubes, seed=123 + rank, graph=not a.no_graph)
        what = "C3: full two_branch inference (I3D backbone + ContextNet + 3 refinement steps with ROIAlign over %d tubes/clip + " \
               "batched per-class NMS), %d x [36,3,400,400] clips per GPU" % (tubes, CLIPS_PER_GPU)
        metric = "clips_per_sec_inference_T36_400"
    elif a.select or os.environ.get("STEP_BENCH_SELECT", "0") == "1":
        # the reference's whole iteration: eval inference over the first two steps + train_select between the steps.  16-bit, one process or
        # several: the padded form replayed as HIP graphs (front | host selection | heads + backward [| eager exchange | update]); --no-graph
        # or fp32: the ragged eager iteration
        graphed = not a.no_graph and a.dtype != "f32"
        w = workloads.C4SelectTrainStep(dev, batch=CLIPS_PER_GPU, seed=123 + rank, dtype=tdt, capturable=graphed, force_exchange=a.force_exchange)
        if graphed:
            w.capture(warmup=max(a.warmup, 2))
        what = "C4 (with proposal selection, train.py:257-348): backbone + ContextNet, no-grad inference() over 2 steps on 34 tubes/clip, " \
               "train_select (<= 5 positives + 2x negatives per clip and step%s) + ROIAlign + head + losses for the 3 steps, backward, " \
               "flat gradient all-reduce, fused Adam, %d x [36,3,400,400] clip(s) per GPU" % (
                   ", padded to %d slots per clip with zero-weight rows" % w.budget if graphed else "", CLIPS_PER_GPU)
        metric = "clips_per_sec_train_T36_400"
    else:
        # one process: the whole step (forward, backward, re-pack, Adam) is captured in a HIP graph and replayed -- the eager step
        # with 16-bit activations is bound by the host issuing ~1300 launches; several ranks: eager (the bucketed exchange)
        # (fp32 is GPU-bound and measured SLOWER replayed than eager -- 54.6 vs 51.0 ms -- so only the 16-bit step is captured)
        # several ranks: the SAME captured program -- the bucket all-reduces on the communication stream are recorded in the graph
        # (RCCL), or, where they cannot be (gloo), forward/backward and the update are two graphs around one eager flat all-reduce
        graphed = not a.no_graph and a.dtype != "f32"
        w = workloads.C4TrainStep(dev, batch=CLIPS_PER_GPU, tubes_per_clip=tubes, seed=123 + rank, dtype=tdt, capturable=graphed,
                                  force_exchange=a.force_exchange)
        if graphed:
            w.capture(warmup=max(a.warmup, 2), mode=a.c4_graph)
        what = "C4: one training step (backbone + ContextNet + max_iter=3 heads on 3/3/9-frame tubes, BCE + smooth-L1 losses, gradient all-reduce, Adam), " \
               "%d x [36,3,400,400] clip(s) per GPU, %d tubes/clip" % (CLIPS_PER_GPU, tubes)
        metric = "clips_per_sec_train_T36_400"
    # C3 with a captured graph: TWO batches in flight (see main(): the same idea) -- a second workload object on the same networks,
    # its own clips / captured graph / stream; batch k + 1 is launched before batch k is post-processed (its one host sync waits for
    # its own stream only).  Every step is still one batch through the whole pipeline, post-processing included.
    nfl = a.in_flight if (a.config == "c3" and not a.no_graph) else 1
    ws, streams = [w], [torch.cuda.current_stream()]
    for i in range(1, nfl):
        si = torch.cuda.Stream()
        with torch.cuda.stream(si):
            ws.append(workloads.C3Inference(dev, tdt, batch=CLIPS_PER_GPU, tubes=tubes, seed=123 + rank, share=w))
            torch.cuda.synchronize()
        streams.append(si)

    def run(k_steps, n):
        if n == 1:
            for _ in range(k_steps):
                w.step()
            return
        pending = None
        for k in range(k_steps):
            i = k % n
            with torch.cuda.stream(streams[i]):
                h = ws[i].launch()
            if pending is not None:
                with torch.cuda.stream(streams[pending[0]]):
                    ws[pending[0]].finish(pending[1])
            pending = (i, h)
        with torch.cuda.stream(streams[pending[0]]):
            ws[pending[0]].finish(pending[1])

    def timed(n):
        run(max(a.warmup, 2), n)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(a.steps, n)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    el = timed(nfl)
    el_one = timed(1) if nfl > 1 else el
    if dist is not None:
        t = torch.tensor([el, el_one], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, el_one = float(t[0].item()), float(t[1].item())
    # dominant kernel of one eager, single-stream, instrumented step -- run by EVERY rank (the C4 step contains the
    # gradient all-reduce: a collective issued by rank 0 alone would dead-lock against the others' final barrier)
    ops.PROFILE = []
    saved, backbone.BRANCH_STREAMS = backbone.BRANCH_STREAMS, False
    (w.eager if a.config == "c3" else getattr(w, "_eager_step", w.step))()
    torch.cuda.synchronize()
    backbone.BRANCH_STREAMS = saved
    REC, ops.PROFILE = ops.PROFILE, None
    if rank == 0:
        val = world * CLIPS_PER_GPU * a.steps / el
        out = {"metric": metric, "value": round(val, 2), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": max(a.warmup, 2),
               "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": what + (", inputs resident in HBM (in the captured step's own input buffer, as the C2 loop has them), random-init weights" if (a.config == "c3" and not a.no_graph)
                                              else ", inputs resident in HBM, random-init weights"), "clips_per_gpu": CLIPS_PER_GPU, "tubes_per_clip": tubes, "T": 36, "HW": 400,
                          "parallelism": "clip-sharded replicas x%d (%s)" % (world, "no data-path collective" if a.config == "c3" else "one gradient all-reduce per step"),
                          "batches_in_flight": nfl,
                          "launch": ("hipGraph replay + eager post-processing" + (", %d batches in flight (own clips / graph / stream each; batch k + 1 launched before batch k is post-processed)" % nfl if nfl > 1 else ""))
                                    if (a.config == "c3" and not a.no_graph) else
                                    ({"one": "hipGraph replay (whole training step%s)" % (", the bucketed RCCL gradient all-reduces recorded in the graph" if getattr(w.reducer, "active", False) else ""),
                                      "split": "two hipGraphs (forward + backward | re-pack + Adam) around one eager flat gradient all-reduce",
                                      "select": "two hipGraphs (backbone + ContextNet + no-grad inference | heads + losses + backward + re-pack + Adam) around the host's proposal selection",
                                      "select-split": "three hipGraphs (backbone + ContextNet + no-grad inference | heads + losses + backward | re-pack + Adam) around the host's proposal "
                                                      "selection and one eager flat gradient all-reduce"}[w.graph_mode]
                                     if getattr(w, "graph", None) is not None else "eager"),
                          "gradient_exchange": (("one flat all-reduce of the gradient arena between the two graphs, %s" if getattr(w, "graph_mode", None) in ("split", "select-split") else
                                                 "bucketed all-reduce, %d buckets, overlapped with backward, %%s" % len(w.reducer.buckets)) %
                                                ("one-rank RCCL group (forced)" if world == 1 else "%d ranks" % world))
                                               if getattr(getattr(w, "reducer", None), "active", False) else None},
               "ranks": {"world_size": world, "backend": (dist.get_backend() + " (RCCL over xGMI)" if dist.get_backend() == "nccl" else dist.get_backend()) if dist is not None else None,
                         "devices_visible": torch.cuda.device_count()}}
        if nfl > 1:
            out["one_batch_in_flight"] = {"value": round(world * CLIPS_PER_GPU * a.steps / el_one, 2), "ms_per_step": round(el_one / a.steps * 1e3, 4),
                                          "note": "the same K steps one after the other on one stream (the loop of rounds 1-2)"}
        rec = REC
        agg = {}
        for name, flops, nbytes, e0, e1 in rec:
            r = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            r[0] += 1; r[1] += e0.elapsed_time(e1); r[2] += flops; r[3] += nbytes
        if agg:
            name, (cnt, ms, flops, nbytes) = max(agg.items(), key=lambda kv: kv[1][1])
            mf = flops / (PEAK[a.dtype] * 1e12) >= nbytes / (PEAK_HBM_GBS * 1e9)
            ach = (flops / (ms * 1e-3) / 1e12) if mf else (nbytes / (ms * 1e-3) / 1e9)
            peak = PEAK[a.dtype] if mf else PEAK_HBM_GBS
            traffic, tsrc = None, None
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
                # per-launch traffic depends on the workload: only a table taken at THIS run's clips / tubes is quoted
                for key in [k for k in tj if k.startswith("kernels_" + a.config)]:
                    wl = tj.get("workload_" + key[len("kernels_"):], {})
                    if wl.get("clips") != CLIPS_PER_GPU or wl.get("tubes") != tubes:
                        continue
                    hit = tj[key].get(name)
                    if hit:
                        traffic = hit.get("hbm_bytes_per_launch")
                        tsrc = "NOT measured in this run: profiles/traffic_latest.json %s (%s; %s; taken at commit %s)" % (
                            key, tj.get("note_" + key[len("kernels_"):], ""), tj.get("source"), tj.get("commit_" + key[len("kernels_"):], tj.get("commit", "?")))
                if traffic is None:
                    tsrc = "no PMC table for this kernel at %d clips x %d tubes in profiles/traffic_latest.json" % (CLIPS_PER_GPU, tubes)
            except Exception:
                pass
            out["roofline"] = {"kernel": name, "bound": "mfma" if mf else "hbm", "achieved": round(ach, 2), "peak": peak,
                               "unit": "TFLOP/s" if mf else "GB/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_from": tsrc,
                               "algorithmic_gflop_per_launch": round(flops / cnt / 1e9, 3), "algorithmic_mb_per_launch": round(nbytes / cnt / 1e6, 3),
                               "launches_per_step": cnt, "avg_launch_ms": round(ms / cnt, 4),
                               "share_of_instrumented_kernel_time": round(ms / sum(v[1] for v in agg.values()), 3),
                               "note": "instrumented launches of step_amd.ops: conv / stem / pool forward, the data-gradient convs and conv weight gradients (torch glue kernels are not in this table)"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_pipeline_baseline(a.config, w, tubes)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
