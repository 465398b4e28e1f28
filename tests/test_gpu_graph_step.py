"""The whole C4 training step captured in one HIP graph (step_amd.workloads.C4TrainStep.capture) against the same steps launched
eagerly: the same parameter trajectory VALUE FOR VALUE (every gradient sums in a fixed order, so the replayed kernels reproduce
the eager ones), the device-side Adam step counter advances on every replay, the loss tensor is refreshed in place."""
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
    # same kernels, same inputs, fixed summation orders: the trajectories agree value for value (1e-5 leaves room for a library
    # GEMM of the heads choosing another algorithm under capture; measured: bit-identical)
    rel = float(np.linalg.norm(da - db) / np.linalg.norm(da))
    em = float(np.linalg.norm(ma - mb) / np.linalg.norm(ma))
    import json, os
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ""), "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT") and os.path.isdir(d):
        json.dump({"rel": rel, "em": em, "identical": bool(np.array_equal(da, db)), "losses": [la.tolist(), lb.tolist()]},
                  open(os.path.join(d, "graph_vs_eager_%s.json" % str(dtype).split(".")[-1]), "w"))
    assert rel < 1e-5 and em < 1e-5, (rel, em)
    assert np.all(np.abs(la - lb) <= 1e-6 * np.abs(la)), (la, lb)
    assert len(set(np.round(lb, 10))) > 1                       # the replays really advance the weights


def test_two_inference_batches_in_flight_equal_one_at_a_time():
    """workloads.C3Inference.launch() / finish(): batch k + 1 (a second workload object on the SAME networks, own clips, own captured
    graph, own stream) is launched before batch k is post-processed -- the detections of both equal the ones each produces alone."""
    from step_amd import workloads

    dev = torch.device("cuda:0")
    a = workloads.C3Inference(dev, torch.bfloat16, batch=1, tubes=5, seed=123)
    sb = torch.cuda.Stream()
    with torch.cuda.stream(sb):
        b = workloads.C3Inference(dev, torch.bfloat16, batch=1, tubes=5, seed=123, share=a)
        torch.cuda.synchronize()
    assert b.base is a.base and not torch.equal(a.x, b.x)
    alone = [a.step(), None]
    with torch.cuda.stream(sb):
        alone[1] = b.step()
    torch.cuda.synchronize()
    sa = torch.cuda.current_stream()
    got = [None, None]
    for _ in range(3):                                          # a few rounds of the pipelined loop
        ha = a.launch()
        with torch.cuda.stream(sb):
            hb = b.launch()
        got[0] = a.finish(ha)
        with torch.cuda.stream(sb):
            got[1] = b.finish(hb)
    torch.cuda.synchronize()
    del sa

    def same(p, q):
        if isinstance(p, torch.Tensor):
            return torch.equal(p, q)
        if isinstance(p, np.ndarray):
            return np.array_equal(p, q)
        if isinstance(p, (list, tuple)):
            return len(p) == len(q) and all(same(x, y) for x, y in zip(p, q))
        if isinstance(p, dict):
            return p.keys() == q.keys() and all(same(p[k], q[k]) for k in p)
        return p == q
    assert same(got[0], alone[0]) and same(got[1], alone[1])


def test_two_backbone_batches_in_flight_equal_one_at_a_time():
    """bench.py's default loop for the forward-only configs: ONE network, one captured step per batch on its own stream, replayed
    round-robin so that consecutive batches overlap on the chip.  The features of both batches equal the eager ones bit for bit
    (scratch buffers are per call, i.e. per graph; packed weights are shared and read-only)."""
    from types import SimpleNamespace as NS

    import step_amd
    from oracle import i3d_ref as R

    dev = torch.device("cuda:0")
    cfg = NS(base_net="i3d", kinetics_pretrain=None, freeze_stats=True, freeze_affine=True, fp16=False)
    net = step_amd.BaseNet(cfg)
    net.load_state_dict(R.fill_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}))
    net = net.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    xs = [(torch.rand(2, 32, 3, 224, 224, generator=g) * 2 - 1).to(dev).bfloat16() for _ in range(2)]
    with torch.no_grad():
        want = [net(x).clone() for x in xs]
    torch.cuda.synchronize()
    assert not torch.equal(want[0], want[1])
    flights = []
    for x in xs:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):
                net(x)
            s.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=s):
                y = net(x)
        flights.append((s, graph, y))
    torch.cuda.synchronize()
    for k in range(12):                                         # round-robin replays: batch k + 1 starts while batch k drains
        s, graph, _ = flights[k % 2]
        with torch.cuda.stream(s):
            graph.replay()
    torch.cuda.synchronize()
    for (s, graph, y), w in zip(flights, want):
        assert torch.equal(y, w)


def test_captured_selected_step_equals_the_eager_padded_step_bit_for_bit():
    """the reference's WHOLE training iteration (train.py:257-348: no-grad inference over two steps, train_select between the
    steps, three heads, backward, Adam) replayed as HIP graphs -- workloads.C4SelectTrainStep.capture(): graph F (backbone + ContextNet with
    gradients + inference), the HOST's proposal selection (same draws from `random` / `numpy.random` as the reference), graph B (ROIAlign + heads +
    losses + backward + re-pack + Adam).  Shapes are static because every clip's selection is padded to 15 slots whose rows have weight zero.
    (1) captured == eager padded, BIT FOR BIT over the parameter trajectory (same kernels on the same buffers, fixed summation orders), with the
        same selections in every iteration;
    (2) padded == the ragged eager iteration up to summation order (the padded rows contribute exact zeros, but every head launch runs on 15 instead
        of n rows per clip: other tile shapes and split-K chunks): the same selections in every iteration, the FIRST iteration's loss -- computed
        before any update -- within 1e-4; later losses within 5 % (Adam turns the sign of every near-zero gradient component into a full +-lr
        step, so trajectories that differ in rounding drift apart at the 1 % level within two updates: 2.2448 against 2.2184 measured)."""
    import random
    from step_amd import workloads

    dev = torch.device("cuda:0")
    iters, warm = 5, 2
    out, first = {}, {}
    for mode in ("ragged", "padded", "graph"):
        torch.manual_seed(7)
        random.seed(5)
        np.random.seed(5)
        w = workloads.C4SelectTrainStep(dev, batch=2, seed=31, dtype=torch.bfloat16, capturable=(mode == "graph"))
        for g_ in w.opt.param_groups:
            g_["lr"] = 1e-4
        p0 = w.opt.flat_param.clone()
        losses, sels = [], []
        if mode == "graph":
            w.capture(warmup=warm)                               # `warm` eager padded steps (they draw from the RNG streams like any other), records, replays nothing
            assert w.graph_mode == "select" and w.opt.step_count == warm
            for _ in range(iters - warm):
                losses.append(float(w.step()))
                sels.append([list(x) for x in w.selected])
            assert w.opt.step_count == iters
        else:
            for i in range(iters):
                l = float(w.step_padded() if mode == "padded" else w.step())
                if i == 0:
                    first[mode] = l
                if i >= warm:
                    losses.append(l)
                    sels.append([list(x) for x in w.selected])
        torch.cuda.synchronize()
        out[mode] = ((w.opt.flat_param - p0).double().cpu().numpy(), np.array(losses), sels)
        del w
        torch.cuda.empty_cache()
    dr, lr_, sr = out["ragged"]
    dp, lp, sp = out["padded"]
    dg, lg, sg = out["graph"]
    import json, os
    rel_pr = float(np.linalg.norm(dp - dr) / np.linalg.norm(dr))
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ""), "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT") and os.path.isdir(d):
        json.dump({"graph_identical_to_padded": bool(np.array_equal(dg, dp)), "padded_vs_ragged_rel": rel_pr, "first_loss": first, "losses": [lr_.tolist(), lp.tolist(), lg.tolist()],
                   "selected": sg}, open(os.path.join(d, "captured_select.json"), "w"))
    assert np.isfinite(dg).all() and np.abs(dg).max() > 0
    assert sg == sp and np.array_equal(dg, dp) and np.array_equal(lg, lp), (sg, sp, lg, lp)
    assert all(1 <= n <= 15 for it in sg for st in it for n in st), sg
    # (the capture's warm-up drew the same numbers as the eager runs' first two iterations: the selections line up with the ragged run too)
    assert sp == sr, (sp, sr)
    assert abs(first["padded"] - first["ragged"]) <= 1e-4 * abs(first["ragged"]), first
    assert np.all(np.abs(lp - lr_) <= 5e-2 * np.abs(lr_)), (lp, lr_)


@pytest.mark.timeout(600)
def test_bench_fed_loop_reports_both_forms():
    """`bench.py --feed u8`: the contract line keeps the resident-input `value`; the `fed` block carries the fed rate (pinned host uint8 frames ->
    copy stream -> the captured step), both forms of it (conversion pass | the stem staging uint8 itself), the copy-only link rate and the
    named bottleneck; the multi-batch steps were captured under the library's throughput profile with identical bits (asserted in the run)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--feed", "u8", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--sustained-seconds", "0.3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"] == "clips_per_sec_T32_224" and j["config"]["batches_in_flight"] == 2 and "throughput" in j["config"]["launch"]
    f = j["fed"]
    assert f["form"] in ("u8_stem", "convert") and f["forms"]["convert"] > 0 and f["forms"]["u8_stem"] > 0
    # structural checks only: rates and ratios depend on the box's link and load -- they are RECORDED (gpurun_out), not bounded
    assert f["value"] == max(f["forms"].values()) and f["fed_over_resident"] > 0 and f["resident_value"] > 0
    assert f["h2d_GBs_copy_only"] > 0 and f["bottleneck"] and "uint8" in f["wire_format"]
    assert j["roofline"]["frac"] > 0 and j["sustained"]["value"] > 0
    # round 6: the line explains itself -- the step launch by launch, the planner profile of each loop, effective clocks, the fp16 leg
    kt = j["kernel_table"]
    assert len(kt) >= 20 and all(r["us"] > 0 and ("tflops" in r or "gbs" in r) for r in kt)
    assert abs(sum(r["us"] for r in kt) * 1e-3 - j["kernel_time_ms_per_step"]) < 0.02 * j["kernel_time_ms_per_step"] + 0.01
    assert j["planner_profile"]["value"] == "throughput" and j["planner_profile"]["one_batch_in_flight"] == "default"
    assert set(j["clock_ghz"]) >= {"two_in_flight_sustained", "one_batch_loop", "prefix_graph_replays"}
    # (recorded, not bounded at 1e-3 here: this run's weights are random draws -- 1.2e-3 measured; the 1e-3 bar is asserted against the oracle on the golden
    # configuration in tests/module_cases.py case_c2_full_size_properties)
    assert j["fp16"]["value"] > 0 and 0 < j["fp16"]["rel_err_vs_fp32"] < j["fp16"]["bf16_rel_err_vs_fp32"] < 5e-2
    assert "r06" in j["roofline"]["profile"]
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ""), "gpurun_out")
    if os.environ.get("GRAFT_REPO_ROOT") and os.path.isdir(d):
        json.dump(j, open(os.path.join(d, "bench_fed_test_line.json"), "w"))
