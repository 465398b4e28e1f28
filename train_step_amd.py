#!/usr/bin/env python
"""train_step_amd.py -- the one-process-per-GPU training launcher INTEGRATION.md names: what replaces the reference's
nn.DataParallel wrapping and cross-GPU head placement (train.py:142-148) around its training iteration (train.py:257-348).

    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 train_step_amd.py --iters 100
    python train_step_amd.py --iters 10                      # one GPU, no process group

Every rank
  1. joins the RCCL process group (`step_amd.dist.init`; backend "nccl" IS RCCL on ROCm, xGMI underneath),
  2. takes its slice of the global clip batch (`dist.shard_clips`: clips r, r+world, ... -- every tensor on the path is per clip and
     BN is frozen, so the shard needs no data-path collective),
  3. builds the replicas (BaseNet + ContextNet + max_iter heads; rank 0's weights are broadcast once, as DDP does at construction),
  4. captures the WHOLE step -- forward, backward with the conv weight gradients accumulated straight into FlatAdam's arena, the
     bucketed gradient all-reduce on the communication stream, the one-launch weight re-pack, the fused Adam -- in one HIP graph
     (`workloads.C4TrainStep.capture`), so the N > 1 step is the same replayed program as the N = 1 step plus its collectives,
  5. replays it --iters times, the learning-rate schedule written through `optimizer.param_groups` exactly as the reference's
     schedulers do (utils/solver.py:96-180: the captured Adam reads per-group lr / weight_decay from device tables that
     `step()` refreshes from the param groups),
  6. rank 0 prints one JSON line per --log-every iterations and a final summary (loss, ms per iteration, clips/s of the whole job).

Data: synthetic AVA-shaped clips [B,36,3,400,400] and fixed anchor tubes (there is no dataset in this repository; the reference's
loader, data/ava.py, hands over the same shapes).  --feed u8 keeps the clips as uint8 frames in pinned host memory and moves them
host -> device every iteration on a copy stream (`step_clip_from_u8` writes the captured step's input): the transfer of iteration
k + 1 runs under the replay of iteration k.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--global-batch", type=int, default=None, help="clips of the whole job per iteration (default: 1 per rank; the reference's scripts: 8)")
    ap.add_argument("--tubes", type=int, default=5, help="tubes per clip and step")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--warmup-iters", type=int, default=3, help="eager iterations before the capture (caches, workspaces, communicator)")
    ap.add_argument("--lr-decay-every", type=int, default=0, help="> 0: multiply every group's lr by 0.1 every that many iterations (scheduler stand-in)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph", default="auto", choices=["auto", "one", "split"])
    ap.add_argument("--select", action="store_true",
                    help="the reference's whole iteration (train.py:257-348): no-grad inference + train_select between the steps (workloads.C4SelectTrainStep, "
                         "captured as graphs around the host's selection)")
    ap.add_argument("--feed", default="none", choices=["none", "u8"])
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend (default: nccl = RCCL; gloo only to exercise the multi-rank program with ranks SHARING one GPU, "
                         "which RCCL refuses -- the exchange is then one eager flat all-reduce between two captured graphs)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("train_step_amd.py needs a ROCm device (there is no CPU fallback)")

    from step_amd import dist as sdist, ops, workloads
    rank, world = sdist.init(a.backend)
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    gb = a.global_batch or world
    mine = sdist.shard_clips(gb, rank, world)
    if len(set(len(sdist.shard_clips(gb, r, world)) for r in range(world))) != 1:
        raise SystemExit("train_step_amd.py: --global-batch must be a multiple of the world size (equal shards keep the mean-of-means "
                         "equal to the single-process loss, SURVEY 8e)")
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    graphed = not a.no_graph and a.dtype != "f32"               # (the fp32 step is GPU-bound and measured slower replayed than eager)
    if a.select:
        import random
        import numpy as np
        random.seed(1000 + rank)                                 # (the selection draws from the reference's two host RNG streams)
        np.random.seed(1000 + rank)
        w = workloads.C4SelectTrainStep(dev, batch=len(mine), seed=123 + rank, dtype=tdt, capturable=graphed)
    else:
        w = workloads.C4TrainStep(dev, batch=len(mine), tubes_per_clip=a.tubes, seed=123 + rank, dtype=tdt, capturable=graphed)
    for g in w.opt.param_groups:
        g["lr"] = a.lr
    if graphed:
        w.capture(warmup=a.warmup_iters, mode=a.graph)             # (C4SelectTrainStep: its own graph forms; `mode` only matters for the fixed-tube step)
    else:
        for _ in range(a.warmup_iters):
            w.step()

    feed = None
    if a.feed == "u8":
        N, T, _, H, W = w.x.shape
        g_ = torch.Generator().manual_seed(999 + rank)
        host = [torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8, generator=g_).pin_memory() for _ in range(2)]
        stage = [torch.empty((N, T, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        # a stream of its own at DEFAULT priority.  (Round 5 gave it high priority, as bench.py's forward-only fed loop does; beside the captured
        # TRAINING step -- whose graph forks onto side streams -- a high-priority stream that spends its time waiting on the main stream made
        # the whole iteration 3x slower: 40.3 against 13.6 ms, tools/feed_train_probe.py, profiles/r06_feed_train_probe.txt)
        copy_stream = torch.cuda.Stream()
        copied = [torch.cuda.Event() for _ in range(2)]
        main_stream = torch.cuda.current_stream()

        def prefetch(k):
            with torch.cuda.stream(copy_stream):
                stage[k % 2].copy_(host[k % 2], non_blocking=True)
                copied[k % 2].record(copy_stream)

        def feed(k):
            main_stream.wait_event(copied[k % 2])
            ops.clip_from_u8(stage[k % 2], scale=2, out=w.x)     # the captured step's static input
            copy_stream.wait_stream(main_stream)                 # (the next copy into the OTHER buffer may start at once; this one is re-used at k + 2)
            prefetch(k + 1)
        prefetch(0)

    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    tl = t0
    for it in range(a.iters):
        if a.lr_decay_every and it and it % a.lr_decay_every == 0:
            for g in w.opt.param_groups:                          # what utils/solver.py's schedulers do between iterations
                g["lr"] *= 0.1
        if feed is not None:
            feed(it)
        loss = w.step()
        if rank == 0 and a.log_every and (it + 1) % a.log_every == 0:
            lv = float(loss)                                      # (one host sync per log line)
            now = time.perf_counter()
            print(json.dumps({"iter": it + 1, "loss": round(lv, 6), "ms_per_iter": round((now - tl) / a.log_every * 1e3, 3),
                              "lr": w.opt.param_groups[0]["lr"]}), flush=True)
            tl = now
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
    if rank == 0:
        print(json.dumps({"summary": True, "world_size": world, "global_batch": gb, "clips_per_rank": len(mine), "iters": a.iters,
                          "ms_per_iter": round(el / max(a.iters, 1) * 1e3, 3), "clips_per_s": round(gb * a.iters / el, 3),
                          "launch": ("hipGraph replay (%s)" % w.graph_mode) if w.graph is not None else "eager",
                          "gradient_exchange": _exchange_label(w, world),
                          "feed": a.feed, "dtype": a.dtype, "final_loss": round(float(w.loss), 6), "adam_steps": w.opt.step_count}), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _exchange_label(w, world):
    """What the step's gradient exchange really was: the split form and gloo run ONE eager flat all-reduce, not the buckets."""
    if not w.reducer.active:
        return None
    backend = torch.distributed.get_backend()
    lib = "RCCL" if backend == "nccl" else backend
    if getattr(w, "graph_mode", None) in ("split", "select-split"):
        return "one eager flat %s all-reduce of the gradient arena between the two graphs, %d ranks" % (lib, world)
    where = "recorded in the step's graph" if w.graph is not None else "eager, overlapped with backward"
    return "bucketed %s all-reduce, %d buckets, %s, %d ranks" % (lib, len(w.reducer.buckets), where, world)


if __name__ == "__main__":
    main()
