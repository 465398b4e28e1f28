"""tools/graph_timeline.py -- the timeline of ONE replayed step from a rocprofv3 kernel trace (GPU only, tuning aid).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/graph_timeline.py /tmp/kt [--anchor stem_stream_kernel]

A step = the kernels from one launch of the anchor kernel (the stem, first kernel of the backbone) to the next.  Prints every kernel
of the step with its start offset, duration and the idle time of the WHOLE GPU right before it started, then the union busy time,
the idle total and how long 1 / 2 / 3+ kernels were in flight.
"""
import argparse
import csv
import glob
import os
import re


def short(n):
    n = n.replace("void step::", "").replace("step::", "")
    n = re.sub(r"\(.*", "", n)
    return n[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--anchor", default="stem_stream_kernel")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if a.anchor in r[2]]
    assert len(anchors) >= 3, "anchor kernel not found often enough"
    spans = []
    for k in range(len(anchors) - 1):
        seg = rows[anchors[k]:anchors[k + 1]]
        spans.append(((max(e for _, e, _ in seg) - seg[0][0]) / 1e3, len(seg), k))
    # the replayed steps: the most common kernel count (bench.py's instrumented pass launches every kernel twice)
    counts = {}
    for _, n, _ in spans:
        counts[n] = counts.get(n, 0) + 1
    modal = max(counts, key=counts.get)
    rep = [x for x in spans if x[1] == modal]
    sp = sorted(x[0] for x in rep)
    print("replayed steps (%d kernels each): %d, span first kernel start -> last kernel end: min %.1f  median %.1f  max %.1f us" % (
        modal, len(rep), sp[0], sp[len(sp) // 2], sp[-1]))
    k = min(rep[len(rep) // 2:], key=lambda x: abs(x[0] - sp[len(sp) // 2]))[2]        # a typical one from the second half
    i0, i1 = anchors[k], anchors[k + 1]
    seg = rows[i0:i1]
    t0 = seg[0][0]
    print("%9s %8s %8s  %s" % ("start_us", "dur_us", "idle_b4", "kernel"))
    busy_until = t0
    ksum = 0.0
    for s, e, n in seg:
        idle = max(0, s - busy_until) / 1e3
        print("%9.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle, short(n)))
        busy_until = max(busy_until, e)
        ksum += (e - s) / 1e3
    # concurrency profile
    ev = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
    depth, last, hist = 0, t0, {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    span = (busy_until - t0) / 1e3
    print("span %.1f us, sum of kernel durations %.1f us, kernels %d" % (span, ksum, len(seg)))
    print("in flight: " + ", ".join("%d: %.1f us" % (k, v / 1e3) for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
