"""tools/isa_live.py -- VGPR liveness of one kernel in a hipcc .s listing (tuning aid, CPU only).

    hipcc ... -S --cuda-device-only x.hip -o x.s ; python tools/isa_live.py x.s <kernel-substring> [--at LINE ...] [--top N]

Builds the control-flow graph of the kernel's assembly (labels, s_branch / s_cbranch_*), classifies every instruction's vector-register
defs and uses (first operand = destination for VALU / loads / MFMA, no destination for stores and waits, read-modify-write for
v_writelane / v_permlane*_swap / v_mac / v_fmac / v_accvgpr partial forms), runs the backward data-flow to a fixed point and prints the
number of live VGPRs before each requested line (1-based line numbers of the .s file) or the --top N pressure points.  Written to find
out WHICH values a persistent tile loop keeps alive across its epilogue (round 6: the allocator spilled the halo requests)."""
import re
import sys


def vregs(tok):
    out = []
    for m in re.finditer(r"\b[va]\[(\d+):(\d+)\]|\b[va](\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


NO_DST = ("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "s_", "ds_gws", "global_atomic", "v_cmp", "v_cmpx", "v_nop", "v_readlane", "v_readfirstlane")
RMW = ("v_writelane", "v_permlane", "v_mac", "v_fmac", "v_dot2c", "v_pk_fmac", "v_movrel")


def parse(lines, lo, hi):
    ins = []          # (lineno, op, defs, uses, label_targets, falls_through)
    labels = {}
    for i in range(lo, hi):
        t = lines[i].split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.L[\w$]+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if t.startswith("."):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        defs, uses, tgt, fall = [], [], None, True
        if op.startswith("s_branch"):
            tgt, fall = args[0], False
        elif op.startswith("s_cbranch"):
            tgt = args[0]
        elif op.startswith("s_endpgm"):
            fall = False
        if op.startswith(NO_DST) and not (op.startswith("v_cmp") and op.endswith("_e64") and False):
            for a in args:
                uses += vregs(a)
        else:
            if args:
                defs += vregs(args[0])
                if op.startswith(RMW):
                    uses += vregs(args[0])
                if op.startswith("v_permlane32_swap") or op.startswith("v_permlane16_swap") or op.startswith("v_swap"):
                    defs += vregs(args[1])
            for a in args[1:]:
                uses += vregs(a)
        ins.append([i + 1, op, set(defs), set(uses), tgt, fall])
    return ins, labels


def main():
    path, key = sys.argv[1], sys.argv[2]
    at, top = [], 0
    rest = sys.argv[3:]
    while rest:
        a = rest.pop(0)
        if a == "--at":
            while rest and not rest[0].startswith("--"):
                at.append(int(rest.pop(0)))
        elif a == "--top":
            top = int(rest.pop(0))
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^[\w$]+:", l) and key in l]
    if not start:
        raise SystemExit("kernel not found")
    lo = start[0]
    hi = next(i for i in range(lo, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    ins, labels = parse(lines, lo + 1, hi)
    n = len(ins)
    succ = [[] for _ in range(n)]
    for k, (_, op, d, u, tgt, fall) in enumerate(ins):
        if fall and k + 1 < n:
            succ[k].append(k + 1)
        if tgt and tgt in labels and labels[tgt] < n:
            succ[k].append(labels[tgt])
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for k in range(n - 1, -1, -1):
            out = set()
            for s_ in succ[k]:
                out |= live_in[s_]
            new = (out - ins[k][2]) | ins[k][3]
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    by_line = {ins[k][0]: k for k in range(n)}

    def ranges(s_):
        s_ = sorted(s_)
        if not s_:
            return "-"
        rs, a, b = [], s_[0], s_[0]
        for x in s_[1:]:
            if x == b + 1:
                b = x
            else:
                rs.append((a, b)); a = b = x
        rs.append((a, b))
        return " ".join("v%d" % a if a == b else "v%d-%d" % (a, b) for a, b in rs)
    for ln in at:
        k = by_line.get(ln)
        if k is None:
            k = min(by_line.values(), key=lambda q: abs(ins[q][0] - ln))
        print("line %d  %-28s live-in %3d : %s" % (ins[k][0], ins[k][1], len(live_in[k]), ranges(live_in[k])))
    if top:
        order = sorted(range(n), key=lambda k: -len(live_in[k]))[:top]
        for k in sorted(order):
            print("line %d  %-28s live-in %3d" % (ins[k][0], ins[k][1], len(live_in[k])))
    print("instructions %d, max live %d" % (n, max(len(s_) for s_ in live_in)))


if __name__ == "__main__":
    main()
