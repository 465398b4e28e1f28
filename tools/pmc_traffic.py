"""tools/pmc_traffic.py -- HBM traffic of one kernel from rocprofv3 PMC passes.

Reads the counter_collection CSVs of two rocprofv3 runs (--pmc FETCH_SIZE ; --pmc WRITE_SIZE, collected in
separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2) and writes
profiles/traffic_latest.json for bench.py.  Units: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports HALF the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section), so the read side
is doubled; WRITE_SIZE is uncalibrated and taken as is."""
import csv
import glob
import json
import os
import sys


def mean_counter(d, kernel_substr, counter):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main(fetch_dir, write_dir, out, *kernels):
    """kernels: full kernel names as rocprofv3 prints them (bench.py looks its dominant kernel up by that name)."""
    res = {}
    for kernel in kernels:
        f, nf = mean_counter(fetch_dir, kernel, "FETCH_SIZE")
        w, nw = mean_counter(write_dir, kernel, "WRITE_SIZE")
        if f is None or w is None:
            print("kernel not found in the PMC output: %s" % kernel)
            continue
        rd = f * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128-B request
        wr = w * 1024
        res[kernel] = {"hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr), "launches_sampled": [nf, nw]}
    commit = "?"
    try:
        import subprocess
        commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rev-parse", "--short", "HEAD"],
                                         stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = os.environ.get("STEP_COMMIT", "?")          # (the GPU box has no .git: tools/gpu_profiles.sh passes it in)
    j = {"kernels": res, "commit": commit,
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE x1024 x2 (gfx950 correction), WRITE_SIZE x1024"}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j))
    return 0 if res else 1


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
