#!/usr/bin/env python3
"""ISA statistics of the HIP kernels (tuning aid): compile a .hip to gfx950 assembly and print per kernel the
MFMA / LDS / scratch instruction counts, register use and static LDS.  usage: isa_stats.py conv_igemm [name-filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "step_amd", "csrc", sys.argv[1] + ".hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_%s.s" % sys.argv[1]
newest = max(os.path.getmtime(os.path.join(os.path.dirname(src), f)) for f in os.listdir(os.path.dirname(src)) if f.endswith((".hip", ".h")))
if not os.path.exists(out) or os.path.getmtime(out) < newest:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-strict-aliasing", "-w",
                           "--cuda-device-only", "-S", "-x", "hip", src, "-o", out])
s = open(out).read()
for m in re.finditer(r"^(_ZN4step\w+):.*?\n(.*?)\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    g = lambda k: re.search(k + r"\s+(\d+)", body).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    print(dem[:110])
    print("   mfma %d  ds_read_b128 %d  ds_read2_b32 %d  ds_read_b64 %d  ds_write %d  global_load %d  scratch %d  s_waitcnt %d  s_barrier %d | vgpr %s accum_offset %s lds %s B" % (
        body.count("v_mfma"), body.count("ds_read_b128"), body.count("ds_read2_b32"), len(re.findall(r"ds_read_b64\b", body)),
        body.count("ds_write"), body.count("global_load"), body.count("scratch_"), body.count("s_waitcnt"), body.count("s_barrier"),
        g(r"\.amdhsa_next_free_vgpr"), g(r"\.amdhsa_accum_offset"), g(r"\.amdhsa_group_segment_fixed_size")))
