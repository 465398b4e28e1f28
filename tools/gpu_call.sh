#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "stem" 2>&1 | tail -2
for v in 1 2 3; do timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; ks=[r]+r['next_kernels']; print(d['value'], [(k['kernel'][11:40], k['avg_launch_ms'], k['frac']) for k in ks[:3]])"; done
