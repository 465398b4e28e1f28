#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --verbose 2>&1 >/dev/null | grep -E "stem_stream|3, 3, 3, 3, 3, 2, 2, 8, 1" | head -3
