#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 300 python tools/c3_bench.py 2>&1 | grep -v amdgpu | tail -25
