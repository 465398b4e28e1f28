#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "basenet_backward or training_step_matches" 2>&1 | grep -E "^E  |passed|failed" | head -12
