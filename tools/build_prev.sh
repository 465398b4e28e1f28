#!/bin/bash
# tools/build_prev.sh [commit] -- build the library of an EARLIER commit into tools/libstep_amd_prev.so, so that tools/ab_bench.py can
# time it against the working tree's library in ONE process (`--var lib=prev`): boxes of the pool differ by 5-10 %, which is more
# than most kernel changes are worth, so only same-process, interleaved A/Bs are trusted.
set -e
C=${1:-HEAD}
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/step_prev_wt
rm -rf $W; git -C $R worktree prune
git -C $R worktree add -f --detach $W $C > /dev/null
make -s -C $W/step_amd/csrc -j8 > /dev/null 2>&1
cp $W/step_amd/libstep_amd.so $R/tools/libstep_amd_prev.so
git -C $R worktree remove --force $W
echo "tools/libstep_amd_prev.so = $(git -C $R rev-parse --short $C)"
