cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -x -q -k "roi or inference_golden" 2>&1 | tail -2
for t in 34 11; do for l in none roiserial; do
if [ $l = none ]; then A=""; else A="lib=$l"; fi
python tools/bench_with.py $A -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t lib=$l:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c3_34 -- python $GRAFT_REPO_ROOT/bench.py --config c3 --tubes 34 --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_c3_34 $GRAFT_REPO_ROOT/gpurun_out/r06p_c3_34_kernel_stats.txt > /dev/null 2>&1; rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c3_34
head -24 $GRAFT_REPO_ROOT/gpurun_out/r06p_c3_34_kernel_stats.txt | cut -c1-150
