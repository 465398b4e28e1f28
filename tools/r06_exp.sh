cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_graph_step.py tests/test_gpu_modules.py -x -q -k "graph or inference or e2e or postprocess" 2>&1 | tail -2
for t in 34 11; do for f in True False True False; do
python tools/bench_with.py driver.COMPACT_KERNEL=$f -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t COMPACT_KERNEL=$f:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done; done
