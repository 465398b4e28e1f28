cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 34 11; do for l in 0 1 0 1; do
python tools/bench_with.py ops.POOL_CONV_FORCE_NB=$l -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t POOL_CONV_FORCE_NB=$l:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done; done
