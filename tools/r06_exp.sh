cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 34 11; do for l in none roiflat none roiflat; do
if [ $l = none ]; then A=""; else A="lib=$l"; fi
python tools/bench_with.py $A -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t lib=$l:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done; done
