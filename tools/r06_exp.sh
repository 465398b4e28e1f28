cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in 2 3 2 3; do
python bench.py --steps 20 --warmup 5 --in-flight $f --no-fp16-leg --no-cpu-baseline --feed none 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c2 in flight $f:', j['value'], 'clips/s', j['ms_per_step'], 'ms; sustained', j['sustained']['value'], '; one at a time', j['one_batch_in_flight']['value'])"
done
