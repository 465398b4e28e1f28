cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in -1 0; do STEP_FEED_PRIO=$p timeout 400 python bench.py --feed u8 --no-cpu-baseline --no-fp16-leg 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); f=j['fed']; print('copy stream priority $p: value', j['value'], 'sustained', j['sustained']['value'], 'fed', f['value'], f['forms'], 'fed/resident', f['fed_over_resident'])"; done
for f in none u8; do timeout 300 python train_step_amd.py --iters 40 --warmup-iters 3 --log-every 0 --feed $f 2>/dev/null | grep summary | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('train_step_amd feed $f:', j['ms_per_iter'], 'ms', j['launch'])"
timeout 300 python train_step_amd.py --iters 40 --warmup-iters 3 --log-every 0 --feed $f --select 2>/dev/null | grep summary | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('train_step_amd --select feed $f:', j['ms_per_iter'], 'ms', j['launch'])"; done
