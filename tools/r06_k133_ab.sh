#!/bin/bash
# round 6: the two-phase form for 1x3x3 windows (option conv_phased = 2) against the classic form on the configs whose heads use them
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for v in 1 2; do
  for t in 11 34; do
    timeout 400 python bench.py --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline --opt conv_phased=$v 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('conv_phased=$v rep $rep c3 tubes $t: %.1f clips/s (%.4f ms), one at a time %.1f' % (j['value'], j['ms_per_step'], j['one_batch_in_flight']['value']))"
  done
  timeout 400 python bench.py --config c4 --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline --opt conv_phased=$v 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('conv_phased=$v rep $rep c4 bf16: %.4f ms per step' % j['ms_per_step'])"
  timeout 400 python bench.py --config c4 --dtype bf16 --clips 8 --tubes 15 --steps 10 --warmup 3 --no-cpu-baseline --opt conv_phased=$v 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('conv_phased=$v rep $rep c4 bf16 8 x 15: %.4f ms per step' % j['ms_per_step'])"
done; done
