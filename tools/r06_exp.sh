cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== heads' residual layers (136 maps x Tl x 7 x 7)"
for D in 9 3; do
python tools/ab_bench.py --batch 136 --rounds 5 --iters 10 \
  --custom b256_1024r,256,1024,1,$D,7,7,1 --custom e256_256r,256,256,1,$D,7,7,1 \
  --var conv_pws=0 --var conv_pws=1 --var default 2>&1 | tail -3
done
for t in 34 11; do
python bench.py --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
python bench.py --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline --opt conv_pws=0 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t conv_pws=0:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done
