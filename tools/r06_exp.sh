cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pointwise or conv_units or pool_conv or big_conv or group" 2>&1 | tail -2
echo "== heads' shapes: XOR-swizzled 64-byte slab pitch (working tree) against the 80-byte pitch (lib=prev: the ring build)"
for D in 9 3; do
python tools/ab_bench.py --batch 136 --rounds 5 --iters 10 \
  --custom a832_1024,832,1024,1,$D,7,7 --custom c1024_256,1024,256,1,$D,7,7 --custom d832_256,832,256,1,$D,7,7 --custom b256_1024r,256,1024,1,$D,7,7,1 \
  --var conv_pws=0,lib=prev --var conv_pws=0 --var conv_pws=0,conv_waves=4,lib=prev --var conv_pws=0,conv_waves=4 --var conv_pws=0,conv_nb=3,lib=prev --var conv_pws=0,conv_nb=3 2>&1 | tail -6
done
python tools/step_ab.py --var lib=prev --var default 2>&1 | tail -4
for t in 34 11; do for l in none prev none prev; do
if [ $l = none ]; then A=""; else A="lib=$l"; fi
python tools/bench_with.py $A -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t lib=$l:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
done; done
