#!/bin/bash
# round 6: the heads' pointwise GEMMs at C3 size with the reference's 34 tubes per clip (4 x 34 tubes x Tl frames x 7 x 7 rows: 59 976 at Tl = 9, 19 992 at Tl = 3)
#   tools/r06_heads_ab.sh shapes | res | poolconv
cd ${GRAFT_REPO_ROOT:-/root/repo}
case ${1:-shapes} in
shapes)
  for D in 9 3; do
    echo "== Tl = $D"
    python tools/ab_bench.py --batch 136 --rounds 5 --iters 10 \
      --custom a832_1024,832,1024,1,$D,7,7 --custom b256_1024,256,1024,1,$D,7,7 --custom c1024_256,1024,256,1,$D,7,7 \
      --custom d832_256,832,256,1,$D,7,7 --custom e256_256,256,256,1,$D,7,7 \
      --var default --var conv_pws=1 --var conv_waves=4 --var conv_waves=8,conv_nb=3 --var conv_waves=8,conv_nb=2 --var conv_waves=8,conv_nb=1 2>&1 | tail -8
  done;;
res)
  for D in 9 3; do
    echo "== Tl = $D, residual added before the ReLU (Bottleneck conv3 / the resample block's second halves)"
    python tools/ab_bench.py --batch 136 --rounds 5 --iters 10 \
      --custom b256_1024r,256,1024,1,$D,7,7,1 --custom e256_256r,256,256,1,$D,7,7,1 --custom b11_256_1024r,256,1024,1,$D,7,7,1 \
      --var conv_pws=0 --var conv_pws=1,conv_pws_waves=8 --var conv_pws=1,conv_pws_waves=16 --var default 2>&1 | tail -5
  done;;
poolconv)
  for t in 34 11; do for nb in 2 1 0; do
    python tools/bench_with.py ops.POOL_CONV_MAX_NB=$nb -- --config c3 --tubes $t --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('c3 tubes $t POOL_CONV_MAX_NB=$nb:', j['value'], 'clips/s', j['ms_per_step'], 'ms; one at a time', j['one_batch_in_flight']['value'])"
  done; done;;
esac
