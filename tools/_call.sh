R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_modules.py -x -q -k "mixed or basenet_c1 or c2_full or data_parallel or i3d_class" 2>&1 | tail -3
python tools/ab_bench.py --only 2c_3x3,3b_b1b,3b_b2b,3c_b1b,3c_b2b,4b_b1b,4d_b1b,4f_b1b,4f_b2b --rounds 5 --var lib=prev --var default 2>&1 | tee gpurun_out/ab_act.log
cd /tmp; export TMPDIR=/tmp
for bs in 0 1; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$bs -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --branch-streams $bs > $R/gpurun_out/bench_kt$bs.json 2> $R/gpurun_out/bench_kt.err
python $R/tools/graph_timeline.py /tmp/kt$bs > $R/gpurun_out/graph_timeline_bs$bs.log 2>&1
cut -c1-160 $R/gpurun_out/bench_kt$bs.json
done
cat $R/gpurun_out/graph_timeline_bs0.log
tail -3 $R/gpurun_out/graph_timeline_bs1.log
