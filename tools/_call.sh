R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for bs in 0 1 2; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$bs -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --branch-streams $bs > $R/gpurun_out/bench_kt$bs.json 2> $R/gpurun_out/bench_kt.err
python $R/tools/graph_timeline.py /tmp/kt$bs > $R/gpurun_out/graph_timeline_bs$bs.log 2>&1
cut -c1-160 $R/gpurun_out/bench_kt$bs.json
done
cat $R/gpurun_out/graph_timeline_bs0.log
tail -3 $R/gpurun_out/graph_timeline_bs2.log
