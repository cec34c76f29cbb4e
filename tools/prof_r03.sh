# round-3 evidence (run on the GPU box from the repo root): per-kernel stats of the bench command and of BASELINE configs 3, 4, 5 as
# bench.py runs them, the default bench line, the --slab line
export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/prof_r03$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_line_profiled.json 2> $OUT/bench_stats.err || exit 1
for c in config3 config4 config5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${c}_stats -- python3 tools/prof_kernels.py $c 3 > $OUT/${c}_stats.log 2>&1 || exit 1
done
MF_P2G_DET=0 python3 tools/prof_kernels.py config3 3 > $OUT/config3_atomic.json 2> $OUT/config3_atomic.err
python3 bench.py --slab --no-other-configs --no-cpu-baseline > $OUT/bench_slab_line.json 2> $OUT/slab.err
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err
tail -c 600 $OUT/bench_line.json
