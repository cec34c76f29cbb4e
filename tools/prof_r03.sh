# round-3 profile collection (run on the GPU box from the repo root): per-kernel stats of BASELINE configs 3, 4, 5 as bench.py runs them
export TMPDIR=/tmp
TAG=${1:-a}
OUT=gpurun_out/prof_r03$TAG; rm -rf $OUT; mkdir -p $OUT
for c in config3 config4 config5; do
  python3 tools/prof_kernels.py $c 3 > $OUT/$c.json 2> $OUT/$c.err || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${c}_stats -- python3 tools/prof_kernels.py $c 3 > $OUT/${c}_stats.log 2>&1 || exit 1
done
MF_P2G_DET=0 python3 tools/prof_kernels.py config3 3 > $OUT/config3_atomic.json 2> $OUT/config3_atomic.err
find $OUT -name "*kernel_stats.csv"
