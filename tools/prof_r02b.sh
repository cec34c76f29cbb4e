# round-2 final evidence (run on the GPU box from the repo root): per-kernel stats of the bench command, HBM counters of the MIC sweeps
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02b; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_line.json 2> $OUT/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/mic_fetch -- python3 tools/prof_kernels.py mic 10 > $OUT/mic_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/mic_write -- python3 tools/prof_kernels.py mic 10 > $OUT/mic_write.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
find $OUT -name "*kernel_stats.csv" | head; tail -c 400 $OUT/bench_line.json
