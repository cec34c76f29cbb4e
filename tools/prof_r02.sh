# round-2 profile collection (run on the GPU box from the repo root): per-kernel stats and HBM counters of the advection kernels
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02; rm -rf $OUT; mkdir -p $OUT
python tools/prof_kernels.py advect 10 > $OUT/advect256.txt 2>&1
MF_GRID=512 python tools/prof_kernels.py advect 3 > $OUT/advect512.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adv_stats -- python3 tools/prof_kernels.py advect 5 > $OUT/adv_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/adv_fetch -- python3 tools/prof_kernels.py advect 3 > $OUT/adv_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/adv_write -- python3 tools/prof_kernels.py advect 3 > $OUT/adv_write.log 2>&1
MF_GRID=512 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adv512_stats -- python3 tools/prof_kernels.py advect 2 > $OUT/adv512_stats.log 2>&1
cat $OUT/advect256.txt $OUT/advect512.txt | grep MacCormack
