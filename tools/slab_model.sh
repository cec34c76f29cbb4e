# pieces of the N>1 per-iteration cost model, measured on one MI355X (run from the repo root)
python bench.py --slab --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench --slab (1 rank through the slab path + RCCL world 1): %.2f ms/step, iterations %s' % (d['ms_per_step'], d['config']['cg_iterations_per_step'][:2]))"
for dims in 256,256,44 256,256,76 256,256,140; do
  MF_DIMS=$dims MF_MIC_BLOCK=64 MF_MIC_BLOCK_X=128 python tools/prof_kernels.py mic 30 2>&1 | tail -1
  MF_DIMS=$dims python tools/prof_kernels.py mic 30 2>&1 | tail -1
done
