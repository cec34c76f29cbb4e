"""Timing of mf_mic_apply with an alternative build of the library (tools/micro/mic_ablate.sh): python mic_ablate.py LIB NX,NY,NZ [reps].
Timing only -- ablated builds compute wrong values."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mantaflow_amd import _lib, core

lib_path, dims = sys.argv[1], tuple(int(v) for v in sys.argv[2].split(","))
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
if lib_path != "default":
    _lib.use_library(os.path.abspath(lib_path), "cuda")
nx, ny, nz = dims
s = core.Solver(name="m", gridSize=core.vec3(nx, ny, nz))
lib = s.lib
flags = core.FlagGrid(s)
flags.initDomain(boundaryWidth=0)
flags.fillGrid()
A0, Ai, Aj, Ak, ap, dst, src = (core.Grid(s) for _ in range(7))
lib.call("mf_make_laplace_matrix", nx, ny, nz, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (nz, ny, nx)).astype(np.float32))
lib.call("mf_mic_init", nx, ny, nz, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    lib.call("mf_mic_apply", nx, ny, nz, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
e0.record()
for _ in range(reps):
    lib.call("mf_mic_apply", nx, ny, nz, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
e1.record()
torch.cuda.synchronize()
print("%s %s: %.1f us per apply" % (os.path.basename(lib_path), dims, e0.elapsed_time(e1) * 1e3 / reps))
