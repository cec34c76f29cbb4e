"""time (and optionally trace) the MIC(0) apply at n^3 on the GPU: python tools/mic_time.py [n] [reps]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mantaflow_amd import _lib, core
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = _lib.get()
s = core.Solver(gridSize=core.vec3(n, n, n), dim=3)
flags = core.FlagGrid(s); flags.initDomain(); flags.fillGrid()
A0, Ai, Aj, Ak, src, dst, ap = (core.Grid(s) for _ in range(7))
lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (n, n, n)).astype(np.float32))
lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
for _ in range(3):
    lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(torch.cuda.current_stream())
for _ in range(reps):
    lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
e1.record(torch.cuda.current_stream())
torch.cuda.synchronize()
print("MIC apply %d^3: %.1f us per apply (%d reps)" % (n, e0.elapsed_time(e1) * 1e3 / reps, reps))
