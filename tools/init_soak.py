"""repeat mf_mic_init at 256^3 (one dataflow sweep + bundle map + packed bytes): time per call and run-to-run identity"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mantaflow_amd import _lib, core
n = 256; reps = 300
lib = _lib.get()
s = core.Solver(gridSize=core.vec3(n, n, n), dim=3)
flags = core.FlagGrid(s); flags.initDomain(); flags.fillGrid()
A0, Ai, Aj, Ak, ap = (core.Grid(s) for _ in range(5))
lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
ref = ap.data.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(torch.cuda.current_stream())
for _ in range(reps):
    lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
print("mic_init %d^3: %.1f us per call (%d reps), identical to first: %s" % (n, e0.elapsed_time(e1) * 1e3 / reps, reps, bool(torch.equal(ref, ap.data))))
