import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mantaflow_amd import _lib, core
if sys.argv[1] != "default":
    _lib.use_library(os.path.abspath(sys.argv[1]), "cuda")
n = 256
s = core.Solver(name="m", gridSize=core.vec3(n, n, n))
lib = s.lib
flags = core.FlagGrid(s); flags.initDomain(boundaryWidth=0); flags.fillGrid()
A0, Ai, Aj, Ak, dst, src = (core.Grid(s) for _ in range(6))
lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (n, n, n)).astype(np.float32))
us = ctypes.c_double()
for _ in range(3):
    lib.call("mf_time_apply_matrix", n, n, n, flags.ptr, dst.ptr, src.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, 200, ctypes.byref(us), s.stream)
    print("%s: %.2f us  %.1f GB/s" % (os.path.basename(sys.argv[1]), us.value, 28 * n ** 3 / us.value / 1e3))
