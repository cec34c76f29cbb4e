"""A/B of library builds on the bench step: python tools/micro/ntv_bench.py LIB [LIB ...] ("default" = the product library); prints ms per
256^3 smoke step (the plugin path of bench.py, 6 steps after 2 warm-ups) for each."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from mantaflow_amd import _lib, core, plugins
lib = sys.argv[1]
if lib != "default":
    _lib.use_library(os.path.abspath(lib), "cuda")
n = 256
s = core.Solver(gridSize=core.vec3(n, n, n), dim=3)
s.timestep = 1.0
flags = core.FlagGrid(s); flags.initDomain(); flags.fillGrid()
vel, vel0, dens, pres = core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
vel0.from_numpy(bench.synthetic_velocity(n, n, n)); dens.from_numpy(bench.synthetic_density(n, n, n))
plugins.setWallBcs(flags, vel0)
def step():
    vel.copyFrom(vel0)
    plugins.advectSemiLagrange(flags, vel, dens, order=2)
    plugins.advectSemiLagrange(flags, vel, vel, order=2)
    plugins.setWallBcs(flags, vel)
    plugins.solvePressure(vel, pres, flags, cgAccuracy=1e-3)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6): step()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 6
print("%%-28s %%.2f ms per step, %%d iterations" %% (os.path.basename(lib), el * 1e3, plugins.lastCgStats()["iterations"]))
''' % ROOT
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, "-c", code, lib], check=False)
