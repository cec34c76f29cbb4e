import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
from mantaflow_amd import core, plugins
import ctypes
from mantaflow_amd import _lib
for warm in (2, 40, 120):
    r = bench.config4_dam(torch, core, plugins, steps=4, warm=warm)
    sc = (ctypes.c_int32 * 3)()
    _lib.get().cdll.mf_cg_last_shortcut(sc)
    print("steps %d-%d: %.2f ms per step, solvePressure %.2f ms, CG iterations %s, swept x-range %s of %d cells" %
          (warm + 1, warm + 4, r["ms_per_step"], r["ops_ms"]["solvePressure"], r["cg_iterations"], (sc[1], sc[1] + sc[2]) if sc[2] else "whole rows", 384))
