#!/usr/bin/env python3
"""Focused launcher for rocprofv3: runs one hot kernel family at 256^3 so kernel-trace / PMC output stays small.
  python3 tools/prof_kernels.py apply_matrix [reps] | apply_matrix_packed [reps] | mic [reps] | flip | dam | wavelet | advect | config3 | config4 | config5
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from mantaflow_amd import _lib, core, plugins  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "apply_matrix"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n = int(os.environ.get("MF_GRID", "256"))
    dims = [int(v) for v in os.environ.get("MF_DIMS", "%d,%d,%d" % (n, n, n)).split(",")]
    if os.environ.get("MF_LIB"):      # an experimental build of the library (tools/micro): A/B timing
        _lib.use_library(os.path.abspath(os.environ["MF_LIB"]), "cuda")
    lib = _lib.get()
    if what in ("config3", "config4", "config5"):
        # exactly the workloads bench.py reports under other_configs (per-operator HIP-event times printed as JSON)
        import json
        fn = {"config3": bench.config3_sflip, "config4": bench.config4_dam, "config5": bench.config5_wavelet}[what]
        kw = {"steps": reps}
        if what == "config3":
            kw["deterministic"] = os.environ.get("MF_P2G_DET", "1") == "1"
        print(json.dumps(fn(torch, core, plugins, **kw), indent=1))
        return
    s = core.Solver(gridSize=core.vec3(*dims), dim=3)
    flags = core.FlagGrid(s); flags.initDomain(); flags.fillGrid()
    A0, Ai, Aj, Ak, src, dst, ap = (core.Grid(s) for _ in range(7))
    if what == "mic":
        nx, ny, nz = dims
        lib.call("mf_make_laplace_matrix", nx, ny, nz, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
        src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (nz, ny, nx)).astype(np.float32))
        # MF_MIC_BLOCK / MF_MIC_BLOCK_X: y / x blocking of the sweeps (timing only unless Aj / Ai are cut as well)
        lib.call("mf_mic_init_blocked", nx, ny, nz, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr,
                 int(os.environ.get("MF_MIC_BLOCK", "0")), int(os.environ.get("MF_MIC_BLOCK_X", "0")), s.stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            lib.call("mf_mic_apply", nx, ny, nz, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        e0.record()
        for _ in range(reps):
            lib.call("mf_mic_apply", nx, ny, nz, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        e1.record()
        torch.cuda.synchronize()
        nt = [(v + 7) // 8 for v in dims]
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("mic_apply %s: %.1f us per apply (2 sweeps), tile levels %d -> %.2f us per level per sweep" % (dims, us, sum(nt) - 2, us / 2 / (sum(nt) - 2)))
        return
    lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
    src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (n, n, n)).astype(np.float32))
    if what == "apply_matrix":
        us = ctypes.c_double()
        lib.call("mf_time_apply_matrix", n, n, n, flags.ptr, dst.ptr, src.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, reps, ctypes.byref(us), s.stream)
        print("apply_matrix avg %.2f us  %.1f GB/s" % (us.value, 28 * n ** 3 / us.value / 1e3))
    elif what == "apply_matrix_packed":
        # the variant the PCG loop runs: flags + Ai + Aj + Ak packed into one byte per cell by mf_mic_init
        lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        us = ctypes.c_double()
        lib.call("mf_time_apply_matrix_packed", n, n, n, flags.ptr, dst.ptr, src.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, reps, ctypes.byref(us), s.stream)
        print("apply_matrix (packed) avg %.2f us  %.1f GB/s of its 13 B per cell" % (us.value, 13 * n ** 3 / us.value / 1e3))
    elif what == "mic":
        lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        e1.record()
        torch.cuda.synchronize()
        print("mic_apply avg %.1f us" % (e0.elapsed_time(e1) * 1e3 / reps))
    elif what == "flip":
        # S-flip: fluid block = lower 0.4 x 0.6 x 1.0 of the box, 8 particles per cell, pvel ~ N(0, 0.5^2)
        from mantaflow_amd import scene
        s.timestep = 0.5
        plugins.setDeterministicP2G(os.environ.get("MF_P2G_DET", "1") == "1")
        flags.initDomain(boundaryWidth=0)
        box = scene.Box(parent=s, p0=core.vec3(0, 0, 0), p1=core.vec3(0.4 * n, 0.6 * n, n))
        flags.updateFromLevelset(box.computeLevelset())
        pp = core.BasicParticleSystem(s)
        scene.sampleFlagsWithParticles(flags, pp, 2, 0.2)
        pv = pp.create(core.PdataVec3)
        pv.from_numpy(np.random.default_rng(9832).normal(0, 0.5, (pp.np, 3)).astype(np.float32))
        vel, velOld, w, pres = core.MACGrid(s), core.MACGrid(s), core.VecGrid(s), core.Grid(s)
        ops = [
            ("advectInGrid(RK4)", lambda: pp.advectInGrid(flags, vel, 2, deleteInObstacle=False)),
            ("mapPartsToMAC", lambda: plugins.mapPartsToMAC(flags, vel, velOld, pp, pv, w)),
            ("extrapolateMACFromWeight", lambda: plugins.extrapolateMACFromWeight(vel, w, distance=2)),
            ("markFluidCells", lambda: plugins.markFluidCells(pp, flags)),
            ("addGravity", lambda: plugins.addGravity(flags, vel, core.vec3(0, -0.002, 0))),
            ("setWallBcs", lambda: plugins.setWallBcs(flags, vel)),
            ("solvePressure", lambda: plugins.solvePressure(vel, pres, flags)),
            ("extrapolateMACSimple", lambda: plugins.extrapolateMACSimple(flags, vel)),
            ("flipVelocityUpdate", lambda: plugins.flipVelocityUpdate(flags, vel, velOld, pp, pv, 0.97)),
        ]
        tot = {k: 0.0 for k, _ in ops}
        for it in range(reps + 1):
            for k, f in ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); f(); e1.record(); torch.cuda.synchronize()
                if it > 0:
                    tot[k] += e0.elapsed_time(e1)
            s.step()
        npart = pp.np
        print("S-flip %d^3, %d particles, %d steps: CG iterations last %s" % (n, npart, reps, plugins.lastCgStats()))
        for k, _ in ops:
            ms = tot[k] / reps
            print("  %-26s %8.3f ms   %8.1f Mparticles/s" % (k, ms, npart / ms / 1e3))
        print("  whole FLIP step            %8.3f ms   %8.2f Mcells/s" % (sum(tot.values()) / reps, n ** 3 / (sum(tot.values()) / reps) / 1e3))
    elif what == "dam":
        # ghost-fluid FLIP dam break with benchmark_dam.py's loop (tests/cases.py:run_dam_pkg), MF_RES = reference resolution
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        import cases
        res = int(os.environ.get("MF_RES", "160"))
        det = os.environ.get("MF_P2G_DET", "1") == "1"
        cases.run_dam_pkg(res, 2, deterministic=det)      # warm-up (allocations, sort scratch)
        plugins._timings.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        out = cases.run_dam_pkg(res, reps, deterministic=det)
        torch.cuda.synchronize()
        el = time.time() - t0
        gs = out["gs"]
        ncell = gs[0] * gs[1] * gs[2]
        print("dam break res %d: grid %s (%d cells), %d particles, %d steps: %.2f ms/step incl. set-up amortised, CG iterations %s"
              % (res, gs, ncell, out["pos"].shape[1], reps, el * 1e3 / reps, out["iters"]))
        plugins.Timings().display()
    elif what == "wavelet":
        # scenes/waveletTurbulence.py's loop (tests/cases.py:run_wavelet_scene_pkg), 3D, MF_RES = coarse resolution (scene default 80)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        import cases
        res = int(os.environ.get("MF_RES", "80"))
        cases.run_wavelet_scene_pkg(res, 3, 2)
        plugins._timings.clear()
        torch.cuda.synchronize()
        t0 = time.time()
        out = cases.run_wavelet_scene_pkg(res, 3, reps)
        torch.cuda.synchronize()
        el = time.time() - t0
        print("waveletTurbulence 3D res %d (xl grid %s), %d steps: %.2f ms/step incl. set-up amortised" % (res, out["xl_density"].shape[::-1], reps, el * 1e3 / reps))
        plugins.Timings().display()
    elif what == "advect":
        vel, dens = core.MACGrid(s), core.Grid(s)
        vel.from_numpy(np.ascontiguousarray(bench.synthetic_velocity(n, n, n).transpose(1, 2, 3, 0)))
        dens.from_numpy(bench.synthetic_density(n, n, n))
        vel0 = core.MACGrid(s)
        vel0.copyFrom(vel)
        plugins.advectSemiLagrange(flags, vel, dens, order=2)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        for _ in range(reps):
            plugins.advectSemiLagrange(flags, vel0, dens, order=2)
        ev[1].record()
        for _ in range(reps):
            vel.copyFrom(vel0)
            plugins.advectSemiLagrange(flags, vel, vel, order=2)
        ev[2].record()
        torch.cuda.synchronize()
        cells = dims[0] * dims[1] * dims[2]
        t_real, t_mac = ev[0].elapsed_time(ev[1]) * 1e3 / reps, ev[1].elapsed_time(ev[2]) * 1e3 / reps
        # algorithmic bytes per cell, SURVEY 8d: MacCormack Real 92, MAC 188 (+ 20 for the outflow-BC sweeps, + 24 for the copy here)
        print("MacCormack Real %s: %.1f us per call = %.2f TB/s of 92 B/cell (%.3f of 8 TB/s)" % (dims, t_real, 92 * cells / t_real / 1e6, 92 * cells / t_real / 8e6))
        print("MacCormack MAC  %s: %.1f us per call (incl. a 24 B/cell restore copy and the outflow-BC sweeps) = %.2f TB/s of 232 B/cell (%.3f of 8 TB/s)"
              % (dims, t_mac, 232 * cells / t_mac / 1e6, 232 * cells / t_mac / 8e6))


if __name__ == "__main__":
    main()
