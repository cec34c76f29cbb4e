#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: Mcells/s per step (advect + CG) on the 256^3 smoke configuration.

One "step" = the hot path of one smoke time step over the synthetic S-smoke input:
    restore velocity (device copy) -> advectSemiLagrange(density, order=2) -> advectSemiLagrange(vel, order=2)
    -> setWallBcs -> solvePressure(MIC-preconditioned CG, cgAccuracy 1e-3)
All inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task prompt)
with two extra objects: `roofline` (ApplyMatrix, the kernel BASELINE.json's north_star names: 28 B/cell algorithmic
bytes / HIP-event launch time, vs 8 TB/s) and `cpu_baseline` (the reference C++/OpenMP path -- oracle/_ref, or the
oracle port when the compiled reference is absent -- timed on the host cores on a bounded sample of the workload).

Launch: python bench.py --gpus 1 [--steps K --warmup W]  |  torchrun --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRID = 256
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
APPLY_MATRIX_BYTES_PER_CELL = 28  # flags 4 + src 4 + A0,Ai,Aj,Ak 16 read; dst 4 written
MIC_BYTES_PER_CELL = 56           # (flags 4, var1 4, Aprecond 4, Ai, Aj, Ak 12 read; dst 4 written) x 2 sweeps
CG_ITERATION_BYTES_PER_CELL = 144  # 96 (unpreconditioned, as structured by the reference) - 8 (copy) + 56 (MIC apply)
STEP_FIXED_BYTES_PER_CELL = 92 + 188 + 20 + 24 + 28 + 20 + 20 + 28 + 32   # once per step, SURVEY 8d table (see roofline_step below)
# what the kernels as built stream per cell: per PCG iteration ApplyMatrix on the packed byte (src 4, dst 4, byte 1),
# k_cg_axpy_r (residual r/w 8, tmp 4), forward sweep (var1 4, Aprecond 4, byte 1, dst 4), backward sweep with the fused dot (dst r/w 8,
# Aprecond 4, byte 1, residual 4), k_cg_update_search_x (search r/w 8, tmp 4, pressure r/w 8)
MOVED_ITERATION_BYTES_PER_CELL = 9 + 12 + 13 + 17 + 20
MOVED_FIXED_BYTES_PER_CELL = 76 + 160 + 24 + 28 + 20 + 20 + 21 + 28 + 36 + 32


def synthetic_velocity(sx, sy, sz, seed=7, vmax=2.0):
    """S-smoke: MAC velocity = discrete curl of a seeded band-limited vector potential living on the cell
    edges, so the field is discretely divergence-free like a projected simulation state; scaled to max |v| dt = vmax
    cells.  SoA [3][z][y][x]."""
    rng = np.random.default_rng(seed)
    z = np.arange(sz + 1, dtype=np.float32)[:, None, None]
    y = np.arange(sy + 1, dtype=np.float32)[None, :, None]
    x = np.arange(sx + 1, dtype=np.float32)[None, None, :]
    psi = []
    for c in range(3):
        p = np.zeros((sz + 1, sy + 1, sx + 1), np.float32)
        for _ in range(3):
            k = rng.uniform(0.5, 3.0, 3) * 2 * np.pi / np.array([sz, sy, sx])
            ph = rng.uniform(0, 2 * np.pi, 3)
            p += np.float32(rng.uniform(-1, 1)) * (np.sin(k[0] * z + ph[0]) * np.sin(k[1] * y + ph[1]) * np.sin(k[2] * x + ph[2])).astype(np.float32)
        psi.append(p)
    px, py, pz = psi
    v = np.empty((3, sz, sy, sx), np.float32)
    # u = d(psi_z)/dy - d(psi_y)/dz on x-faces, etc. (edge-centred potential -> face-centred curl)
    v[0] = (pz[:-1, 1:, :-1] - pz[:-1, :-1, :-1]) - (py[1:, :-1, :-1] - py[:-1, :-1, :-1])
    v[1] = (px[1:, :-1, :-1] - px[:-1, :-1, :-1]) - (pz[:-1, :-1, 1:] - pz[:-1, :-1, :-1])
    v[2] = (py[:-1, :-1, 1:] - py[:-1, :-1, :-1]) - (px[:-1, 1:, :-1] - px[:-1, :-1, :-1])
    v *= np.float32(vmax / max(np.abs(v).max(), 1e-9))
    return v


def synthetic_density(sx, sy, sz):
    z = np.arange(sz, dtype=np.float32)[:, None, None]
    y = np.arange(sy, dtype=np.float32)[None, :, None]
    x = np.arange(sx, dtype=np.float32)[None, None, :]
    r2 = ((x - sx * 0.5) / (0.2 * sx)) ** 2 + ((y - sy * 0.3) / (0.15 * sy)) ** 2 + ((z - sz * 0.5) / (0.2 * sz)) ** 2
    return np.exp(-r2).astype(np.float32)


def domain_flags(sx, sy, sz):
    f = np.full((sz, sy, sx), 1, np.int32)     # initDomain(0) + fillGrid(): 1-cell obstacle shell, fluid interior
    f[:, :, 0] = f[:, :, -1] = f[:, 0, :] = f[:, -1, :] = 2
    f[0] = f[-1] = 2
    return f


def cpu_baseline(dims, vel, dens, dt, gpu_iterations):
    """The reference CPU path timed on this host: ONE step of the same workload at full size (256^3: about half a minute on the GPU
    box's host cores -- the MIC sweeps and nothing else are serial in the reference).  kind 'reference' = the reference's own
    C++/OpenMP (oracle/_ref), 'port' = the plain-C oracle where the compiled reference did not travel."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    sx, sy, sz = dims
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    v = np.ascontiguousarray(vel)
    d = np.ascontiguousarray(dens)
    f = domain_flags(sx, sy, sz)
    cf = ctypes.c_float
    iters = None
    if util.have_ref():
        import cases  # noqa: F401
        kind = "reference"
        util.refcall("ref_set_wall_bcs", sx, sy, sz, f, v, None)
        t0 = time.time()
        util.refcall("ref_advect_semi_lagrange", sx, sy, sz, cf(dt), f, v, d, 0, 2, cf(1.0), 1, 2, 1)
        v2 = v.copy()
        util.refcall("ref_advect_semi_lagrange", sx, sy, sz, cf(dt), f, v, v2, 2, 2, cf(1.0), 1, 2, 1)
        util.refcall("ref_set_wall_bcs", sx, sy, sz, f, v2, None)
        p = np.zeros((sz, sy, sx), np.float32)
        util.refcall("ref_solve_pressure", sx, sy, sz, v2, p, f, cf(1e-3), None, None, None, None, cf(1e-4), cf(1.5), 1, 1, 0, 0, 0, None, cf(0.0), None)
        el = time.time() - t0
        it_note = ("%s CG iterations: the GPU's count on this input; the reference's own solvePressure does not return its count, the oracle's is "
                   "identical to the reference's (tests/test_oracle_vs_reference.py) and to the GPU's at this size (tests/test_gpu_fullsize.py config 2)"
                   % gpu_iterations)
    else:
        import cases
        from mantaflow_amd import _lib, core, plugins
        kind = "port"
        _lib.use_library(util.build_oracle(), "cpu")
        s = cases._mk_solver(dims, dt)
        fl, vg, dg, pg = core.FlagGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
        cases.soa_to_grid(fl, f); cases.soa_to_grid(vg, v); cases.soa_to_grid(dg, d)
        plugins.setWallBcs(fl, vg)
        t0 = time.time()
        plugins.advectSemiLagrange(fl, vg, dg, order=2)
        plugins.advectSemiLagrange(fl, vg, vg, order=2)
        plugins.setWallBcs(fl, vg)
        plugins.solvePressure(vg, pg, fl)
        el = time.time() - t0
        iters = plugins.lastCgStats().get("iterations")
        it_note = "%s CG iterations (counted)" % iters
        _lib.reset()
    cells = sx * sy * sz
    return {"value": round(cells / el / 1e6, 4), "unit": "Mcells/s", "cores": cores, "kind": kind,
            "sample": "1 full step (advect density+vel order 2, setWallBcs, solvePressure MIC-CG 1e-3) of the same synthetic "
                      "%dx%dx%d input, %.1f s, %s" % (sx, sy, sz, el, it_note)}


class OpClock(object):
    """HIP-event timing of the operators of a step (events on torch's current stream = the solver's launch stream).  Disabled it is a
    plain call, so the same step function serves the whole-step wall time and the per-operator pass."""
    def __init__(self, torch):
        self.torch, self.on, self.ev = torch, False, []

    def __call__(self, name, fn, *a, **k):
        if not self.on:
            return fn(*a, **k)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        self.ev.append((name, e0, e1))
        return r

    def per_op_ms(self, steps):
        self.torch.cuda.synchronize()
        tot = {}
        for name, e0, e1 in self.ev:
            tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
        self.ev = []
        return {k: v / steps for k, v in tot.items()}


def _timed(torch, step, steps, warm):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _op_rooflines(ops_ms, table):
    """per-operator objects: HIP-event time, SURVEY 8d's algorithmic bytes (B per particle x particles + B per cell x cells) / time
    against the 8 TB/s HBM peak"""
    out = {}
    for name, (nbytes, note) in table.items():
        if name not in ops_ms:
            continue
        ms = ops_ms[name]
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[name] = {"ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "achieved": round(gbs, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes": note}
    return out


def config3_sflip(torch, core, plugins, n=128, steps=5, warm=2, deterministic=True):
    """BASELINE config 3: the S-flip step of SURVEY 8d at 128^3 -- scenes/flip01_simple.py's loop, 8 particles per cell in the lower
    0.4 x 0.6 x 1.0 block of the box (flip.cpp:619-742, particle.h:458-550)."""
    from mantaflow_amd import scene
    s = core.Solver(gridSize=core.vec3(n, n, n), dim=3)
    s.timestep = 0.5
    plugins.setDeterministicP2G(deterministic)
    flags = core.FlagGrid(s)
    flags.initDomain(boundaryWidth=0)
    flags.updateFromLevelset(scene.Box(parent=s, p0=core.vec3(0, 0, 0), p1=core.vec3(0.4 * n, 0.6 * n, n)).computeLevelset())
    pp = core.BasicParticleSystem(s)
    scene.sampleFlagsWithParticles(flags, pp, 2, 0.2)
    pv = pp.create(core.PdataVec3)
    pv.from_numpy(np.random.default_rng(9832).normal(0, 0.5, (pp.np, 3)).astype(np.float32))
    vel, velOld, w, pres = core.MACGrid(s), core.MACGrid(s), core.VecGrid(s), core.Grid(s)
    clk = OpClock(torch)

    def flip_step():
        clk("advectInGrid", pp.advectInGrid, flags, vel, 2, deleteInObstacle=False)
        clk("mapPartsToMAC", plugins.mapPartsToMAC, flags, vel, velOld, pp, pv, w)
        clk("extrapolateMACFromWeight", plugins.extrapolateMACFromWeight, vel, w, distance=2)
        clk("markFluidCells", plugins.markFluidCells, pp, flags)
        clk("addGravity", plugins.addGravity, flags, vel, core.vec3(0, -0.002, 0))
        clk("setWallBcs", plugins.setWallBcs, flags, vel)
        clk("solvePressure", plugins.solvePressure, vel, pres, flags)
        clk("extrapolateMACSimple", plugins.extrapolateMACSimple, flags, vel)
        clk("flipVelocityUpdate", plugins.flipVelocityUpdate, flags, vel, velOld, pp, pv, 0.97)
        s.step()

    t = _timed(torch, flip_step, steps, warm)
    clk.on = True
    for _ in range(steps):
        flip_step()
    ops = clk.per_op_ms(steps)
    clk.on = False
    npart, cells = pp.np, n ** 3
    mode = "ordered gather, bit-exact" if deterministic else "fp32 atomics"
    out = {"workload": "flip01_simple.py loop, %d^3, %d particles (P2G: %s)" % (n, npart, mode), "ms_per_step": round(t * 1e3, 3),
           "Mcells_per_s": round(cells / t / 1e6, 2), "Mparticles_per_s": round(npart / t / 1e6, 2),
           "cg_iterations_last": plugins.lastCgStats().get("iterations"),
           "ops_ms": {k: round(v, 4) for k, v in ops.items()},
           "solvePressure_share": round(ops.get("solvePressure", 0.0) / max(sum(ops.values()), 1e-9), 3),
           "rooflines": _op_rooflines(ops, {
               "mapPartsToMAC": (40 * npart + 108 * cells, "SURVEY 8d: 40 B/particle (pos/flag 28 + pvel 12) + 108 B/cell (clear 24, stomp 24, divide 36, copy 24); 48 fp32 RMWs per particle on top"),
               "advectInGrid": (160 * npart, "SURVEY 8d: ~160 B/particle as the reference structures RK4 (4 x 28 gathers + pos 24 + scratch)"),
               "flipVelocityUpdate": (52 * npart, "SURVEY 8d: 52 B/particle (pos/flag 28 + pvel 12 read, 12 written; 48 gathers from cache)")})}
    if "mapPartsToMAC" in ops:
        out["rooflines"]["mapPartsToMAC"]["rmw_per_s"] = round(48 * npart / (ops["mapPartsToMAC"] * 1e-3) / 1e9, 2)
        out["rooflines"]["mapPartsToMAC"]["rmw_unit"] = "G fp32 accumulations/s (48 per particle; atomics in the atomic mode, ordered adds in the default mode)"
    plugins.setDeterministicP2G(True)
    return out


def dam_scene(core, plugins, scene, res):
    """Set-up of scenes/benchmark_dam.py (:33-92) at reference resolution `res` through the package API: returns the objects of its
    main loop and `step()` = one pass of that loop (:99-135, without the GUI-only second level-set pass)."""
    bnd, sres = 4, 2
    dx = 1.0 / sres
    sc = float(res)
    gs = [round(sc * 3.2) + bnd * 2, res * 3 + bnd * 2, res + bnd * 2]
    grav = -9.8 * sc
    FF, FO, FE = 1, 2, 4
    s = core.Solver(name="FLIP", gridSize=core.vec3(*gs), dim=3)
    s.cfl, s.frameLength, s.timestepMin = 1, 1.0 / 30.0, 0
    s.timestepMax = s.timestep = s.frameLength
    fl, V, Vold, P = s.create(core.FlagGrid), s.create(core.MACGrid), s.create(core.MACGrid), s.create(core.RealGrid)
    phiS, phi = s.create(core.LevelsetGrid), s.create(core.LevelsetGrid)
    isys, idx = s.create(core.ParticleIndexSystem), s.create(core.IntGrid)
    pp = s.create(core.BasicParticleSystem)
    pT, pV, pX = pp.create(core.PdataInt), pp.create(core.PdataVec3), pp.create(core.PdataVec3)
    fl.initDomain(bnd - 1)
    outer = s.create(scene.Box, p0=core.vec3(0), p1=core.vec3(*gs))
    inner = s.create(scene.Box, p0=core.vec3(bnd, bnd, bnd), p1=core.vec3(gs[0] - bnd, gs[1] - bnd, gs[0] - bnd))
    phiS.join(outer.computeLevelset())
    phiS.subtract(inner.computeLevelset())
    obs = s.create(scene.Box, center=core.vec3(0.744 * sc + bnd, 0.161 * 0.5 * sc + bnd, 0.5 * gs[2]),
                   size=core.vec3(0.161 * 0.5 * sc, 0.161 * 0.5 * sc, 0.403 * 0.5 * sc))
    obs.applyToGrid(grid=fl, value=FO, respectFlags=fl)
    phiS.join(obs.computeLevelset())
    dam = s.create(scene.Box, center=core.vec3(2.606 * sc + bnd, 0.275 * sc + bnd, 0.5 * sc + bnd),
                   size=core.vec3(1.228 * 0.5 * sc, 0.55 * 0.5 * sc, 0.5 * sc))
    dam.applyToGrid(grid=fl, value=FF, respectFlags=fl)
    scene.sampleShapeWithParticles(shape=dam, flags=fl, parts=pp, discretization=sres, randomness=0)
    pT.setConstRange(FF, 0, pp.pySize())
    g = core.vec3(0, grav, 0)
    st = {"iters": []}

    def step(clk=lambda name, fn, *a, **k: fn(*a, **k)):
        clk("mapPartsToMAC", plugins.mapPartsToMAC, vel=V, flags=fl, velOld=Vold, parts=pp, partVel=pV, ptype=pT, exclude=FE)
        s.adaptTimestep(V.getMaxAbs())
        plugins.addGravityNoScale(flags=fl, vel=V, gravity=g)
        clk("gridParticleIndex", plugins.gridParticleIndex, parts=pp, flags=fl, indexSys=isys, index=idx)
        clk("unionParticleLevelset", plugins.unionParticleLevelset, parts=pp, indexSys=isys, flags=fl, index=idx, phi=phi, radiusFactor=1.0)
        clk("extrapolateLsSimple", plugins.extrapolateLsSimple, phi=phi, distance=4, inside=True)
        if st.get("hook"):
            st["hook"]()
        plugins.setWallBcs(flags=fl, vel=V)
        clk("solvePressure", plugins.solvePressure, flags=fl, vel=V, pressure=P, cgAccuracy=1e-3, phi=phi)
        st["iters"].append(plugins.lastCgStats()["iterations"])
        plugins.setWallBcs(flags=fl, vel=V)
        clk("extrapolateMACSimple", plugins.extrapolateMACSimple, flags=fl, vel=V)
        clk("flipVelocityUpdate", plugins.flipVelocityUpdate, vel=V, velOld=Vold, flags=fl, parts=pp, partVel=pV, flipRatio=0.97, ptype=pT, exclude=FE)
        plugins.addForcePvel(vel=pV, a=g, dt=s.timestep, ptype=pT, exclude=FF)
        pp.getPosPdata(target=pX)
        clk("advectInGrid", pp.advectInGrid, flags=fl, vel=V, integrationMode=2, deleteInObstacle=False, ptype=pT, exclude=FE)
        plugins.eulerStep(parts=pp, vel=pV, ptype=pT, exclude=FF)
        pp.projectOutOfBnd(flags=fl, bnd=bnd + dx * 0.5, plane="xXyYzZ", ptype=pT)
        plugins.pushOutofObs(parts=pp, flags=fl, phiObs=phiS, thresh=dx * 0.5, ptype=pT)
        plugins.updateVelocityFromDeltaPos(parts=pp, vel=pV, x_prev=pX, dt=s.timestep, ptype=pT, exclude=FF)
        clk("markFluidCells+setPartType", lambda: (plugins.markFluidCells(parts=pp, flags=fl, ptype=pT),
                                                   plugins.setPartType(parts=pp, ptype=pT, mark=FF, stype=FE, flags=fl, cflag=FF),
                                                   plugins.markIsolatedFluidCell(flags=fl, mark=FE),
                                                   plugins.setPartType(parts=pp, ptype=pT, mark=FE, stype=FF, flags=fl, cflag=FE)))
        s.step()

    return dict(s=s, gs=gs, flags=fl, vel=V, velOld=Vold, pressure=P, phi=phi, phiObs=phiS, parts=pp, ptype=pT, pvel=pV, step=step, state=st)


DAM_RES = 116    # benchmark_dam.py's params['res'] for which its grid (3.2 res + 8) x (3 res + 8) x (res + 8) = 379 x 356 x 124 has 256^3 cells (16.73 M)


def config4_dam(torch, core, plugins, res=DAM_RES, steps=4, warm=2):
    """BASELINE config 4 on one GPU: the ghost-fluid FLIP dam break of scenes/benchmark_dam.py (main loop :99-135) on a grid of
    256^3 cells (res 116: 379 x 356 x 124)."""
    from mantaflow_amd import scene
    sc = dam_scene(core, plugins, scene, res)
    clk = OpClock(torch)
    step = lambda: sc["step"](clk)
    t = _timed(torch, step, steps, warm)
    clk.on = True
    for _ in range(steps):
        step()
    ops = clk.per_op_ms(steps)
    clk.on = False
    npart = sc["parts"].np
    cells = sc["gs"][0] * sc["gs"][1] * sc["gs"][2]
    out = {"workload": "benchmark_dam.py loop (ghost fluid, cgAccuracy 1e-3), res %d: grid %dx%dx%d = %d cells, %d particles, steps %d-%d"
                       % (res, sc["gs"][0], sc["gs"][1], sc["gs"][2], cells, npart, warm + 1, warm + steps),
           "ms_per_step": round(t * 1e3, 3), "Mcells_per_s": round(cells / t / 1e6, 2), "Mparticles_per_s": round(npart / t / 1e6, 2),
           "cg_iterations": sc["state"]["iters"][warm:warm + steps], "ops_ms": {k: round(v, 4) for k, v in ops.items()},
           "solvePressure_share": round(ops.get("solvePressure", 0.0) / max(t * 1e3, 1e-9), 3),
           "rooflines": _op_rooflines(ops, {
               "mapPartsToMAC": (40 * npart + 108 * cells, "SURVEY 8d: 40 B/particle + 108 B/cell"),
               "advectInGrid": (160 * npart, "SURVEY 8d: ~160 B/particle"),
               "flipVelocityUpdate": (52 * npart, "SURVEY 8d: 52 B/particle")})}
    return out


def config5_wavelet(torch, core, plugins, nc=256, up=2, steps=3, warm=2):
    """BASELINE config 5 on one GPU: the up-res loop of scenes/waveletTurbulence.py (:105-146) with a 256^3 coarse and a 512^3 fine
    grid; the fine-grid MacCormack advection (advection.cpp:25-42, 82-92, 242-268) is the HBM-bound part."""
    from mantaflow_amd import scene
    gs, xgs = core.vec3(nc, nc, nc), core.vec3(nc * up, nc * up, nc * up)
    sm, xl = core.Solver(gridSize=gs, dim=3), core.Solver(gridSize=xgs, dim=3)
    sm.timestep = xl.timestep = 1.5
    noise = scene.NoiseField(parent=sm, fixedSeed=265, loadFromFile=True)
    noise.posScale, noise.clamp, noise.clampNeg, noise.clampPos, noise.valScale, noise.valOffset, noise.timeAnim = core.vec3(20), True, 0, 2, 1, 0.075, 0.3
    source = scene.Cylinder(parent=sm, center=gs * core.vec3(0.3, 0.2, 0.5), radius=nc * 0.081, z=gs * core.vec3(0.081, 0, 0))
    sourceVel = scene.Cylinder(parent=sm, center=gs * core.vec3(0.3, 0.2, 0.5), radius=nc * 0.15, z=gs * core.vec3(0.15, 0, 0))
    xl_source = scene.Cylinder(parent=xl, center=xgs * core.vec3(0.3, 0.2, 0.5), radius=xgs.x * 0.081, z=xgs * core.vec3(0.081, 0, 0))
    xl_noise = scene.NoiseField(parent=xl, fixedSeed=265, loadFromFile=True)
    xl_noise.posScale, xl_noise.clamp, xl_noise.clampNeg, xl_noise.clampPos = noise.posScale, True, 0, 2
    xl_noise.valScale, xl_noise.valOffset, xl_noise.timeAnim = 1, 0.075, noise.timeAnim * up
    wl = []
    for k in range(3):
        f = scene.NoiseField(parent=xl, loadFromFile=True)
        f.posScale, f.timeAnim = core.vec3(int(1.0 * nc)) * (0.5 * 2.0 ** k), 0.1
        wl.append(f)
    flags, vel, dens, pres, energy = core.FlagGrid(sm), core.MACGrid(sm), core.Grid(sm), core.Grid(sm), core.Grid(sm)
    xfl, xvel, xdens, xw = core.FlagGrid(xl), core.MACGrid(xl), core.Grid(xl), core.Grid(xl)
    flags.initDomain(); flags.fillGrid(); plugins.setOpenBound(flags, 0, "Y", 16 | 4)
    xfl.initDomain(); xfl.fillGrid()
    velInflow = core.vec3(0.025, 0, 0) * float(nc)
    wlt = 0.4
    clk = OpClock(torch)

    def wavelet_step():
        clk("coarse: advect density+vel (MacCormack)", lambda: (plugins.advectSemiLagrange(flags, vel, dens, order=2),
                                                               plugins.advectSemiLagrange(flags, vel, vel, order=2)))
        scene.densityInflow(flags=flags, density=dens, noise=noise, shape=source, scale=1, sigma=0.5)
        sourceVel.applyToGrid(grid=vel, value=velInflow)
        plugins.setWallBcs(flags, vel)
        plugins.addBuoyancy(flags, dens, vel, core.vec3(0, -1e-3, 0))
        plugins.vorticityConfinement(vel, flags, strength=0.3)
        clk("coarse: solvePressure", plugins.solvePressure, vel, pres, flags, cgMaxIterFac=1.0, cgAccuracy=0.01)
        plugins.setWallBcs(flags, vel)
        clk("coarse: computeEnergy+computeWaveletCoeffs", lambda: (plugins.computeEnergy(flags, vel, energy), plugins.computeWaveletCoeffs(energy)))
        sm.step()
        clk("fine: interpolateGrid+interpolateMACGrid", lambda: (plugins.interpolateGrid(target=xw, source=energy),
                                                                plugins.interpolateMACGrid(source=vel, target=xvel)))
        clk("fine: applyNoiseVec3 x3", lambda: [plugins.applyNoiseVec3(flags=xfl, target=xvel, noise=wl[k], scale=wlt * 0.6 ** k, weight=xw) for k in range(3)])
        for _ in range(up):
            clk("fine: advectSemiLagrange(density, order 2)", plugins.advectSemiLagrange, xfl, xvel, xdens, order=2)
        clk("fine: densityInflow", scene.densityInflow, flags=xfl, density=xdens, noise=xl_noise, shape=xl_source, scale=1, sigma=0.5)
        xl.step()

    t = _timed(torch, wavelet_step, steps, warm)
    clk.on = True
    for _ in range(steps):
        wavelet_step()
    ops = clk.per_op_ms(steps)
    clk.on = False
    nf = (nc * up) ** 3
    fine_ms = sum(v for k, v in ops.items() if k.startswith("fine:"))
    mc = "fine: advectSemiLagrange(density, order 2)"
    out = {"workload": "waveletTurbulence.py loop, coarse %d^3 (MIC-CG 1e-2) + fine %d^3 (resampling, 3 noise octaves, %d MacCormack substeps)" % (nc, nc * up, up),
           "ms_per_step": round(t * 1e3, 2), "fine_part_ms": round(fine_ms, 2), "fine_Mcells_per_s": round(nf / t / 1e6, 1),
           "cg_iterations_last": plugins.lastCgStats().get("iterations"), "ops_ms": {k: round(v, 4) for k, v in ops.items()},
           "rooflines": _op_rooflines({mc: ops[mc] / up}, {mc: (92 * nf, "SURVEY 8d: MacCormack Real 92 B/cell (20 + 20 + correct 20 + clamp 32), per call at %d^3" % (nc * up))})}
    return out


def other_configs(torch, core, plugins):
    """BASELINE configs 3, 4 and 5 on one GPU, a few steps each, so that the driver's record carries a number and per-operator
    rooflines for them as well (the headline `value` stays config 2)."""
    out = {}
    out["config3_sflip128"] = config3_sflip(torch, core, plugins)
    torch.cuda.empty_cache()
    out["config4_dam256"] = config4_dam(torch, core, plugins)
    torch.cuda.empty_cache()
    out["config5_wavelet512"] = config5_wavelet(torch, core, plugins)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=GRID)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short single-GPU runs of BASELINE configs 3, 4 and 5")
    ap.add_argument("--slab", action="store_true", help="N=1 only: run the z-slab code path with one rank (overhead check)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    assert world == a.gpus, "launch with --nproc-per-node equal to --gpus"

    from mantaflow_amd import _lib, core, plugins
    lib = _lib.get()
    assert lib.backend == "hip"
    n = a.grid
    dt = 1.0

    if world > 1 or a.slab:
        from mantaflow_amd import slab
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        result = slab.bench_slab_step(n, dt, a.steps, a.warmup, rank, world)
        if rank == 0:
            # the roofline object of the line, for the kernel north_star names, on what one rank computes per PCG iteration: its owned
            # planes + one ghost plane per interior face (HIP events on the launch stream, after the timed region)
            nz = n // world + (1 if world > 1 else 0) + (1 if world > 2 else 0)
            s2 = core.Solver(gridSize=core.vec3(n, n, nz), dim=3)
            fl2 = core.FlagGrid(s2); fl2.initDomain(); fl2.fillGrid()
            g = [core.Grid(s2) for _ in range(6)]
            lib.call("mf_make_laplace_matrix", n, n, nz, fl2.ptr, g[0].ptr, g[1].ptr, g[2].ptr, g[3].ptr, None, s2.stream)
            g[4].from_numpy(np.random.default_rng(1234).uniform(-1, 1, (nz, n, n)).astype(np.float32))
            us = ctypes.c_double()
            lib.call("mf_time_apply_matrix", n, n, nz, fl2.ptr, g[5].ptr, g[4].ptr, g[0].ptr, g[1].ptr, g[2].ptr, g[3].ptr, 200, ctypes.byref(us), s2.stream)
            gbs = APPLY_MATRIX_BYTES_PER_CELL * n * n * nz / (us.value * 1e-6) / 1e9
            result["roofline"] = {"kernel": "k_apply_matrix_v5 (ApplyMatrix, conjugategrad.h:118-133) on one rank's PCG window %dx%dx%d" % (n, n, nz),
                                  "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                  "traffic": None, "avg_launch_us": round(us.value, 2),
                                  "algorithmic_bytes_per_launch": APPLY_MATRIX_BYTES_PER_CELL * n * n * nz}
            del g, fl2, s2
    else:
        s = core.Solver(gridSize=core.vec3(n, n, n), dim=3)
        s.timestep = dt
        flags = core.FlagGrid(s); flags.initDomain(); flags.fillGrid()
        vel, vel0, dens, pres = core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
        v_np = synthetic_velocity(n, n, n)
        d_np = synthetic_density(n, n, n)
        vel0.from_numpy(np.ascontiguousarray(v_np.transpose(1, 2, 3, 0)))
        plugins.setWallBcs(flags, vel0)
        dens.from_numpy(d_np)
        iters = []

        def step():
            vel.copyFrom(vel0)
            plugins.advectSemiLagrange(flags, vel, dens, order=2)
            plugins.advectSemiLagrange(flags, vel, vel, order=2)
            plugins.setWallBcs(flags, vel)
            plugins.solvePressure(vel, pres, flags)
            iters.append(plugins.lastCgStats()["iterations"])

        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        # the interpreter's cyclic collector otherwise lands a ~35 ms full collection inside some timed steps (seen in the
        # kernel trace as one idle gap between two steps): collect now, and keep the survivors out of later passes
        import gc
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        result = {"elapsed": el, "cells": n ** 3, "cg_iterations": iters[-a.steps:] if a.steps else []}

        # ---- roofline of the ApplyMatrix stencil (HIP events on the launch stream, inside the library) ----
        A0, Ai, Aj, Ak, src, dst = (core.Grid(s) for _ in range(6))
        lib.call("mf_make_laplace_matrix", n, n, n, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, s.stream)
        src.from_numpy(np.random.default_rng(1234).uniform(-1, 1, (n, n, n)).astype(np.float32))
        us = ctypes.c_double()
        lib.call("mf_time_apply_matrix", n, n, n, flags.ptr, dst.ptr, src.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, 200, ctypes.byref(us), s.stream)
        gbs = APPLY_MATRIX_BYTES_PER_CELL * n ** 3 / (us.value * 1e-6) / 1e9
        # the variant the PCG loop runs for this matrix: flags + Ai + Aj + Ak packed into one byte per cell by mf_mic_init
        # (13 B per cell: src, dst, A0, packed byte); reported next to the general kernel, never instead of it
        apk = core.Grid(s)
        lib.call("mf_mic_init", n, n, n, flags.ptr, apk.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        us_pk = ctypes.c_double()
        lib.call("mf_time_apply_matrix_packed", n, n, n, flags.ptr, dst.ptr, src.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, 200, ctypes.byref(us_pk), s.stream)
        del apk
        # HBM traffic needs the PMC counters (rocprofv3 --pmc, separate passes): not measurable from inside this process, so
        # the number is read from the committed counter summary and labelled with its source
        traffic, traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "apply_matrix_traffic.json")
        if os.path.exists(tp) and n == GRID:
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
                traffic_src = "profiles/apply_matrix_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
            except Exception:
                traffic = None
        result["roofline"] = {"kernel": "k_apply_matrix_v5 (ApplyMatrix, conjugategrad.h:118-133)", "bound": "hbm",
                              "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                              "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(us.value, 2),
                              "algorithmic_bytes_per_launch": APPLY_MATRIX_BYTES_PER_CELL * n ** 3,
                              "pcg_variant": {"kernel": "k_apply_matrix_v5<PACKED> (flags + Ai + Aj + Ak + the integer diagonal A0 as one byte per cell, exact for a MakeLaplaceMatrix system)",
                                              "avg_launch_us": round(us_pk.value, 2), "bytes_per_cell": 9,
                                              "achieved": round(9 * n ** 3 / (us_pk.value * 1e-6) / 1e9, 1),
                                              "frac": round(9 * n ** 3 / (us_pk.value * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}
        # ---- the kernel that takes most of the step: the two MIC(0) substitution sweeps (conjugategrad.cpp:135-159).
        # forward reads flags, rhs, Ai, Aj, Ak, Aprecond, dst and writes dst (32 B/cell), backward does not read rhs (28 B/cell)
        ap = core.Grid(s)
        lib.call("mf_mic_init", n, n, n, flags.ptr, ap.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        for _ in range(3):
            lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        for _ in range(50):
            lib.call("mf_mic_apply", n, n, n, flags.ptr, dst.ptr, src.ptr, ap.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        mic_us = e0.elapsed_time(e1) * 1e3 / 50
        mic_bytes = MIC_BYTES_PER_CELL * n ** 3
        mic_traffic, mic_traffic_src = None, None
        tp = os.path.join(ROOT, "profiles", "mic_traffic.json")
        if os.path.exists(tp) and n == GRID:
            try:
                mic_traffic = json.load(open(tp)).get("hbm_bytes_per_apply")     # PMC counters, see profiles/README.md
                mic_traffic_src = "profiles/mic_traffic.json (rocprofv3 --pmc passes, not measured in this run)"
            except Exception:
                mic_traffic = None
        result["roofline_mic"] = {"kernel": "k_mic_rows<1> + k_mic_rows<2> (ApplyPreconditionModifiedIncompCholesky2, conjugategrad.cpp:135-159)",
                                  "bound": "dependency chain (serial sweep in the reference), hbm if it were free",
                                  "achieved": round(mic_bytes / (mic_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(mic_bytes / (mic_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "avg_apply_us": round(mic_us, 1),
                                  "algorithmic_bytes_per_apply": mic_bytes, "traffic": mic_traffic, "traffic_source": mic_traffic_src,
                                  "achieved_on_measured_traffic": None if not mic_traffic else round(mic_traffic / (mic_us * 1e-6) / 1e9, 1),
                                  "frac_on_measured_traffic": None if not mic_traffic else round(mic_traffic / (mic_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                  "note": "achieved / frac price the reference's unfused structure (SURVEY 8d: 56 B per cell and apply) at the measured "
                                          "time -- a speed against the reference-structure roofline; the packed sweeps move fewer bytes (traffic), "
                                          "and *_on_measured_traffic is the bandwidth they really draw"}
        # ---- the whole step against the HBM roofline: SURVEY 8d's algorithmic bytes of every operator of the step / ms_per_step
        its = float(np.mean(result["cg_iterations"])) if result["cg_iterations"] else 0.0
        step_bytes = (STEP_FIXED_BYTES_PER_CELL + its * CG_ITERATION_BYTES_PER_CELL) * n ** 3
        step_gbs = step_bytes / (el / max(a.steps, 1)) / 1e9
        result["roofline_step"] = {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(step_gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_step": int(step_bytes),
                                   "bytes_per_cell": {"fixed (MacCormack Real 92 + MAC 188 + outflow sweeps 20 + velocity restore 24 + setWallBcs 28 "
                                                      "+ MakeRhs 20 + MakeLaplaceMatrix 20 + MIC init 28 + correctVelocity 32)": STEP_FIXED_BYTES_PER_CELL,
                                                      "per CG iteration (ApplyMatrix 28 + 2 dot 16 + 2 axpy 24 + max-norm 8 + search update 12 + MIC apply 56)": CG_ITERATION_BYTES_PER_CELL,
                                                      "mean CG iterations": round(its, 1)},
                                   "moved_bytes_per_step": int((MOVED_FIXED_BYTES_PER_CELL + its * MOVED_ITERATION_BYTES_PER_CELL) * n ** 3),
                                   "moved_bytes_per_cell": {"per CG iteration as built (ApplyMatrix on packed bytes 9 + residual update and max-norm 12 + forward sweep 13 "
                                                            "+ backward sweep with dot 17 + search / pressure update 20)": MOVED_ITERATION_BYTES_PER_CELL,
                                                            "fixed as built (MacCormack Real 76 + MAC 160 + restore 24 + setWallBcs 28 + MakeRhs 20 + MakeLaplaceMatrix 20 "
                                                            "+ pack 21 + MIC init 28 + doInit copies 36 + correctVelocity 32)": MOVED_FIXED_BYTES_PER_CELL},
                                   "frac_on_moved_bytes": round((MOVED_FIXED_BYTES_PER_CELL + its * MOVED_ITERATION_BYTES_PER_CELL) * n ** 3
                                                                / (el / max(a.steps, 1)) / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "achieved / frac price the reference's structure (SURVEY 8d) at the measured step time; frac_on_moved_bytes uses the "
                                           "bytes the fused / packed kernels stream (own model, DESIGN.md section 3)"}
        del A0, Ai, Aj, Ak, src, dst, ap
        if not a.no_other_configs and n == GRID:
            del flags, vel, vel0, dens, pres, s
            torch.cuda.empty_cache()
            result["other_configs"] = other_configs(torch, core, plugins)
        if not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline((n, n, n), v_np, d_np, dt, result["cg_iterations"][-1] if result["cg_iterations"] else None)

    if world > 1:
        t = torch.tensor([result["elapsed"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        result["elapsed"] = float(t.item())
    if rank == 0:
        el = result["elapsed"]
        cells = n ** 3
        line = {
            "metric": "Mcells/s per step (advect+CG+FLIP), %d^3 grid" % n,
            "value": round(cells * a.steps / el / 1e6, 2), "unit": "Mcells/s", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(el / max(a.steps, 1) * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d^3 smoke step on %d x MI355X: advectSemiLagrange(density, order 2) + "
                                   "advectSemiLagrange(vel, order 2) + setWallBcs + solvePressure (MIC-CG, cgAccuracy 1e-3)" % (n, a.gpus),
                       "grid": [n, n, n], "cg_iterations_per_step": result.get("cg_iterations"),
                       "parallelism": "single GPU" if a.gpus == 1 else ("z-slab x%d, 1-plane halo p2p per PCG iteration + one RCCL all-gather per reduction point "
                                                                          "(16-byte rows, summed in rank order); N = 1 of this code path is `bench.py --slab`" % a.gpus)},
        }
        if "roofline" in result:
            line["roofline"] = result["roofline"]
        if "roofline_mic" in result:
            line["roofline_mic"] = result["roofline_mic"]
        if "roofline_step" in result:
            line["roofline_step"] = result["roofline_step"]
        if "other_configs" in result:
            line["other_configs"] = result["other_configs"]
        if "cpu_baseline" in result:
            line["cpu_baseline"] = result["cpu_baseline"]
        if "notes" in result:
            line["config"]["notes"] = result["notes"]
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
