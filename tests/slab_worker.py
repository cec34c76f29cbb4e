"""Worker for the z-slab tests: run under torch.distributed.run (gloo on CPU with the oracle library, or -- on the
GPU box -- gloo with every rank on cuda:0 through the HIP library).  Writes the owned planes of its results."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import util  # noqa: E402
from mantaflow_amd import _lib  # noqa: E402


def flip_case(out, dims):
    """S-flip step on slabs: advectInGrid(RK4) + migration, mapPartsToMAC with the reverse halo, solvePressure,
    flipVelocityUpdate.  Particles carry their global index (pid) so that the ranks' results can be put back in order."""
    from mantaflow_amd import core, slab
    NX, NY, NZ = dims
    dom = slab.SlabDomain((NX, NY, NZ), slab.required_ghost(2.0))
    s = dom.solver
    s.timestep = 0.8
    flags_g = util.make_flags(NX, NY, NZ, 61, obstacles=True, empty_top=True)
    vel_g = util.smooth_vel(NX, NY, NZ, 62, 2.2)
    pos, pflag, pvel = util.make_particles(flags_g, 2, 63, include_border=False)
    pid = np.arange(pos.shape[1], dtype=np.int32)
    flags, vel, velOld, pres, w = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.VecGrid(s)
    dom.scatter_global(flags, flags_g); dom.scatter_global(vel, vel_g)
    sp = slab.SlabParticles(dom)
    pv, pi = sp.create(core.PdataVec3), sp.create(core.PdataInt)
    sp.scatter_global(pos, pflag, [(pv, pvel), (pi, pid)])
    n0 = sp.np
    moved = slab.advectInGrid(dom, sp, flags, vel, 2, deleteInObstacle=False)
    adv = sp.gather()
    slab.mapPartsToMAC(dom, flags, vel, velOld, sp, pv, w)
    p2g_vel, p2g_w = dom.gather_owned(vel).copy(), dom.gather_owned(w).copy()
    dom.exchange(vel, 1)
    slab.setWallBcs(dom, flags, vel)
    st = {}
    slab.solvePressure(dom, vel, pres, flags, cgAccuracy=1e-5, stats=st)
    slab.flipVelocityUpdate(dom, flags, vel, velOld, sp, pv, 0.97)
    fin = sp.gather()
    np.savez(out + ".%d.npz" % dom.comm.rank, n0=n0, moved=moved, adv_pos=adv["pos"], adv_flag=adv["flag"], adv_pid=adv["pdata1"][0],
             p2g_vel=p2g_vel, p2g_w=p2g_w, pid=fin["pdata1"][0], pvel=fin["pdata0"], iters=st["iterations"])


def liquid_case(out, dims):
    """two steps of the flip01_simple.py loop on slabs: advectInGrid, mapPartsToMAC, extrapolateMACFromWeight, markFluidCells,
    addGravity, setWallBcs, solvePressure, extrapolateMACSimple, flipVelocityUpdate"""
    from mantaflow_amd import core, slab
    NX, NY, NZ = dims
    dom = slab.SlabDomain((NX, NY, NZ), slab.required_ghost(2.0))
    s = dom.solver
    s.timestep = 0.8
    flags_g = np.full((NZ, NY, NX), 4, np.int32)                    # empty box with a 1-cell wall
    flags_g[:, :, 0] = flags_g[:, :, -1] = flags_g[:, 0, :] = flags_g[:, -1, :] = 2
    flags_g[0] = flags_g[-1] = 2
    fluid = np.zeros_like(flags_g, bool)
    fluid[1:NZ - 1, 1:int(0.6 * NY), 1:int(0.5 * NX)] = True        # a liquid block spanning every slab
    flags_g[fluid] = 1
    pos, pflag, pvel = util.make_particles(flags_g, 2, 73, vel_scale=0.6, deleted_frac=0.0, include_border=False)
    pid = np.arange(pos.shape[1], dtype=np.int32)
    flags, vel, velOld, pres, w = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.Grid(s), core.VecGrid(s)
    dom.scatter_global(flags, flags_g)
    sp = slab.SlabParticles(dom)
    pv, pi = sp.create(core.PdataVec3), sp.create(core.PdataInt)
    sp.scatter_global(pos, pflag, [(pv, pvel), (pi, pid)])
    rec, moved, iters = {}, 0, []
    for step in range(2):
        moved += slab.advectInGrid(dom, sp, flags, vel, 2, deleteInObstacle=False)
        slab.mapPartsToMAC(dom, flags, vel, velOld, sp, pv, w)
        slab.extrapolateMACFromWeight(dom, vel, w, distance=2)
        slab.markFluidCells(dom, sp, flags)
        if step == 0:
            rec["flags0"], rec["vel_ext0"] = dom.gather_owned(flags).copy(), dom.gather_owned(vel).copy()
        slab.addGravity(dom, flags, vel, core.vec3(0, -0.01, 0))
        dom.exchange(vel, 1)
        slab.setWallBcs(dom, flags, vel)
        st = {}
        slab.solvePressure(dom, vel, pres, flags, cgAccuracy=1e-6, stats=st)
        iters.append(st["iterations"])
        dom.exchange(vel, 1)
        slab.setWallBcs(dom, flags, vel)
        slab.extrapolateMACSimple(dom, flags, vel, distance=4)
        slab.flipVelocityUpdate(dom, flags, vel, velOld, sp, pv, 0.97)
        s.step()
    fin = sp.gather()
    np.savez(out + ".%d.npz" % dom.comm.rank, moved=moved, pid=fin["pdata1"][0], pos=fin["pos"], pvel=fin["pdata0"], flags=dom.gather_owned(flags),
             vel=dom.gather_owned(vel), iters=np.array(iters), **rec)


def dam_case(out, res, steps=3):
    """the main loop of scenes/benchmark_dam.py (ghost-fluid FLIP dam break, BASELINE config 4) on z-slabs, flowing along z: P2G with
    ptype exclusion, domain-wide adaptTimestep, particle level set with a one-cell particle halo, extrapolateLsSimple on the
    ghosts, ghost-fluid solvePressure, FLIP update, RK4 advection + eulerStep / projectOutOfBnd / pushOutofObs, migration,
    markFluidCells / setPartType / markIsolatedFluidCell.  The set-up runs on a plain whole-domain solver IN THE SAME PROCESS
    (its slab window stays (0, 0) while the slab solver's is (lo, NZ))."""
    import cases
    from mantaflow_amd import core, plugins, slab
    FF, FE = 1, 4
    bnd, gs = cases.dam_geometry(res, zflow=True)[:2]
    dom = slab.SlabDomain(gs, slab.required_ghost(2.0))
    s = dom.solver
    s.cfl, s.frameLength, s.timestepMin = 1, 1.0 / 30, 0
    s.timestepMax = s.timestep = s.frameLength
    # ---- set-up on the whole domain, then scattered ----
    sg = core.Solver(name="setup", gridSize=core.vec3(*gs), dim=3)
    flg, phiSg, ppg = sg.create(core.FlagGrid), sg.create(core.LevelsetGrid), sg.create(core.BasicParticleSystem)
    pTg = ppg.create(core.PdataInt)
    cases.dam_setup(sg, flg, phiSg, ppg, pTg, res, zflow=True)
    flags_g, phiS_g = cases.grid_to_soa(flg), cases.grid_to_soa(phiSg)
    pos, pflag, ptype = cases._ppos(ppg), ppg.get_flags(), pTg.data[:ppg.np].cpu().numpy().copy()
    del flg, phiSg, ppg, pTg, sg
    pid = np.arange(pos.shape[1], dtype=np.int32)
    fl, V, Vold, P = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.Grid(s)
    phiS, phi = core.LevelsetGrid(s), core.LevelsetGrid(s)
    isys, idx = core.ParticleIndexSystem(s), core.IntGrid(s)
    dom.scatter_global(fl, flags_g); dom.scatter_global(phiS, phiS_g)
    sp = slab.SlabParticles(dom)
    pT, pV, pX, pI = sp.create(core.PdataInt), sp.create(core.PdataVec3), sp.create(core.PdataVec3), sp.create(core.PdataInt)
    sp.scatter_global(pos, pflag, [(pT, ptype), (pI, pid)])
    n0 = sp.np
    grav = core.vec3(0, -9.8 * res, 0)
    rec, iters, dts, moved = {}, [], [], 0
    for step in range(steps):
        slab.mapPartsToMAC(dom, fl, V, Vold, sp, pV, ptype=pT, exclude=FE)
        slab.adaptTimestep(dom, V)
        dts.append(s.timestep)
        plugins.addGravityNoScale(flags=fl, vel=V, gravity=grav)
        slab.unionParticleLevelset(dom, sp, fl, phi, indexSys=isys, index=idx, radiusFactor=1.0)
        slab.extrapolateLsSimple(dom, phi, distance=4, inside=True)
        if step in (0, steps - 1):
            rec["phi_ls%d" % (0 if step == 0 else 1)] = dom.gather_owned(phi).copy()
        dom.exchange(V, 1)
        slab.setWallBcs(dom, fl, V)
        st = {}
        slab.solvePressure(dom, V, P, fl, cgAccuracy=1e-6, phi=phi, stats=st)
        iters.append(st["iterations"])
        dom.exchange(V, 1)
        slab.setWallBcs(dom, fl, V)
        slab.extrapolateMACSimple(dom, fl, V)
        slab.flipVelocityUpdate(dom, fl, V, Vold, sp, pV, 0.97, ptype=pT, exclude=FE)
        plugins.addForcePvel(vel=pV, a=grav, dt=s.timestep, ptype=pT, exclude=FF)
        sp.pp.getPosPdata(target=pX)
        slab.advectInGrid(dom, sp, fl, V, 2, deleteInObstacle=False, ptype=pT, exclude=FE, migrate=False)
        plugins.eulerStep(parts=sp.pp, vel=pV, ptype=pT, exclude=FF)
        sp.pp.projectOutOfBnd(flags=fl, bnd=bnd + 0.25, plane="xXyYzZ", ptype=pT)
        plugins.pushOutofObs(parts=sp.pp, flags=fl, phiObs=phiS, thresh=0.25, ptype=pT)
        plugins.updateVelocityFromDeltaPos(parts=sp.pp, vel=pV, x_prev=pX, dt=s.timestep, ptype=pT, exclude=FF)
        moved += sp.migrate()
        slab.markFluidCells(dom, sp, fl, ptype=pT)
        plugins.setPartType(parts=sp.pp, ptype=pT, mark=FF, stype=FE, flags=fl, cflag=FF)
        slab.markIsolatedFluidCell(dom, fl, FE)
        plugins.setPartType(parts=sp.pp, ptype=pT, mark=FE, stype=FF, flags=fl, cflag=FE)
        s.step()
    fin = sp.gather()
    np.savez(out + ".%d.npz" % dom.comm.rank, n0=n0, moved=moved, pid=fin["pdata3"][0], pos=fin["pos"], ptype=fin["pdata0"][0],
             pvel=fin["pdata1"], flags=dom.gather_owned(fl), vel=dom.gather_owned(V), pres=dom.gather_owned(P),
             phi=dom.gather_owned(phi), iters=np.array(iters), dts=np.array(dts), **rec)


def wavelet_case(out, gs, steps=2):
    """the loop of scenes/waveletTurbulence.py:105-146 (BASELINE config 5) on z-slabs: the coarse solver `sm` and the 2x finer `xl`
    on the same z-ranges, preceded by one pass of the up-res pipeline on synthetic input (no solve -> bit-exact check)"""
    import cases
    from mantaflow_amd import core, plugins, scene, slab
    U, WS = cases.WLT_UPRES, cases.WLT_STRENGTH
    inp = cases.wavelet_inputs(gs)                           # plain whole-domain solvers in the same process
    dom = slab.SlabDomain(gs, slab.required_ghost(2.0))
    xdom = slab.refine(dom, U)
    sm, xl = dom.solver, xdom.solver
    sm.timestep = xl.timestep = cases.WLT_DT
    o = cases.wavelet_objects(sm, xl, gs)
    fl, V, D, P, E = core.FlagGrid(sm), core.MACGrid(sm), core.Grid(sm), core.Grid(sm), core.Grid(sm)
    xfl, xV, xD, xW = core.FlagGrid(xl), core.MACGrid(xl), core.Grid(xl), core.Grid(xl)
    dom.scatter_global(fl, inp["flags"]); xdom.scatter_global(xfl, inp["xl_flags"])

    def upres_pass():
        # coarse ghosts (energy, vel) are current: the resampled fields are right on every fine plane, ghosts included
        slab.interpolateGrid(dom, xdom, target=xW, source=E)
        slab.interpolateMACGrid(dom, xdom, source=V, target=xV)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt1"], scale=WS * 1.0, weight=xW)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt2"], scale=WS * 0.6, weight=xW)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt3"], scale=WS * 0.6 * 0.6, weight=xW)
        for _ in range(U):
            xdom.exchange(xD)
            slab.advectSemiLagrange(xdom, xfl, xV, xD, order=2)

    def energy_step():
        dom.exchange(V, 1)
        plugins.computeEnergy(flags=fl, vel=V, energy=E)
        slab.computeWaveletCoeffs(dom, E)
        dom.exchange(E)
        dom.exchange(V)

    rec = {}
    dom.scatter_global(V, inp["vel_syn"]); xdom.scatter_global(xD, inp["xl_dens_syn"])
    slab.setWallBcs(dom, fl, V)
    energy_step()
    upres_pass()
    rec["energy0"], rec["xl_vel0"], rec["xl_dens0"] = dom.gather_owned(E).copy(), xdom.gather_owned(xV).copy(), xdom.gather_owned(xD).copy()
    V.clear(); xD.clear(); xV.clear(); E.clear()
    iters = []
    for t in range(steps):
        dom.exchange(D); dom.exchange(V)
        slab.advectSemiLagrange(dom, fl, V, D, order=2)
        slab.advectSemiLagrange(dom, fl, V, V, order=2)
        scene.densityInflow(flags=fl, density=D, noise=o["noise"], shape=o["source"], scale=1, sigma=0.5)
        o["sourceVel"].applyToGrid(grid=V, value=o["velInflow"])
        slab.setWallBcs(dom, fl, V)
        dom.exchange(D, 1)
        slab.addBuoyancy(dom, fl, D, V, core.vec3(0, -1e-3, 0))
        slab.vorticityConfinement(dom, V, fl, strength=0.3)
        if t == 0:
            rec["vel_pre0"], rec["dens_pre0"] = dom.gather_owned(V).copy(), dom.gather_owned(D).copy()
        dom.exchange(V, 1)
        st = {}
        slab.solvePressure(dom, V, P, fl, cgAccuracy=1e-6, cgMaxIterFac=2.0, stats=st)
        iters.append(st["iterations"])
        slab.setWallBcs(dom, fl, V)
        energy_step()
        sm.step()
        upres_pass()
        scene.densityInflow(flags=xfl, density=xD, noise=o["xl_noise"], shape=o["xl_source"], scale=1, sigma=0.5)
        xl.step()
    np.savez(out + ".%d.npz" % dom.comm.rank, dens=dom.gather_owned(D), vel=dom.gather_owned(V), energy=dom.gather_owned(E),
             xl_dens=xdom.gather_owned(xD), xl_vel=xdom.gather_owned(xV), iters=np.array(iters), **rec)


def main():
    out, backend = sys.argv[1], sys.argv[2]
    dims = tuple(int(v) for v in sys.argv[3].split("x"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    if backend == "oracle":
        _lib.use_library(util.build_oracle(), "cpu")
    else:
        torch.cuda.set_device(0)
        _lib.get()
    if len(sys.argv) > 4 and sys.argv[4] in ("flip", "liquid", "dam", "wavelet"):
        if sys.argv[4] == "wavelet":
            wavelet_case(out, dims)
        elif sys.argv[4] == "dam":
            dam_case(out, dims[0])        # "dims" carries the resolution of the dam case: RESx0x0
        else:
            (flip_case if sys.argv[4] == "flip" else liquid_case)(out, dims)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    from mantaflow_amd import core, slab
    NX, NY, NZ = dims
    dom = slab.SlabDomain((NX, NY, NZ), slab.required_ghost(2.0))
    s = dom.solver
    s.timestep = 0.9
    flags_g = util.make_flags(NX, NY, NZ, 51, obstacles=True, empty_top=True)
    vel_g = util.smooth_vel(NX, NY, NZ, 52, 2.0)
    vel_g[2] *= 0.95
    dens_g = util.rand_real((NZ, NY, NX), 53)
    flags, vel, dens, pres = core.FlagGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
    dom.scatter_global(flags, flags_g); dom.scatter_global(vel, vel_g); dom.scatter_global(dens, dens_g)
    slab.advectSemiLagrange(dom, flags, vel, dens, order=2)
    slab.advectSemiLagrange(dom, flags, vel, vel, order=2)
    dom.exchange(vel, 1)
    slab.setWallBcs(dom, flags, vel)
    vel_adv = dom.gather_owned(vel).copy()
    st = {}
    slab.solvePressure(dom, vel, pres, flags, cgAccuracy=1e-4, stats=st)
    # divergence of the projected field on the owned planes
    dom.exchange(vel, 1)
    rhs = core.Grid(s)
    s.lib.call("mf_make_rhs", NX, NY, dom.LZ, flags.ptr, rhs.ptr, vel.ptr, None, None, None, None, None, 0.0, 1e-4, None, None, s.stream)
    # a plain whole-domain solver created AFTER the slab domain, in the same process and thread: its grids are the whole domain
    # (window (0, 0)) whatever the slab solver's window is -- its advection must be the single-device result on every rank
    import cases
    from mantaflow_amd import plugins
    sg = core.Solver(gridSize=core.vec3(NX, NY, NZ), dim=3)
    sg.timestep = 0.9
    fg, vg, dg = core.FlagGrid(sg), core.MACGrid(sg), core.Grid(sg)
    cases.soa_to_grid(fg, flags_g); cases.soa_to_grid(vg, vel_g); cases.soa_to_grid(dg, dens_g)
    plugins.advectSemiLagrange(fg, vg, dg, order=2)
    slab.setWallBcs(dom, flags, vel)                     # a slab call in between (idempotent here) ...
    plugins.advectSemiLagrange(fg, vg, dg, order=1)      # ... and the plain solver again
    np.savez(out + ".%d.npz" % dom.comm.rank, z0=dom.z0, z1=dom.z1, dens=dom.gather_owned(dens), vel_adv=vel_adv,
             vel=dom.gather_owned(vel), pres=dom.gather_owned(pres), div=dom.gather_owned(rhs), iters=st["iterations"], res=st["residual"],
             plain_dens=cases.grid_to_soa(dg), mic_blocking=np.array(st["mic_blocking"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
