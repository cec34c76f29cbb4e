"""Parity cases: each case builds seeded inputs (numpy), runs them through
   - `ref`  : the compiled reference behind oracle/ref_shim.cpp (host arrays), and
   - `pkg`  : the package's host layer (mantaflow_amd.core/plugins) on whatever ABI library is active
              (the oracle restatement in CPU tests, the HIP library in -m gpu tests), or a raw util.Impl,
and returns dicts of numpy outputs that the tests compare bit for bit."""
import ctypes

import numpy as np
import torch

import util
from util import P, refcall

SIZES_3D = [(12, 10, 9), (16, 16, 16), (20, 13, 11)]
SIZE_2D = (24, 18, 1)


def soa_to_grid(g, arr):
    """numpy SoA [ncomp][sz][sy][sx] (or [sz][sy][sx]) -> package grid storage"""
    g.data.copy_(torch.from_numpy(np.ascontiguousarray(arr).reshape(-1)).to(g.data.device))
    return g


def grid_to_soa(g):
    a = g.data.detach().cpu().numpy().copy()
    return a.reshape((g._ncomp, g.sz, g.sy, g.sx)) if g._ncomp == 3 else a.reshape((g.sz, g.sy, g.sx))


# ---------------------------------------------------------------------------------------------------------
# kernel-level cases (raw ABI through util.Impl)
# ---------------------------------------------------------------------------------------------------------
def laplace_inputs(dims, seed, fractions=False):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=(seed % 2 == 1))
    fr = None
    if fractions:
        fr = np.random.default_rng(seed + 5).uniform(0, 1, (3, sz, sy, sx)).astype(np.float32)
    return flags, fr


def run_laplace_impl(impl, dims, flags, fr):
    sx, sy, sz = dims
    A = [impl.dev(np.zeros((sz, sy, sx), np.float32)) for _ in range(4)]
    impl.call("mf_make_laplace_matrix", sx, sy, sz, impl.dev(flags), A[0], A[1], A[2], A[3], impl.dev(fr), None)
    impl.sync()
    return [impl.host(a) for a in A]


def run_laplace_ref(dims, flags, fr):
    sx, sy, sz = dims
    A = [np.zeros((sz, sy, sx), np.float32) for _ in range(4)]
    refcall("ref_make_laplace_matrix", sx, sy, sz, flags, A[0], A[1], A[2], A[3], fr)
    return A


def system_inputs(dims, seed):
    """flags + Laplace matrix (built by the reference when available, else by the oracle) + random src"""
    sx, sy, sz = dims
    flags, _ = laplace_inputs(dims, seed)
    if util.have_ref():
        A = run_laplace_ref(dims, flags, None)
    else:
        A = run_laplace_impl(util.Impl("oracle"), dims, flags, None)
    src = util.rand_real((sz, sy, sx), seed + 11)
    return flags, A, src


def run_apply_matrix_impl(impl, dims, flags, A, src):
    sx, sy, sz = dims
    dst = impl.dev(np.full((sz, sy, sx), 7.0, np.float32))
    impl.call("mf_apply_matrix", sx, sy, sz, impl.dev(flags), dst, impl.dev(src), *[impl.dev(a) for a in A], None)
    impl.sync()
    return impl.host(dst)


def run_apply_matrix_ref(dims, flags, A, src):
    sx, sy, sz = dims
    dst = np.full((sz, sy, sx), 7.0, np.float32)
    refcall("ref_apply_matrix", sx, sy, sz, flags, dst, src, *A)
    return dst


def run_mic_impl(impl, dims, flags, A, var1, blocking=None, then_plain=None):
    """blocking = (rows_j, cells_x): mf_mic_init_blocked (the caller has cut A accordingly).  then_plain = another (uncut) system:
    after the blocked one, an ordinary mf_mic_init / mf_mic_apply on it in the same process must be the uncut reference sweep
    (the blocking belongs to the system it was given with)"""
    sx, sy, sz = dims
    ap = impl.dev(np.full((sz, sy, sx), 3.0, np.float32))
    dA = [impl.dev(a) for a in A]
    f = impl.dev(flags)
    if blocking is None:
        impl.call("mf_mic_init", sx, sy, sz, f, ap, dA[0], dA[1], dA[2], dA[3], None)
    else:
        impl.call("mf_mic_init_blocked", sx, sy, sz, f, ap, dA[0], dA[1], dA[2], dA[3], int(blocking[0]), int(blocking[1]), None)
    dst = impl.dev(np.zeros((sz, sy, sx), np.float32))
    impl.call("mf_mic_apply", sx, sy, sz, f, dst, impl.dev(var1), ap, dA[1], dA[2], dA[3], None)
    impl.sync()
    if then_plain is not None:
        return impl.host(ap), impl.host(dst), run_mic_impl(impl, dims, flags, then_plain, var1)
    return impl.host(ap), impl.host(dst)


def run_mic_ref(dims, flags, A, var1):
    sx, sy, sz = dims
    ap = np.full((sz, sy, sx), 3.0, np.float32)
    refcall("ref_mic_init", sx, sy, sz, flags, ap, *A)
    dst = np.zeros((sz, sy, sx), np.float32)
    refcall("ref_mic_apply", sx, sy, sz, flags, dst, var1, ap, *A)
    return ap, dst


def cg_rhs(dims, flags, seed):
    sx, sy, sz = dims
    rhs = util.rand_real((sz, sy, sx), seed + 23)
    rhs[(flags & util.FLUID) == 0] = 0
    return rhs


def run_cg_impl(impl, dims, flags, A, rhs, pc, accuracy, maxIter, useL2=0):
    sx, sy, sz = dims
    z = lambda: impl.dev(np.zeros((sz, sy, sx), np.float32))
    dst, residual, search, tmp, ap = z(), z(), z(), z(), z()
    out = (ctypes.c_float * 3)()
    impl.call("mf_cg_solve", sx, sy, sz, impl.dev(flags), dst, impl.dev(rhs), residual, search, tmp,
              *[impl.dev(a) for a in A], ap, pc, accuracy, maxIter, useL2, out, None)
    impl.sync()
    return impl.host(dst), (int(out[0]), float(out[1]), float(out[2]))


def run_cg_ref(dims, flags, A, rhs, pc, accuracy, maxIter, useL2=0):
    sx, sy, sz = dims
    z = lambda: np.zeros((sz, sy, sx), np.float32)
    dst, residual, search, tmp, ap = z(), z(), z(), z(), z()
    out = np.zeros(3, np.float32)
    refcall("ref_cg_solve", sx, sy, sz, flags, dst, rhs, residual, search, tmp, *A, ap, pc, ctypes.c_float(accuracy), maxIter, useL2, out)
    return dst, (int(out[0]), float(out[1]), float(out[2]))


# ---------------------------------------------------------------------------------------------------------
# plugin-level cases (package host layer vs reference plugins)
# ---------------------------------------------------------------------------------------------------------
def _mk_solver(dims, dt=1.0):
    from mantaflow_amd import core
    sx, sy, sz = dims
    s = core.Solver(gridSize=core.vec3(sx, sy, sz), dim=3 if sz > 1 else 2)
    s.timestep = dt
    return s


def pressure_inputs(dims, seed, liquid=False):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=liquid)
    vel = util.rand_vel(sx, sy, sz, seed + 3, 0.5)
    phi = None
    if liquid:
        # a smooth level set that is negative in fluid cells and positive in empty ones
        zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
        phi = (yy - (2 * sy) // 3 + 0.3 + 0.2 * np.sin(xx * 0.7)).astype(np.float32)
    return flags, vel, phi


def pressure_optional_terms(dims, seed, which):
    """numpy inputs for the optional terms of MakeRhs / MakeLaplaceMatrix (pressure.cpp:32-84, conjugategrad.h:154-187):
    perCellCorr (Real), fractions (MAC, fill fractions in [0.1, 1] with some faces closed), obvel (MAC), curv (Real)"""
    sx, sy, sz = dims
    rng = np.random.default_rng(seed)
    out = {}
    if "perCellCorr" in which:
        out["perCellCorr"] = rng.uniform(-0.2, 0.2, (sz, sy, sx)).astype(np.float32)
    if "fractions" in which:
        fr = rng.uniform(0.1, 1.0, (3, sz, sy, sx)).astype(np.float32)
        fr[rng.uniform(size=fr.shape) < 0.3] = 1.0
        fr[rng.uniform(size=fr.shape) < 0.05] = 0.0
        out["fractions"] = fr
    if "obvel" in which:
        out["obvel"] = util.rand_vel(sx, sy, sz, seed + 11, 0.3)
    if "curv" in which:
        out["curv"] = rng.uniform(-0.5, 0.5, (sz, sy, sx)).astype(np.float32)
    return out


def run_solve_pressure_pkg(dims, flags, vel, phi, extra=None, **kw):
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    fl, v, p = core.FlagGrid(s), core.MACGrid(s), core.Grid(s)
    soa_to_grid(fl, flags); soa_to_grid(v, vel)
    ph = soa_to_grid(core.LevelsetGrid(s), phi) if phi is not None else None
    rr = core.Grid(s)
    keep = []
    for name, arr in (extra or {}).items():
        g = soa_to_grid((core.MACGrid if arr.ndim == 4 else core.Grid)(s), arr)
        keep.append(g)
        kw[name] = g
    plugins.solvePressure(v, p, fl, phi=ph, retRhs=rr, **kw)
    s.sync()
    return dict(vel=grid_to_soa(v), pressure=grid_to_soa(p), rhs=grid_to_soa(rr), stats=plugins.lastCgStats())


def run_solve_pressure_ref(dims, flags, vel, phi, cgAccuracy=1e-3, cgMaxIterFac=1.5, enforceCompatibility=False,
                           useL2Norm=False, zeroPressureFixing=False, gfClamp=1e-4, extra=None, surfTens=0.0):
    sx, sy, sz = dims
    v = vel.copy()
    p = np.zeros((sz, sy, sx), np.float32)
    rr = np.zeros((sz, sy, sx), np.float32)
    ex = extra or {}
    refcall("ref_solve_pressure", sx, sy, sz, v, p, flags, ctypes.c_float(cgAccuracy), phi, ex.get("perCellCorr"), ex.get("fractions"),
            ex.get("obvel"), ctypes.c_float(gfClamp), ctypes.c_float(cgMaxIterFac), 1, 1, int(enforceCompatibility), int(useL2Norm),
            int(zeroPressureFixing), ex.get("curv"), ctypes.c_float(surfTens), rr)
    return dict(vel=v, pressure=p, rhs=rr)


# (terms given, liquid case?) -- every optional input of solvePressure (pressure.cpp:482-497) at least once, alone and combined
PRESSURE_OPTIONAL_CASES = [(("perCellCorr",), False), (("fractions",), False), (("fractions", "obvel"), False),
                           (("curv",), True), (("perCellCorr", "fractions", "obvel", "curv"), True)]


def advect_inputs(dims, seed, vmax=2.5, outflow=False):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=True, outflow=outflow)
    vel = util.smooth_vel(sx, sy, sz, seed + 1, vmax)
    return flags, vel


def run_advect_pkg(dims, dt, flags, vel, field, kind, **kw):
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims, dt)
    fl, v = core.FlagGrid(s), core.MACGrid(s)
    soa_to_grid(fl, flags); soa_to_grid(v, vel)
    G = {0: core.Grid, 1: core.VecGrid, 2: core.MACGrid, 3: core.LevelsetGrid}[kind]
    g = soa_to_grid(G(s), field)
    plugins.advectSemiLagrange(fl, v, g, **kw)
    s.sync()
    return grid_to_soa(g)


def run_advect_ref(dims, dt, flags, vel, field, kind, order=1, strength=1.0, orderSpace=1, clampMode=2, orderTrace=1):
    sx, sy, sz = dims
    g = field.copy()
    refcall("ref_advect_semi_lagrange", sx, sy, sz, ctypes.c_float(dt), flags, vel, g, kind, order, ctypes.c_float(strength),
            orderSpace, clampMode, orderTrace)
    return g


def _mk_parts(s, pos, pflag):
    from mantaflow_amd import core
    pp = core.BasicParticleSystem(s)
    pp.resizeAll(pos.shape[1])
    for c in range(3):
        pp.pos[c * pp.cap:c * pp.cap + pp.np] = torch.from_numpy(pos[c]).to(pp.pos.device)
    pp.flag[:pp.np] = torch.from_numpy(pflag).to(pp.flag.device)
    return pp


def _parts_get(pp):
    a = pp.pos.detach().cpu().numpy()
    return np.stack([a[c * pp.cap:c * pp.cap + pp.np] for c in range(3)], 0), pp.flag[:pp.np].cpu().numpy().copy()


def _pd_vec3(s, pp, arr):
    from mantaflow_amd import core
    pd = pp.create(core.PdataVec3)
    for c in range(3):
        pd.data[c * pd.cap:c * pd.cap + pp.np] = torch.from_numpy(arr[c]).to(pd.data.device)
    return pd


def _pd_get(pd, n):
    a = pd.data.detach().cpu().numpy()
    return np.stack([a[c * pd.cap:c * pd.cap + n] for c in range(pd._ncomp)], 0) if pd._ncomp == 3 else a[:n].copy()


def run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype=None, exclude=0, deterministic=True):
    """mapPartsToMAC -> (vel, velOld, weight); mapMACToParts; flipVelocityUpdate; mapPartsToGrid(+Vec3); mapGridToParts"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    plugins.setDeterministicP2G(deterministic)
    fl = soa_to_grid(core.FlagGrid(s), flags)
    pp = _mk_parts(s, pos, pflag)
    pv = _pd_vec3(s, pp, pvel)
    pt = None
    if ptype is not None:
        pt = pp.create(core.PdataInt)
        pt.data[:pp.np] = torch.from_numpy(ptype).to(pt.data.device)
    out = {}
    v, vo, w = core.MACGrid(s), core.MACGrid(s), core.VecGrid(s)
    soa_to_grid(v, util.rand_vel(*dims, 99)); soa_to_grid(w, util.rand_vel(*dims, 98))     # must be cleared by the plugin
    plugins.mapPartsToMAC(fl, v, vo, pp, pv, w, ptype=pt, exclude=exclude)
    out["p2g_vel"], out["p2g_velOld"], out["p2g_weight"] = grid_to_soa(v), grid_to_soa(vo), grid_to_soa(w)
    gv, gvo = soa_to_grid(core.MACGrid(s), vel), soa_to_grid(core.MACGrid(s), velOld)
    pv2 = _pd_vec3(s, pp, pvel)
    plugins.mapMACToParts(fl, gv, pp, pv2, ptype=pt, exclude=exclude)
    out["pic"] = _pd_get(pv2, pp.np)
    pv3 = _pd_vec3(s, pp, pvel)
    plugins.flipVelocityUpdate(fl, gv, gvo, pp, pv3, 0.97, ptype=pt, exclude=exclude)
    out["flip"] = _pd_get(pv3, pp.np)
    tgt = core.Grid(s)
    ps = pp.create(core.PdataReal)
    ps.data[:pp.np] = torch.from_numpy(pvel[0]).to(ps.data.device)
    plugins.mapPartsToGrid(fl, tgt, pp, ps)
    out["p2g_real"] = grid_to_soa(tgt)
    tv = core.VecGrid(s)
    plugins.mapPartsToGridVec3(fl, tv, pp, pv)
    out["p2g_vec3"] = grid_to_soa(tv)
    pr = pp.create(core.PdataReal)
    plugins.mapGridToParts(soa_to_grid(core.Grid(s), vel[0]), pp, pr)
    out["g2p_real"] = _pd_get(pr, pp.np)
    pv4 = _pd_vec3(s, pp, pvel)
    plugins.mapGridToPartsVec3(soa_to_grid(core.VecGrid(s), vel), pp, pv4)
    out["g2p_vec3"] = _pd_get(pv4, pp.np)
    s.sync()
    plugins.setDeterministicP2G(True)
    return out


def run_flip_ref(dims, flags, vel, velOld, pos, pflag, pvel, ptype=None, exclude=0):
    sx, sy, sz = dims
    n = pos.shape[1]
    out = {}
    v, vo, w = util.rand_vel(*dims, 99), np.zeros((3, sz, sy, sx), np.float32), util.rand_vel(*dims, 98)
    refcall("ref_map_parts_to_mac", sx, sy, sz, flags, v, vo, w, n, n, pos, pflag, pvel, ptype, exclude)
    out["p2g_vel"], out["p2g_velOld"], out["p2g_weight"] = v, vo, w
    pv2 = pvel.copy()
    refcall("ref_map_mac_to_parts", sx, sy, sz, flags, vel, n, n, pos, pflag, pv2, ptype, exclude)
    out["pic"] = pv2
    pv3 = pvel.copy()
    refcall("ref_flip_velocity_update", sx, sy, sz, flags, vel, velOld, n, n, pos, pflag, pv3, ctypes.c_float(0.97), ptype, exclude)
    out["flip"] = pv3
    tgt = np.zeros((sz, sy, sx), np.float32)
    refcall("ref_map_parts_to_grid", sx, sy, sz, 1, flags, tgt, n, n, pos, pflag, np.ascontiguousarray(pvel[0]))
    out["p2g_real"] = tgt
    tv = np.zeros((3, sz, sy, sx), np.float32)
    refcall("ref_map_parts_to_grid", sx, sy, sz, 3, flags, tv, n, n, pos, pflag, pvel)
    out["p2g_vec3"] = tv
    pr = np.zeros(n, np.float32)
    refcall("ref_map_grid_to_parts", sx, sy, sz, 1, np.ascontiguousarray(vel[0]), n, n, pos, pflag, pr)
    out["g2p_real"] = pr
    pv4 = pvel.copy()
    refcall("ref_map_grid_to_parts", sx, sy, sz, 3, vel, n, n, pos, pflag, pv4)
    out["g2p_vec3"] = pv4
    return out


def _sorted_parts(P, F):
    """the reference's compress() moves tail particles into the holes (particle.h:614-633): compare as sets"""
    o = np.lexsort((P[2], P[1], P[0]))
    return np.ascontiguousarray(P[:, o]), F[o]


def outflow_inputs(dims, seed):
    """open (outflow) +-y faces as setOpenBound makes them, a random phi / density, particles everywhere"""
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=False)
    flags[:, :2, 1:-1] = 16 | 4
    flags[:, -2:, 1:-1] = 16 | 1            # outflow cells still marked fluid: resetOutflow must turn them empty
    phi, real = util.rand_real((sz, sy, sx), seed + 1), util.rand_real((sz, sy, sx), seed + 2)
    rng = np.random.default_rng(seed + 3)
    n = 4000
    pos = np.stack([rng.uniform(-0.5, sx + 0.5, n), rng.uniform(-0.5, sy + 0.5, n),
                    rng.uniform(-0.5, sz + 0.5, n) if sz > 1 else np.full(n, 0.5)], 0).astype(np.float32)
    pflag = np.zeros(n, np.int32)
    pflag[rng.random(n) < 0.05] = 1 << 10
    return flags, phi, real, pos, pflag


def run_outflow_pkg(dims, flags, phi, real, pos, pflag):
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    fl = soa_to_grid(core.FlagGrid(s), flags)
    p, r = soa_to_grid(core.Grid(s), phi), soa_to_grid(core.Grid(s), real)
    pp = _mk_parts(s, pos, pflag)
    plugins.resetOutflow(flags=fl, phi=p, parts=pp, real=r)
    s.sync()
    P, F = _parts_get(pp)
    keep = (F & (1 << 10)) == 0
    out = {"flags": grid_to_soa(fl), "phi": grid_to_soa(p), "real": grid_to_soa(r)}
    out["pos"], out["pflag"] = _sorted_parts(P[:, keep], F[keep])
    fl2 = soa_to_grid(core.FlagGrid(s), flags)
    plugins.resetOutflow(flags=fl2)
    out["flags_only"] = grid_to_soa(fl2)
    return out


def run_outflow_ref(dims, flags, phi, real, pos, pflag):
    sx, sy, sz = dims
    n = pos.shape[1]
    f, p, r, P, F = flags.copy(), phi.copy(), real.copy(), pos.copy(), pflag.copy()
    m = ctypes.c_int64(0)
    refcall("ref_reset_outflow", sx, sy, sz, f, p, r, n, n, P, F, ctypes.byref(m))
    out = {"flags": f, "phi": p, "real": r}
    out["pos"], out["pflag"] = _sorted_parts(P[:, :m.value], F[:m.value])
    f2 = flags.copy()
    refcall("ref_reset_outflow", sx, sy, sz, f2, None, None, 0, 0, None, None, None)
    out["flags_only"] = f2
    return out


def run_diffusion_pkg(dims, flags, real, vel, alpha=0.4, fac=1.0, acc=1e-5):
    """cgSolveDiffusion on a Real grid and on a MAC grid"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    fl = soa_to_grid(core.FlagGrid(s), flags)
    r, v = soa_to_grid(core.Grid(s), real), soa_to_grid(core.MACGrid(s), vel)
    plugins.cgSolveDiffusion(fl, r, alpha, fac, acc)
    it_r = plugins.lastCgStats()["iterations"]
    plugins.cgSolveDiffusion(fl, v, alpha, fac, acc)
    s.sync()
    return {"real": grid_to_soa(r), "mac": grid_to_soa(v), "iters": np.array([it_r, plugins.lastCgStats()["iterations"]])}


def run_diffusion_ref(dims, flags, real, vel, alpha=0.4, fac=1.0, acc=1e-5):
    sx, sy, sz = dims
    r, v = real.copy(), vel.copy()
    cf = ctypes.c_float
    refcall("ref_cg_solve_diffusion", sx, sy, sz, flags, r, 1, cf(alpha), cf(fac), cf(acc))
    refcall("ref_cg_solve_diffusion", sx, sy, sz, flags, v, 3, cf(alpha), cf(fac), cf(acc))
    return {"real": r, "mac": v}


def apic_inputs(dims, seed, per_cell=3, include_border=False):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=True)
    vel = util.rand_vel(sx, sy, sz, seed + 1)
    pos, pflag, pvel = util.make_particles(flags, per_cell, seed + 2, include_border=include_border)
    rng = np.random.default_rng(seed + 3)
    cp = [rng.normal(0, 0.3, pvel.shape).astype(np.float32) for _ in range(3)]
    return flags, vel, pos, pflag, pvel, cp


def run_apic_pkg(dims, flags, vel, pos, pflag, pvel, cp, ptype=None, exclude=0, with_mass=True):
    """apicMapPartsToMAC -> (vel, mass); apicMapMACGridToParts -> (pvel, cpx, cpy, cpz)"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    fl = soa_to_grid(core.FlagGrid(s), flags)
    pp = _mk_parts(s, pos, pflag)
    pv = _pd_vec3(s, pp, pvel)
    c = [_pd_vec3(s, pp, x) for x in cp]
    pt = None
    if ptype is not None:
        pt = pp.create(core.PdataInt)
        pt.data[:pp.np] = torch.from_numpy(ptype).to(pt.data.device)
    out = {}
    v, m = core.MACGrid(s), core.MACGrid(s)
    soa_to_grid(v, util.rand_vel(*dims, 99)); soa_to_grid(m, util.rand_vel(*dims, 98))     # must be cleared by the plugin
    plugins.apicMapPartsToMAC(fl, v, pp, pv, c[0], c[1], c[2], mass=m if with_mass else None, ptype=pt, exclude=exclude)
    out["apic_vel"] = grid_to_soa(v)
    if with_mass:
        out["apic_mass"] = grid_to_soa(m)
    gv = soa_to_grid(core.MACGrid(s), vel)
    plugins.apicMapMACGridToParts(pv, c[0], c[1], c[2], pp, gv, fl, ptype=pt, exclude=exclude)
    out["apic_pvel"] = _pd_get(pv, pp.np)
    for q, nm in enumerate(("cpx", "cpy", "cpz")):
        out["apic_" + nm] = _pd_get(c[q], pp.np)
    s.sync()
    return out


def run_apic_ref(dims, flags, vel, pos, pflag, pvel, cp, ptype=None, exclude=0):
    sx, sy, sz = dims
    n = pos.shape[1]
    out = {}
    v, m = util.rand_vel(*dims, 99), util.rand_vel(*dims, 98)
    refcall("ref_apic_map_parts_to_mac", sx, sy, sz, flags, v, m, n, n, pos, pflag, pvel, cp[0], cp[1], cp[2], ptype, exclude)
    out["apic_vel"], out["apic_mass"] = v, m
    pv, c = pvel.copy(), [x.copy() for x in cp]
    refcall("ref_apic_map_mac_to_parts", sx, sy, sz, flags, vel, n, n, pos, pflag, pv, c[0], c[1], c[2], ptype, exclude)
    out["apic_pvel"], out["apic_cpx"], out["apic_cpy"], out["apic_cpz"] = pv, c[0], c[1], c[2]
    return out


def run_advect_parts_pkg(dims, dt, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle=True, skipNew=False,
                         ptype=None, exclude=0):
    from mantaflow_amd import core
    s = _mk_solver(dims, dt)
    fl, v = soa_to_grid(core.FlagGrid(s), flags), soa_to_grid(core.MACGrid(s), vel)
    pp = _mk_parts(s, pos, pflag)
    pt = None
    if ptype is not None:
        pt = pp.create(core.PdataInt)
        pt.data[:pp.np] = torch.from_numpy(ptype).to(pt.data.device)
    pp.advectInGrid(fl, v, mode, deleteInObstacle=deleteInObstacle, stopInObstacle=stopInObstacle, skipNew=skipNew,
                    ptype=pt, exclude=exclude)
    s.sync()
    a = pp.pos.detach().cpu().numpy()
    return np.stack([a[c * pp.cap:c * pp.cap + pp.np] for c in range(3)], 0), pp.flag[:pp.np].cpu().numpy().copy()


def run_advect_parts_ref(dims, dt, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle=True, skipNew=False,
                         ptype=None, exclude=0):
    sx, sy, sz = dims
    p, f = pos.copy(), pflag.copy()
    n = p.shape[1]
    refcall("ref_advect_in_grid", sx, sy, sz, flags, vel, n, n, p, f, ctypes.c_float(dt), mode, int(deleteInObstacle),
            int(stopInObstacle), int(skipNew), ptype, exclude)
    return p, f


def flipglue_inputs(dims, seed):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=True)
    pos, pflag, pvel = util.make_particles(flags, 2, seed + 1)
    # liquid-style flags: everything non-obstacle starts empty; particles mark the fluid
    fl0 = np.where(flags & util.OBS, flags, util.EMPTY).astype(np.int32)
    vel = util.rand_vel(sx, sy, sz, seed + 2)
    return fl0, pos, pflag, pvel, vel


def run_flipglue_pkg(dims, fl0, pos, pflag, pvel, vel, phiObs=None):
    """mapPartsToMAC -> extrapolateMACFromWeight -> markFluidCells -> extrapolateMACSimple (flip01_simple.py:49-63)"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims)
    plugins.setDeterministicP2G(True)
    fl = soa_to_grid(core.FlagGrid(s), fl0)
    pp = _mk_parts(s, pos, pflag)
    pv = _pd_vec3(s, pp, pvel)
    v, vo, w = core.MACGrid(s), core.MACGrid(s), core.VecGrid(s)
    plugins.mapPartsToMAC(fl, v, vo, pp, pv, w)
    plugins.extrapolateMACFromWeight(v, w, distance=2)
    out = {"efw_vel": grid_to_soa(v), "efw_weight": grid_to_soa(w)}
    ph = soa_to_grid(core.Grid(s), phiObs) if phiObs is not None else None
    plugins.markFluidCells(pp, fl, phiObs=ph)
    out["flags"] = grid_to_soa(fl)
    v2 = soa_to_grid(core.MACGrid(s), vel)
    plugins.extrapolateMACSimple(fl, v2, distance=3)
    out["ems_vel"] = grid_to_soa(v2)
    v3 = soa_to_grid(core.MACGrid(s), vel)
    plugins.extrapolateMACSimple(fl, v3, distance=2, intoObs=True)
    out["ems_into"] = grid_to_soa(v3)
    s.sync()
    plugins.setDeterministicP2G(True)
    return out


def run_flipglue_ref(dims, fl0, pos, pflag, pvel, vel, phiObs=None):
    sx, sy, sz = dims
    n = pos.shape[1]
    v, vo, w = np.zeros((3, sz, sy, sx), np.float32), np.zeros((3, sz, sy, sx), np.float32), np.zeros((3, sz, sy, sx), np.float32)
    refcall("ref_map_parts_to_mac", sx, sy, sz, fl0, v, vo, w, n, n, pos, pflag, pvel, None, 0)
    refcall("ref_extrapolate_mac_from_weight", sx, sy, sz, v, w, 2)
    out = {"efw_vel": v, "efw_weight": w}
    fl = fl0.copy()
    refcall("ref_mark_fluid_cells", sx, sy, sz, fl, n, n, pos, pflag, None, 0, phiObs)
    out["flags"] = fl
    v2 = vel.copy()
    refcall("ref_extrapolate_mac_simple", sx, sy, sz, fl, v2, 3, 0)
    out["ems_vel"] = v2
    v3 = vel.copy()
    refcall("ref_extrapolate_mac_simple", sx, sy, sz, fl, v3, 2, 1)
    out["ems_into"] = v3
    return out


# ---- free-surface / particle maintenance pieces of benchmark_dam.py (SURVEY 8f-2 leftovers, 8f-3) ----
def surface_inputs(dims, seed):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=True)
    pos, pflag, pvel = util.make_particles(flags, 2, seed + 1)
    rng = np.random.default_rng(seed + 2)
    n = pos.shape[1]
    # a few particles outside the grid and near the walls (projectOutOfBnd / gridParticleIndex skip paths)
    k = max(n // 50, 3)
    pos[:, :k] = rng.uniform(-1.5, max(dims) + 1.5, (3, k)).astype(np.float32)
    if sz == 1:
        pos[2] = 0.5
    ptype = rng.choice(np.array([1, 4, 1, 1], np.int32), n).astype(np.int32)
    phiObs = (util.rand_real((sz, sy, sx), seed + 3) * 2).astype(np.float32)
    phi = (util.rand_real((sz, sy, sx), seed + 4) * 3).astype(np.float32)
    xprev = (pos + rng.normal(0, 0.3, pos.shape)).astype(np.float32)
    # a flag field with isolated fluid cells
    fiso = np.where(flags & util.OBS, flags, util.EMPTY).astype(np.int32)
    inner = (slice(1, -1) if sz > 1 else slice(None), slice(1, -1), slice(1, -1))
    m = rng.random(fiso.shape) < 0.3
    sel = np.zeros_like(m)
    sel[inner] = m[inner]
    fiso[sel & (fiso == util.EMPTY)] = util.FLUID
    return dict(flags=flags, pos=pos, pflag=pflag, pvel=pvel, ptype=ptype, phiObs=phiObs, phi=phi, xprev=xprev, fiso=fiso)


def _pd_int(s, pp, arr):
    from mantaflow_amd import core
    pd = pp.create(core.PdataInt)
    pd.data[:pp.np] = torch.from_numpy(arr).to(pd.data.device)
    return pd


def _ppos(pp):
    a = pp.pos.detach().cpu().numpy()
    return np.stack([a[c * pp.cap:c * pp.cap + pp.np] for c in range(3)], 0)


def run_surface_pkg(dims, I, dt=0.35):
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims, dt)
    out = {}
    fl = soa_to_grid(core.FlagGrid(s), I["flags"])
    # projectOutOfBnd, pushOutofObs
    pp = _mk_parts(s, I["pos"], I["pflag"])
    pt = _pd_int(s, pp, I["ptype"])
    pp.projectOutOfBnd(fl, 1.25, "xXyYzZ", pt, 4)
    out["project"] = _ppos(pp)
    pp = _mk_parts(s, I["pos"], I["pflag"])
    pp.projectOutOfBnd(fl, 2.5, "xYz")
    out["project_xYz"] = _ppos(pp)
    pp = _mk_parts(s, I["pos"], I["pflag"])
    pt = _pd_int(s, pp, I["ptype"])
    plugins.pushOutofObs(pp, fl, soa_to_grid(core.Grid(s), I["phiObs"]), shift=0.05, thresh=0.25, ptype=pt, exclude=4)
    out["push"] = _ppos(pp)
    # gridParticleIndex + unionParticleLevelset + extrapolateLsSimple
    pp = _mk_parts(s, I["pos"], I["pflag"])
    pt = _pd_int(s, pp, I["ptype"])
    isys, idx = core.ParticleIndexSystem(s), core.IntGrid(s)
    plugins.gridParticleIndex(pp, isys, fl, idx)
    out["gpi_index"] = grid_to_soa(idx)
    out["gpi_sys"] = isys.to_numpy()
    phi = core.LevelsetGrid(s)
    plugins.unionParticleLevelset(pp, isys, fl, idx, phi, 1.0, pt, 4)
    out["union"] = grid_to_soa(phi)
    plugins.extrapolateLsSimple(phi, distance=4, inside=True)
    out["union_els"] = grid_to_soa(phi)
    for nm, kw in (("els_out", dict(distance=3)), ("els_in", dict(distance=4, inside=True)),
                   ("els_walls", dict(distance=2, inside=True, include_walls=True))):
        g = soa_to_grid(core.LevelsetGrid(s), I["phi"])
        plugins.extrapolateLsSimple(g, **kw)
        out[nm] = grid_to_soa(g)
    # setPartType, markIsolatedFluidCell
    pt = _pd_int(s, pp, I["ptype"])
    plugins.setPartType(pp, pt, 4, 1, fl, 4)
    out["ptype"] = pt.data[:pp.np].cpu().numpy().copy()
    fi = soa_to_grid(core.FlagGrid(s), I["fiso"])
    plugins.markIsolatedFluidCell(fi, 4)
    out["isolated"] = grid_to_soa(fi)
    # ptsplugins
    pt = _pd_int(s, pp, I["ptype"])
    pv = _pd_vec3(s, pp, I["pvel"])
    plugins.addForcePvel(pv, core.vec3(0.1, -9.8 * 7.3, 0.02), s.getDt(), pt, 1)
    out["addforce"] = np.ascontiguousarray(pv.to_numpy().T)
    pv = _pd_vec3(s, pp, I["pvel"])
    xp = _pd_vec3(s, pp, I["xprev"])
    plugins.updateVelocityFromDeltaPos(pp, pv, xp, s.getDt(), pt, 4)
    out["updvel"] = np.ascontiguousarray(pv.to_numpy().T)
    pv = _pd_vec3(s, pp, I["pvel"])
    plugins.eulerStep(pp, pv, pt, 1)
    out["euler"] = _ppos(pp)
    # levelset set ops, setBound
    a, b = soa_to_grid(core.LevelsetGrid(s), I["phi"]), soa_to_grid(core.LevelsetGrid(s), I["phiObs"])
    a.join(b)
    out["join"] = grid_to_soa(a)
    a = soa_to_grid(core.LevelsetGrid(s), I["phi"])
    a.subtract(b)
    out["subtract"] = grid_to_soa(a)
    a = soa_to_grid(core.LevelsetGrid(s), I["phi"])
    a.subtract(b, fl, 1)
    out["subtract_flags"] = grid_to_soa(a)
    a = soa_to_grid(core.Grid(s), I["phi"])
    a.setBound(0.75, 2)
    out["setbound"] = grid_to_soa(a)
    s.sync()
    return out


def run_surface_ref(dims, I, dt=0.35):
    sx, sy, sz = dims
    cf = ctypes.c_float
    n = I["pos"].shape[1]
    out = {}
    p = I["pos"].copy()
    refcall("ref_project_out_of_bnd", sx, sy, sz, n, n, p, I["pflag"], cf(1.25), b"xXyYzZ", I["ptype"], 4)
    out["project"] = p
    p = I["pos"].copy()
    refcall("ref_project_out_of_bnd", sx, sy, sz, n, n, p, I["pflag"], cf(2.5), b"xYz", None, 0)
    out["project_xYz"] = p
    p = I["pos"].copy()
    refcall("ref_push_out_of_obs", sx, sy, sz, n, n, p, I["pflag"], I["phiObs"], cf(0.05), cf(0.25), I["ptype"], 4)
    out["push"] = p
    isys, idx, cnt = np.zeros(n, np.int32), np.zeros((sz, sy, sx), np.int32), ctypes.c_int64(0)
    refcall("ref_grid_particle_index", sx, sy, sz, n, n, I["pos"], I["pflag"], isys, idx, ctypes.byref(cnt))
    out["gpi_index"] = idx
    out["gpi_sys"] = isys[:cnt.value].copy()
    phi = np.zeros((sz, sy, sx), np.float32)
    refcall("ref_union_particle_levelset", sx, sy, sz, n, n, I["pos"], I["pflag"], phi, cf(1.0), I["ptype"], 4)
    out["union"] = phi.copy()
    refcall("ref_extrapolate_ls_simple", sx, sy, sz, phi, 4, 1, 0)
    out["union_els"] = phi
    for nm, (dist, ins, walls) in (("els_out", (3, 0, 0)), ("els_in", (4, 1, 0)), ("els_walls", (2, 1, 1))):
        g = I["phi"].copy()
        refcall("ref_extrapolate_ls_simple", sx, sy, sz, g, dist, ins, walls)
        out[nm] = g
    pt = I["ptype"].copy()
    refcall("ref_set_part_type", sx, sy, sz, I["flags"], n, n, I["pos"], pt, 4, 1, 4)
    out["ptype"] = pt
    fi = I["fiso"].copy()
    refcall("ref_mark_isolated_fluid_cell", sx, sy, sz, fi, 4)
    out["isolated"] = fi
    v = I["pvel"].copy()
    refcall("ref_add_force_pvel", n, n, v, cf(0.1), cf(-9.8 * 7.3), cf(0.02), cf(dt), I["ptype"], 1)
    out["addforce"] = v
    v = I["pvel"].copy()
    refcall("ref_update_velocity_from_delta_pos", n, n, I["pos"], v, I["xprev"], cf(dt), I["ptype"], 4)
    out["updvel"] = v
    p = I["pos"].copy()
    refcall("ref_euler_step", n, n, p, I["pvel"], cf(dt), I["ptype"], 1)
    out["euler"] = p
    N = sx * sy * sz
    a = I["phi"].copy()
    refcall("ref_levelset_join", N, a, I["phiObs"])
    out["join"] = a
    a = I["phi"].copy()
    refcall("ref_levelset_subtract", N, a, I["phiObs"], None, 0)
    out["subtract"] = a
    a = I["phi"].copy()
    refcall("ref_levelset_subtract", N, a, I["phiObs"], I["flags"], 1)
    out["subtract_flags"] = a
    a = I["phi"].copy()
    refcall("ref_grid_set_bound", sx, sy, sz, a, cf(0.75), 2)
    out["setbound"] = a
    return out


def dam_geometry(res, zflow=False):
    """grid size and the boxes of the dam-break case: the long axis is x (as in scenes/benchmark_dam.py) or, for the z-slab
    tests, z -- the liquid then crosses the slab faces"""
    bnd = 3
    ext = (int(res * 1.6) + 2 * bnd, res + 2 * bnd, res // 2 + 2 * bnd)
    obs_c, obs_s = (0.45 * ext[0], bnd + 0.1 * res, 0.5 * ext[2]), (0.06 * res, 0.1 * res, 0.2 * res)
    dam_c, dam_s = (ext[0] - bnd - 0.25 * res, bnd + 0.3 * res, 0.5 * ext[2]), (0.25 * res, 0.3 * res, 0.25 * res)
    if zflow:
        # a longer dam (it straddles the slab faces of a 2- and a 3-rank split) and the obstacle further down-stream
        obs_c = (0.25 * ext[0], bnd + 0.1 * res, 0.5 * ext[2])
        dam_c, dam_s = (ext[0] - bnd - 0.45 * res, bnd + 0.3 * res, 0.5 * ext[2]), (0.45 * res, 0.3 * res, 0.25 * res)
        sw = lambda t: (t[2], t[1], t[0])
        ext, obs_c, obs_s, dam_c, dam_s = sw(ext), sw(obs_c), sw(obs_s), sw(dam_c), sw(dam_s)
    return bnd, ext, obs_c, obs_s, dam_c, dam_s


def dam_setup(s, fl, phiS, pp, pT, res, zflow=False):
    """set-up of the dam break on a whole-domain solver (own code; the call sequence of scenes/benchmark_dam.py:66-92)"""
    from mantaflow_amd import core, scene
    bnd, gs, obs_c, obs_s, dam_c, dam_s = dam_geometry(res, zflow)
    fl.initDomain(bnd - 1)
    outer = s.create(scene.Box, p0=core.vec3(0), p1=core.vec3(*gs))
    inner = s.create(scene.Box, p0=core.vec3(bnd), p1=core.vec3(gs[0] - bnd, gs[1] - bnd, gs[2] - bnd))
    phiS.join(outer.computeLevelset())
    phiS.subtract(inner.computeLevelset())
    obs = s.create(scene.Box, center=core.vec3(*obs_c), size=core.vec3(*obs_s))
    obs.applyToGrid(grid=fl, value=2, respectFlags=fl)
    phiS.join(obs.computeLevelset())
    dam = s.create(scene.Box, center=core.vec3(*dam_c), size=core.vec3(*dam_s))
    dam.applyToGrid(grid=fl, value=1, respectFlags=fl)
    scene.sampleShapeWithParticles(shape=dam, flags=fl, parts=pp, discretization=2, randomness=0)
    pT.setConstRange(1, 0, pp.pySize())


def run_dam_pkg(res, steps, deterministic=True, zflow=False, cgacc=1e-3):
    """a ghost-fluid FLIP dam break with the call sequence of scenes/benchmark_dam.py's main loop (own set-up code)"""
    from mantaflow_amd import core, plugins, scene
    FF, FE = 1, 4
    bnd, gs = dam_geometry(res, zflow)[:2]
    s = core.Solver(name="dam", gridSize=core.vec3(*gs), dim=3)
    s.cfl, s.frameLength, s.timestepMin = 1, 1.0 / 30, 0
    s.timestepMax = s.timestep = s.frameLength
    plugins.setDeterministicP2G(deterministic)
    fl, V, Vold, P = s.create(core.FlagGrid), s.create(core.MACGrid), s.create(core.MACGrid), s.create(core.RealGrid)
    phiS, phi = s.create(core.LevelsetGrid), s.create(core.LevelsetGrid)
    isys, idx = s.create(core.ParticleIndexSystem), s.create(core.IntGrid)
    pp = s.create(core.BasicParticleSystem)
    pT, pV, pX = pp.create(core.PdataInt), pp.create(core.PdataVec3), pp.create(core.PdataVec3)
    dam_setup(s, fl, phiS, pp, pT, res, zflow)
    grav = core.vec3(0, -9.8 * res, 0)
    iters, rec = [], {}
    for step in range(steps):
        plugins.mapPartsToMAC(vel=V, flags=fl, velOld=Vold, parts=pp, partVel=pV, ptype=pT, exclude=FE)
        s.adaptTimestep(V.getMaxAbs())
        plugins.addGravityNoScale(flags=fl, vel=V, gravity=grav)
        plugins.gridParticleIndex(parts=pp, flags=fl, indexSys=isys, index=idx)
        plugins.unionParticleLevelset(parts=pp, indexSys=isys, flags=fl, index=idx, phi=phi, radiusFactor=1.0)
        plugins.extrapolateLsSimple(phi=phi, distance=4, inside=True)
        if step == steps - 1:      # the last step's level set, before the solve feeds back
            rec["phi_ls"], rec["dt_ls"] = grid_to_soa(phi), float(s.timestep)
        plugins.setWallBcs(flags=fl, vel=V)
        plugins.solvePressure(flags=fl, vel=V, pressure=P, cgAccuracy=cgacc, phi=phi)
        iters.append(plugins.lastCgStats()["iterations"])
        plugins.setWallBcs(flags=fl, vel=V)
        plugins.extrapolateMACSimple(flags=fl, vel=V)
        plugins.flipVelocityUpdate(vel=V, velOld=Vold, flags=fl, parts=pp, partVel=pV, flipRatio=0.97, ptype=pT, exclude=FE)
        plugins.addForcePvel(vel=pV, a=grav, dt=s.timestep, ptype=pT, exclude=FF)
        pp.getPosPdata(target=pX)
        pp.advectInGrid(flags=fl, vel=V, integrationMode=2, deleteInObstacle=False, ptype=pT, exclude=FE)
        plugins.eulerStep(parts=pp, vel=pV, ptype=pT, exclude=FF)
        pp.projectOutOfBnd(flags=fl, bnd=bnd + 0.25, plane="xXyYzZ", ptype=pT)
        plugins.pushOutofObs(parts=pp, flags=fl, phiObs=phiS, thresh=0.25, ptype=pT)
        plugins.updateVelocityFromDeltaPos(parts=pp, vel=pV, x_prev=pX, dt=s.timestep, ptype=pT, exclude=FF)
        plugins.markFluidCells(parts=pp, flags=fl, ptype=pT)
        plugins.setPartType(parts=pp, ptype=pT, mark=FF, stype=FE, flags=fl, cflag=FF)
        plugins.markIsolatedFluidCell(flags=fl, mark=FE)
        plugins.setPartType(parts=pp, ptype=pT, mark=FE, stype=FF, flags=fl, cflag=FE)
        s.step()
    s.sync()
    plugins.setDeterministicP2G(True)
    return dict(pos=_ppos(pp), pvel=np.ascontiguousarray(pV.to_numpy().T), ptype=pT.data[:pp.np].cpu().numpy().copy(),
                flags=grid_to_soa(fl), phi=grid_to_soa(phi), vel=grid_to_soa(V), pres=grid_to_soa(P), iters=iters, gs=gs,
                dt=float(s.timestep), rec=rec)


# ---- resampling between grids of different size ----
INTERP_CASES = [  # (source dims, target dims, scale, offset, size)
    ((10, 8, 6), (20, 16, 12), (1., 1., 1.), (0., 0., 0.), None),
    ((16, 12, 10), (9, 7, 5), (1., 1., 1.), (0., 0., 0.), None),
    ((12, 9, 7), (17, 13, 10), (1.5, 0.75, 1.25), (0.5, -1.0, 0.25), (20, -1, 12)),
    ((14, 10, 1), (21, 15, 1), (1., 1., 1.), (0., 0., 0.), None),
]


def run_interp_pkg(sd, td, scale, offset, size, fields, orderSpace=1):
    from mantaflow_amd import core, plugins
    ss, ts = _mk_solver(sd), _mk_solver(td)
    kw = dict(scale=core.vec3(*scale), offset=core.vec3(*offset), orderSpace=orderSpace)
    if size is not None:
        kw["size"] = size
    out = {}
    g = core.Grid(ts)
    plugins.interpolateGrid(g, soa_to_grid(core.Grid(ss), fields["real"]), **kw)
    out["real"] = grid_to_soa(g)
    g = core.VecGrid(ts)
    plugins.interpolateGridVec3(g, soa_to_grid(core.VecGrid(ss), fields["vec"]), **kw)
    out["vec"] = grid_to_soa(g)
    g = core.MACGrid(ts)
    plugins.interpolateMACGrid(g, soa_to_grid(core.MACGrid(ss), fields["vec"]), **kw)
    out["mac"] = grid_to_soa(g)
    ts.sync()
    return out


def run_interp_ref(sd, td, scale, offset, size, fields, orderSpace=1):
    cf = ctypes.c_float
    zs = size if size is not None else (-1, -1, -1)
    out = {}
    for kind, key, src in ((0, "real", fields["real"]), (1, "vec", fields["vec"]), (2, "mac", fields["vec"])):
        shape = (td[2], td[1], td[0]) if kind == 0 else (3, td[2], td[1], td[0])
        t = np.zeros(shape, np.float32)
        refcall("ref_interpolate_grid", kind, td[0], td[1], td[2], t, sd[0], sd[1], sd[2], src, cf(scale[0]), cf(scale[1]), cf(scale[2]),
                cf(offset[0]), cf(offset[1]), cf(offset[2]), zs[0], zs[1], zs[2], int(orderSpace))
        out[key] = t
    return out


def run_simpleplume_pkg(res, steps, inflow_steps=100):
    """the main loop of scenes/simpleplume.py (BASELINE config 0) through the package"""
    from mantaflow_amd import core, plugins, scene
    v3 = core.vec3
    gs = v3(res, int(1.5 * res), res)
    s = core.FluidSolver(name="main", gridSize=gs)
    flags, vel, density, pressure = s.create(core.FlagGrid), s.create(core.MACGrid), s.create(core.RealGrid), s.create(core.RealGrid)
    noise = s.create(scene.NoiseField, loadFromFile=True)
    noise.posScale = v3(45)
    noise.clamp, noise.clampNeg, noise.clampPos = True, 0, 1
    noise.valOffset, noise.timeAnim = 0.75, 0.2
    source = s.create(scene.Cylinder, center=gs * v3(0.5, 0.1, 0.5), radius=res * 0.14, z=gs * v3(0, 0.02, 0))
    flags.initDomain()
    flags.fillGrid()
    iters = []
    for t in range(steps):
        if t < inflow_steps:
            scene.densityInflow(flags=flags, density=density, noise=noise, shape=source, scale=1, sigma=0.5)
        plugins.advectSemiLagrange(flags=flags, vel=vel, grid=density, order=2)
        plugins.advectSemiLagrange(flags=flags, vel=vel, grid=vel, order=2, strength=1.0)
        plugins.setWallBcs(flags=flags, vel=vel)
        plugins.addBuoyancy(density=density, vel=vel, gravity=v3(0, -6e-4, 0), flags=flags)
        plugins.solvePressure(flags=flags, vel=vel, pressure=pressure)
        iters.append(plugins.lastCgStats()["iterations"])
        s.step()
    s.sync()
    return {"density": grid_to_soa(density), "vel": grid_to_soa(vel), "iters": iters}


def run_simpleplume_ref(res, steps, inflow_steps=100):
    sx, sy, sz = res, int(1.5 * res), res
    rd, rv = np.zeros((sz, sy, sx), np.float32), np.zeros((3, sz, sy, sx), np.float32)
    refcall("ref_simpleplume", res, steps, inflow_steps, rd, rv)
    return {"density": rd, "vel": rv}


def run_wavelet_scene_pkg(res, dim, steps, upres=2):
    """the main loop of scenes/waveletTurbulence.py through the package (same calls, arguments and order)"""
    from mantaflow_amd import api as m
    wltStrength = 0.4
    gs = m.vec3(res, int(1.5 * res), res)
    if dim == 2:
        gs.z = 1
    sm = m.Solver(name="main", gridSize=gs, dim=dim)
    sm.timestep = 1.5
    velInflow = m.vec3(0.025, 0, 0)
    noise = m.NoiseField(parent=sm, fixedSeed=265, loadFromFile=True)
    noise.posScale = m.vec3(20)
    noise.clamp, noise.clampNeg, noise.clampPos = True, 0, 2
    noise.valScale, noise.valOffset, noise.timeAnim = 1, 0.075, 0.3
    source = m.Cylinder(parent=sm, center=gs * m.vec3(0.3, 0.2, 0.5), radius=res * 0.081, z=gs * m.vec3(0.081, 0, 0))
    sourceVel = m.Cylinder(parent=sm, center=gs * m.vec3(0.3, 0.2, 0.5), radius=res * 0.15, z=gs * m.vec3(0.15, 0, 0))
    xl_gs = m.vec3(upres * gs.x, upres * gs.y, upres * gs.z)
    if dim == 2:
        xl_gs.z = 1
    xl = m.Solver(name="larger", gridSize=xl_gs, dim=dim)
    xl.timestep = sm.timestep
    xl_flags, xl_vel, xl_density, xl_weight = xl.create(m.FlagGrid), xl.create(m.MACGrid), xl.create(m.RealGrid), xl.create(m.RealGrid)
    xl_flags.initDomain()
    xl_flags.fillGrid()
    xl_source = m.Cylinder(parent=xl, center=xl_gs * m.vec3(0.3, 0.2, 0.5), radius=xl_gs.x * 0.081, z=xl_gs * m.vec3(0.081, 0, 0))
    xl_noise = m.NoiseField(parent=xl, fixedSeed=265, loadFromFile=True)
    xl_noise.posScale, xl_noise.clamp, xl_noise.clampNeg, xl_noise.clampPos = noise.posScale, noise.clamp, noise.clampNeg, noise.clampPos
    xl_noise.valScale, xl_noise.valOffset, xl_noise.timeAnim = noise.valScale, noise.valOffset, noise.timeAnim * upres
    wlt = []
    for lvl in range(3):
        n = m.NoiseField(parent=xl, loadFromFile=True)
        n.posScale = m.vec3(int(1.0 * gs.x)) * 0.5 if lvl == 0 else wlt[-1].posScale * 2.0
        n.timeAnim = 0.1
        wlt.append(n)
    flags, vel, density, pressure, energy = (sm.create(t) for t in (m.FlagGrid, m.MACGrid, m.RealGrid, m.RealGrid, m.RealGrid))
    flags.initDomain(boundaryWidth=0)
    flags.fillGrid()
    m.setOpenBound(flags, 0, "Y", m.FlagOutflow | m.FlagEmpty)
    for t in range(steps):
        m.advectSemiLagrange(flags=flags, vel=vel, grid=density, order=2)
        m.advectSemiLagrange(flags=flags, vel=vel, grid=vel, order=2)
        applyInflow = False
        if sm.timeTotal >= 0 and sm.timeTotal < 50.:
            m.densityInflow(flags=flags, density=density, noise=noise, shape=source, scale=1, sigma=0.5)
            sourceVel.applyToGrid(grid=vel, value=(velInflow * float(res)))
            applyInflow = True
        m.setWallBcs(flags=flags, vel=vel)
        m.addBuoyancy(density=density, vel=vel, gravity=m.vec3(0, -1e-3, 0), flags=flags)
        m.vorticityConfinement(vel=vel, flags=flags, strength=0.3)
        m.solvePressure(flags=flags, vel=vel, pressure=pressure, cgMaxIterFac=1.0, cgAccuracy=0.01)
        m.setWallBcs(flags=flags, vel=vel)
        m.computeEnergy(flags=flags, vel=vel, energy=energy)
        m.computeWaveletCoeffs(energy)
        sm.step()
        m.interpolateGrid(target=xl_weight, source=energy)
        m.interpolateMACGrid(source=vel, target=xl_vel)
        m.applyNoiseVec3(flags=xl_flags, target=xl_vel, noise=wlt[0], scale=wltStrength * 1.0, weight=xl_weight)
        m.applyNoiseVec3(flags=xl_flags, target=xl_vel, noise=wlt[1], scale=wltStrength * 0.6, weight=xl_weight)
        m.applyNoiseVec3(flags=xl_flags, target=xl_vel, noise=wlt[2], scale=wltStrength * 0.6 * 0.6, weight=xl_weight)
        for substep in range(upres):
            m.advectSemiLagrange(flags=xl_flags, vel=xl_vel, grid=xl_density, order=2)
        if applyInflow:
            m.densityInflow(flags=xl_flags, density=xl_density, noise=xl_noise, shape=xl_source, scale=1, sigma=0.5)
        xl.step()
    sm.sync()
    xl.sync()
    return {"density": grid_to_soa(density), "vel": grid_to_soa(vel), "xl_density": grid_to_soa(xl_density), "xl_vel": grid_to_soa(xl_vel)}


def run_wavelet_scene_ref(res, dim, steps, upres=2):
    """scenes/waveletTurbulence.py through the compiled reference (oracle/ref_shim.cpp:ref_waveletturbulence)"""
    gs = (res, int(1.5 * res), res if dim == 3 else 1)
    xg = (upres * gs[0], upres * gs[1], upres * gs[2] if dim == 3 else 1)
    d = np.zeros(gs[::-1], np.float32)
    v = np.zeros((3,) + gs[::-1], np.float32)
    xd = np.zeros(xg[::-1], np.float32)
    xv = np.zeros((3,) + xg[::-1], np.float32)
    refcall("ref_waveletturbulence", res, dim, steps, upres, d, v, xd, xv)
    return {"density": d, "vel": v, "xl_density": xd, "xl_vel": xv}


# ---- wavelet turbulence pieces (scenes/waveletTurbulence.py) ----
def run_turb_pkg(dims, flags, vel, energy_in, weight_small, small_dims, t_total=2.5):
    from mantaflow_amd import core, plugins, scene
    s = _mk_solver(dims)
    s.timeTotal = t_total
    out = {}
    fl = soa_to_grid(core.FlagGrid(s), flags)
    v = soa_to_grid(core.MACGrid(s), vel)
    e = core.Grid(s)
    plugins.computeEnergy(flags=fl, vel=v, energy=e)
    out["energy"] = grid_to_soa(e)
    w = soa_to_grid(core.Grid(s), energy_in)
    plugins.computeWaveletCoeffs(w)
    out["coeffs"] = grid_to_soa(w)
    v = soa_to_grid(core.MACGrid(s), vel)
    plugins.vorticityConfinement(vel=v, flags=fl, strength=0.3)
    out["vortconf"] = grid_to_soa(v)
    v = soa_to_grid(core.MACGrid(s), vel)
    plugins.vorticityConfinement(vel=v, flags=fl, strength=0.1, strengthCell=soa_to_grid(core.Grid(s), energy_in))
    out["vortconf_cell"] = grid_to_soa(v)
    for nm, bw, ob in (("open_Y", 0, "Y"), ("open_xYz", 1, "xYz")):
        f2 = soa_to_grid(core.FlagGrid(s), np.where(flags & util.OBS, flags, 1).astype(np.int32))
        f2.initDomain(boundaryWidth=bw)
        f2.fillGrid()
        plugins.setOpenBound(f2, bw, ob, 16 | 4)
        out[nm] = grid_to_soa(f2)
    noise = scene.NoiseField(parent=s, loadFromFile=True)
    noise.posScale = core.vec3(int(1.0 * dims[0])) * 0.5
    noise.timeAnim = 0.1
    t = soa_to_grid(core.VecGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise, scale=0.4, weight=w)
    out["noise_same"] = grid_to_soa(t)
    ss = _mk_solver(small_dims)
    t = soa_to_grid(core.MACGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise, scale=0.24, scaleSpatial=1.5, weight=soa_to_grid(core.Grid(ss), weight_small))
    out["noise_interp"] = grid_to_soa(t)
    t = soa_to_grid(core.VecGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise)
    out["noise_plain"] = grid_to_soa(t)
    # uv variant (waveletturbulence.cpp:139-147): positions from a uv grid of the same size (+ weight), and of another size (interpolated)
    uv_same, uv_small = turb_uv(dims, small_dims)
    t = soa_to_grid(core.VecGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise, scale=0.4, weight=w, uv=soa_to_grid(core.VecGrid(s), uv_same))
    out["noise_uv_same"] = grid_to_soa(t)
    t = soa_to_grid(core.VecGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise, scale=0.3, scaleSpatial=1.25, weight=soa_to_grid(core.Grid(ss), weight_small),
                           uv=soa_to_grid(core.VecGrid(ss), uv_small))
    out["noise_uv_interp"] = grid_to_soa(t)
    t = soa_to_grid(core.VecGrid(s), vel)
    plugins.applyNoiseVec3(flags=fl, target=t, noise=noise, uv=soa_to_grid(core.VecGrid(ss), uv_small))
    out["noise_uv_only"] = grid_to_soa(t)
    s.sync()
    return out


def turb_uv(dims, small_dims):
    """uv grids for applyNoiseVec3: advected-texture-like coordinates (cell centres + a smooth displacement) at both sizes"""
    out = []
    for dd, seed in ((dims, 81), (small_dims, 82)):
        sx, sy, sz = dd
        zz, yy, xx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
        disp = util.smooth_vel(sx, sy, sz, seed, 1.5)
        uv = np.stack([xx + 0.5, yy + 0.5, zz + 0.5]).astype(np.float32) + disp
        out.append(np.ascontiguousarray(uv.astype(np.float32)))
    return out


def run_turb_ref(dims, flags, vel, energy_in, weight_small, small_dims, t_total=2.5):
    sx, sy, sz = dims
    cf = ctypes.c_float
    out = {}
    e = np.zeros((sz, sy, sx), np.float32)
    refcall("ref_compute_energy", sx, sy, sz, flags, vel, e)
    out["energy"] = e
    w = energy_in.copy()
    refcall("ref_compute_wavelet_coeffs", sx, sy, sz, w)
    out["coeffs"] = w
    v = vel.copy()
    refcall("ref_vorticity_confinement", sx, sy, sz, v, flags, cf(0.3), None)
    out["vortconf"] = v
    v = vel.copy()
    refcall("ref_vorticity_confinement", sx, sy, sz, v, flags, cf(0.1), energy_in)
    out["vortconf_cell"] = v
    pad = lambda x: (x + "      ").encode()
    for nm, bw, ob in (("open_Y", 0, "Y"), ("open_xYz", 1, "xYz")):
        f2 = np.zeros((sz, sy, sx), np.int32)
        refcall("ref_init_domain", sx, sy, sz, f2, bw, b"xXyYzZ", pad(""), pad(""), pad(""), 1)
        refcall("ref_set_open_bound", sx, sy, sz, f2, bw, ob.encode(), 16 | 4)
        out[nm] = f2
    ps = np.float32(int(1.0 * dims[0])) * np.float32(0.5)
    P = np.array([ps, ps, ps, 0, 0, 0, 0.0, 1.0, 0, 0, 1, 0.1], np.float32)
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(0.4), cf(1.0), w, sx, sy, sz, None)
    out["noise_same"] = t
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(0.24), cf(1.5), weight_small, small_dims[0], small_dims[1], small_dims[2], None)
    out["noise_interp"] = t
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(1.0), cf(1.0), None, 0, 0, 0, None)
    out["noise_plain"] = t
    uv_same, uv_small = turb_uv(dims, small_dims)
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(0.4), cf(1.0), w, sx, sy, sz, uv_same)
    out["noise_uv_same"] = t
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(0.3), cf(1.25), weight_small, small_dims[0], small_dims[1], small_dims[2], uv_small)
    out["noise_uv_interp"] = t
    t = vel.copy()
    refcall("ref_apply_noise_vec3", sx, sy, sz, cf(t_total), flags, t, -1, P, cf(1.0), cf(1.0), None, small_dims[0], small_dims[1], small_dims[2], uv_small)
    out["noise_uv_only"] = t
    return out


def turb_inputs(dims, small_dims, seed):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, seed, empty_top=True)
    vel = util.smooth_vel(sx, sy, sz, seed + 1, 1.5)
    if sz == 1:
        vel[2] = 0
    energy_in = np.abs(util.rand_real((sz, sy, sx), seed + 2)).astype(np.float32)
    weight_small = np.abs(util.rand_real((small_dims[2], small_dims[1], small_dims[0]), seed + 3)).astype(np.float32)
    return flags, vel, energy_in, weight_small


def run_glue_pkg(dims, dt, flags, vel, density, obvel=None):
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims, dt)
    fl = soa_to_grid(core.FlagGrid(s), flags)
    out = {}
    v = soa_to_grid(core.MACGrid(s), vel)
    plugins.setWallBcs(fl, v, obvel=None if obvel is None else soa_to_grid(core.MACGrid(s), obvel))
    out["wall"] = grid_to_soa(v)
    v = soa_to_grid(core.MACGrid(s), vel)
    plugins.addBuoyancy(fl, soa_to_grid(core.Grid(s), density), v, core.vec3(0.1, -4e-3, 0.02))
    out["buoy"] = grid_to_soa(v)
    v = soa_to_grid(core.MACGrid(s), vel)
    plugins.addGravity(fl, v, core.vec3(0.0, -0.003, 0.001))
    out["grav"] = grid_to_soa(v)
    s.sync()
    return out


def run_glue_ref(dims, dt, flags, vel, density, obvel=None):
    sx, sy, sz = dims
    out = {}
    v = vel.copy()
    refcall("ref_set_wall_bcs", sx, sy, sz, flags, v, obvel)
    out["wall"] = v
    v = vel.copy()
    cf = ctypes.c_float
    refcall("ref_add_buoyancy", sx, sy, sz, cf(dt), flags, density, v, cf(0.1), cf(-4e-3), cf(0.02), cf(1.0), 1)
    out["buoy"] = v
    v = vel.copy()
    refcall("ref_add_gravity", sx, sy, sz, cf(dt), flags, v, cf(0.0), cf(-0.003), cf(0.001), None, 1)
    out["grav"] = v
    return out


# ---------------------------------------------------------------------------------------------------------
# the golden set: same scenarios as tests/golden/make_golden.py, computed by an implementation under test
# ---------------------------------------------------------------------------------------------------------
def golden_outputs(impl, deterministic_p2g=True):
    """impl: util.Impl for the raw-ABI cases; the package host layer must already be routed to the same library."""
    out = {}
    for tag, dims, seed in [("a", (16, 16, 16), 5), ("b", (20, 13, 11), 7)]:
        flags, A, src = system_inputs(dims, seed)
        out["apply_%s" % tag] = run_apply_matrix_impl(impl, dims, flags, A, src)
        ap, dst = run_mic_impl(impl, dims, flags, A, src)
        out["micinit_%s" % tag], out["micapply_%s" % tag] = ap, dst
        rhs = cg_rhs(dims, flags, seed)
        x, st = run_cg_impl(impl, dims, flags, A, rhs, 2, 1e-3, 60)
        out["cg_%s" % tag], out["cgstat_%s" % tag] = x, np.array(st, np.float64)
    for tag, dims, liquid in [("smoke", (16, 16, 16), False), ("liquid", (20, 13, 11), True), ("2d", (24, 18, 1), True)]:
        flags, vel, phi = pressure_inputs(dims, 6, liquid)
        r = run_solve_pressure_pkg(dims, flags, vel, phi)
        for k in ("rhs", "pressure", "vel"):
            out["sp_%s_%s" % (tag, k)] = r[k]
    dims = (14, 12, 10)
    sx, sy, sz = dims
    for kind in (0, 1, 2):
        flags, vel = advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
        field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
        for order, cm in ((1, 2), (2, 1), (2, 2)):
            out["adv_k%d_o%d_c%d" % (kind, order, cm)] = run_advect_pkg(dims, 0.9, flags, vel, field, kind, order=order, clampMode=cm,
                                                                        strength=0.8 if order == 2 else 1.0)
    dims = (12, 10, 9)
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 14, empty_top=True)
    vel, velOld = util.rand_vel(sx, sy, sz, 15), util.rand_vel(sx, sy, sz, 16)
    pos, pflag, pvel = util.make_particles(flags, 3, 17)
    r = run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, deterministic=deterministic_p2g)
    for k, v in r.items():
        out["flip_" + k] = v
    flags = util.make_flags(sx, sy, sz, 19, empty_top=True)
    v2 = util.smooth_vel(sx, sy, sz, 20, 2.0)
    pos, pflag, _ = util.make_particles(flags, 2, 21)
    for mode in (0, 1, 2):
        p, f = run_advect_parts_pkg(dims, 0.8, flags, v2, pos, pflag, mode, False, True)
        out["padv_m%d_pos" % mode], out["padv_m%d_flag" % mode] = p, f
    p, f = run_advect_parts_pkg(dims, 0.8, flags, v2, pos, pflag, 2, True, True)
    out["padv_del_pos"], out["padv_del_flag"] = p, f
    gd = (14, 12, 10)
    fl0, pos, pflag, pvel, vel = flipglue_inputs(gd, 31)
    for k, v in run_flipglue_pkg(gd, fl0, pos, pflag, pvel, vel, None).items():
        out["glue_" + k] = v
    for k, v in run_surface_pkg(gd, surface_inputs(gd, 51)).items():
        out["surf_" + k] = v
    r = run_simpleplume_pkg(16, 5)
    out["plume_density"], out["plume_vel"] = r["density"], r["vel"]
    td, ts = (20, 14, 12), (10, 7, 6)
    for k, v in run_turb_pkg(td, *turb_inputs(td, ts, 91), ts).items():
        out["turb_" + k] = v
    for tag, dim, res, steps in (("wlt2d_", 2, 32, 6), ("wlt3d_", 3, 16, 4)):
        for k, v in run_wavelet_scene_pkg(res, dim, steps).items():
            out[tag + k] = v
    ad = (12, 10, 9)
    out.update(run_apic_pkg(ad, *apic_inputs(ad, 61)))
    return out


def load_golden():
    import os
    return dict(np.load(os.path.join(util.GOLDEN, "reference_vectors.npz")))


# ---- scenes/waveletTurbulence.py, 3-D, shared by the plugin path and the z-slab path (tests/slab_worker.py) -----------------
WLT_UPRES, WLT_STRENGTH, WLT_DT = 2, 0.4, 1.5


def wavelet_inputs(gs):
    """global set-up arrays of the wavelet-turbulence case: coarse / fine flags with the open +y boundary of the scene
    (initDomain + fillGrid + setOpenBound on plain whole-domain solvers), a smooth synthetic velocity and a random fine density for
    the bit-exact check of the up-res pipeline"""
    from mantaflow_amd import core, plugins
    out = {}
    for name, f in (("flags", 1), ("xl_flags", WLT_UPRES)):
        s = _mk_solver(tuple(v * f for v in gs), WLT_DT)
        fl = core.FlagGrid(s)
        fl.initDomain(0)
        fl.fillGrid()
        if f == 1:
            plugins.setOpenBound(fl, 0, "Y", 16 | 4)
        out[name] = grid_to_soa(fl)
    out["vel_syn"] = util.smooth_vel(*gs, 91, 0.6)
    out["xl_dens_syn"] = util.rand_real((gs[2] * WLT_UPRES, gs[1] * WLT_UPRES, gs[0] * WLT_UPRES), 92)
    return out


def wavelet_objects(sm, xl, gs):
    """noise fields and shapes of the scene (waveletTurbulence.py:27-80), parents = the two solvers (plain or slab)"""
    from mantaflow_amd import core, scene
    vec3 = core.vec3
    res = gs[0]
    g, xg = vec3(*gs), vec3(*[v * WLT_UPRES for v in gs])
    o = {}
    n = o["noise"] = scene.NoiseField(parent=sm, fixedSeed=265, loadFromFile=True)
    n.posScale, n.clamp, n.clampNeg, n.clampPos, n.valScale, n.valOffset, n.timeAnim = vec3(20), True, 0, 2, 1, 0.075, 0.3
    o["source"] = scene.Cylinder(parent=sm, center=g * vec3(0.3, 0.2, 0.5), radius=res * 0.081, z=g * vec3(0.081, 0, 0))
    o["sourceVel"] = scene.Cylinder(parent=sm, center=g * vec3(0.3, 0.2, 0.5), radius=res * 0.15, z=g * vec3(0.15, 0, 0))
    o["xl_source"] = scene.Cylinder(parent=xl, center=xg * vec3(0.3, 0.2, 0.5), radius=xg.x * 0.081, z=xg * vec3(0.081, 0, 0))
    xn = o["xl_noise"] = scene.NoiseField(parent=xl, fixedSeed=265, loadFromFile=True)
    xn.posScale, xn.clamp, xn.clampNeg, xn.clampPos, xn.valScale, xn.valOffset = n.posScale, n.clamp, n.clampNeg, n.clampPos, n.valScale, n.valOffset
    xn.timeAnim = n.timeAnim * WLT_UPRES
    w1 = o["wlt1"] = scene.NoiseField(parent=xl, loadFromFile=True)
    w1.posScale, w1.timeAnim = vec3(int(1.0 * gs[0])) * 0.5, 0.1
    w2 = o["wlt2"] = scene.NoiseField(parent=xl, loadFromFile=True)
    w2.posScale, w2.timeAnim = w1.posScale * 2.0, 0.1
    w3 = o["wlt3"] = scene.NoiseField(parent=xl, loadFromFile=True)
    w3.posScale, w3.timeAnim = w2.posScale * 2.0, 0.1
    o["velInflow"] = vec3(0.025, 0, 0) * float(res)
    return o


def run_wavelet_pkg(gs, steps, cgacc=1e-6):
    """the loop of scenes/waveletTurbulence.py:105-146 (3-D) through the plugins on two plain solvers, preceded by one pass of the
    up-res pipeline on synthetic input (recorded: it involves no pressure solve, so a z-slab run must reproduce it bit for bit)"""
    from mantaflow_amd import core, plugins, scene
    inp = wavelet_inputs(gs)
    sm, xl = _mk_solver(gs, WLT_DT), _mk_solver(tuple(v * WLT_UPRES for v in gs), WLT_DT)
    o = wavelet_objects(sm, xl, gs)
    fl, V, D, P, E = core.FlagGrid(sm), core.MACGrid(sm), core.Grid(sm), core.Grid(sm), core.Grid(sm)
    xfl, xV, xD, xW = core.FlagGrid(xl), core.MACGrid(xl), core.Grid(xl), core.Grid(xl)
    soa_to_grid(fl, inp["flags"]); soa_to_grid(xfl, inp["xl_flags"])

    def upres_pass():
        plugins.interpolateGrid(target=xW, source=E)
        plugins.interpolateMACGrid(source=V, target=xV)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt1"], scale=WLT_STRENGTH * 1.0, weight=xW)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt2"], scale=WLT_STRENGTH * 0.6, weight=xW)
        plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt3"], scale=WLT_STRENGTH * 0.6 * 0.6, weight=xW)
        for _ in range(WLT_UPRES):
            plugins.advectSemiLagrange(flags=xfl, vel=xV, grid=xD, order=2)

    rec = {}
    soa_to_grid(V, inp["vel_syn"]); soa_to_grid(xD, inp["xl_dens_syn"])
    plugins.setWallBcs(flags=fl, vel=V)
    plugins.computeEnergy(flags=fl, vel=V, energy=E)
    plugins.computeWaveletCoeffs(E)
    upres_pass()
    rec["energy0"], rec["xl_vel0"], rec["xl_dens0"] = grid_to_soa(E), grid_to_soa(xV), grid_to_soa(xD)
    V.clear(); xD.clear(); xV.clear(); E.clear()
    iters = []
    for t in range(steps):
        plugins.advectSemiLagrange(flags=fl, vel=V, grid=D, order=2)
        plugins.advectSemiLagrange(flags=fl, vel=V, grid=V, order=2)
        scene.densityInflow(flags=fl, density=D, noise=o["noise"], shape=o["source"], scale=1, sigma=0.5)
        o["sourceVel"].applyToGrid(grid=V, value=o["velInflow"])
        plugins.setWallBcs(flags=fl, vel=V)
        plugins.addBuoyancy(density=D, vel=V, gravity=core.vec3(0, -1e-3, 0), flags=fl)
        plugins.vorticityConfinement(vel=V, flags=fl, strength=0.3)
        if t == 0:
            rec["vel_pre0"], rec["dens_pre0"] = grid_to_soa(V), grid_to_soa(D)
        plugins.solvePressure(flags=fl, vel=V, pressure=P, cgMaxIterFac=2.0, cgAccuracy=cgacc)
        iters.append(plugins.lastCgStats()["iterations"])
        plugins.setWallBcs(flags=fl, vel=V)
        plugins.computeEnergy(flags=fl, vel=V, energy=E)
        plugins.computeWaveletCoeffs(E)
        sm.step()
        upres_pass()
        scene.densityInflow(flags=xfl, density=xD, noise=o["xl_noise"], shape=o["xl_source"], scale=1, sigma=0.5)
        xl.step()
    sm.sync()
    return dict(dens=grid_to_soa(D), vel=grid_to_soa(V), energy=grid_to_soa(E), xl_dens=grid_to_soa(xD), xl_vel=grid_to_soa(xV), iters=iters, **rec)


# ---- whole steps at BASELINE's stated sizes (tests/test_gpu_fullsize.py) -----------------------------------------------------
def run_smoke_step_pkg(dims, dt, flags, vel, dens):
    """bench.py's step: advect density + velocity (MacCormack), setWallBcs, solvePressure (MIC-CG 1e-3)"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims, dt)
    fl, v, d, p = core.FlagGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
    soa_to_grid(fl, flags); soa_to_grid(v, vel); soa_to_grid(d, dens)
    plugins.setWallBcs(fl, v)
    plugins.advectSemiLagrange(fl, v, d, order=2)
    plugins.advectSemiLagrange(fl, v, v, order=2)
    plugins.setWallBcs(fl, v)
    out = dict(dens=grid_to_soa(d), vel_adv=grid_to_soa(v).copy())
    plugins.solvePressure(v, p, fl)
    s.sync()
    out.update(pressure=grid_to_soa(p), vel=grid_to_soa(v), iters=plugins.lastCgStats()["iterations"])
    return out


def run_flip_step_pkg(dims, dt, flags, vel, pos, pflag, pvel):
    """S-flip: advectInGrid(RK4), mapPartsToMAC, setWallBcs, solvePressure, flipVelocityUpdate"""
    from mantaflow_amd import core, plugins
    s = _mk_solver(dims, dt)
    fl, v, vo, w, p = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.VecGrid(s), core.Grid(s)
    soa_to_grid(fl, flags); soa_to_grid(v, vel)
    pp = _mk_parts(s, pos, pflag)
    pv = _pd_vec3(s, pp, pvel)
    pp.advectInGrid(fl, v, 2, deleteInObstacle=False)
    P, F = _parts_get(pp)
    plugins.mapPartsToMAC(fl, v, vo, pp, pv, w)
    out = dict(pos=P, pflag=F, p2g_vel=grid_to_soa(v).copy(), p2g_w=grid_to_soa(w).copy())
    plugins.setWallBcs(fl, v)
    plugins.solvePressure(v, p, fl)
    plugins.flipVelocityUpdate(fl, v, vo, pp, pv, 0.97)
    s.sync()
    out.update(vel=grid_to_soa(v), pvel=_pd_get(pv, pp.np), iters=plugins.lastCgStats()["iterations"])
    return out


def run_upres_pass_pkg(nc, vel_coarse):
    """the fine-grid part of scenes/waveletTurbulence.py:128-140 on a coarse nc^3 / fine (2 nc)^3 pair with a synthetic coarse velocity:
    computeEnergy + computeWaveletCoeffs (coarse), interpolateGrid, interpolateMACGrid, three octaves of applyNoiseVec3, two MacCormack
    advections of the fine density -- no pressure solve, so every field has to come out bit for bit"""
    from mantaflow_amd import core, plugins
    gs = (nc, nc, nc)
    sm, xl = _mk_solver(gs, WLT_DT), _mk_solver(tuple(v * WLT_UPRES for v in gs), WLT_DT)
    o = wavelet_objects(sm, xl, gs)
    fl, V, E = core.FlagGrid(sm), core.MACGrid(sm), core.Grid(sm)
    xfl, xV, xD, xW = core.FlagGrid(xl), core.MACGrid(xl), core.Grid(xl), core.Grid(xl)
    fl.initDomain(0); fl.fillGrid(); plugins.setOpenBound(fl, 0, "Y", 16 | 4)
    xfl.initDomain(0); xfl.fillGrid()
    soa_to_grid(V, vel_coarse)
    n2 = nc * WLT_UPRES
    z = np.arange(n2, dtype=np.float32)[:, None, None]; y = np.arange(n2, dtype=np.float32)[None, :, None]; x = np.arange(n2, dtype=np.float32)[None, None, :]
    xD.from_numpy((0.5 + 0.5 * np.sin(0.05 * x + 0.3) * np.cos(0.04 * y) * np.sin(0.03 * z + 1.0)).astype(np.float32))
    plugins.setWallBcs(flags=fl, vel=V)
    plugins.computeEnergy(flags=fl, vel=V, energy=E)
    plugins.computeWaveletCoeffs(E)
    plugins.interpolateGrid(target=xW, source=E)
    plugins.interpolateMACGrid(source=V, target=xV)
    plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt1"], scale=WLT_STRENGTH * 1.0, weight=xW)
    plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt2"], scale=WLT_STRENGTH * 0.6, weight=xW)
    plugins.applyNoiseVec3(flags=xfl, target=xV, noise=o["wlt3"], scale=WLT_STRENGTH * 0.6 * 0.6, weight=xW)
    for _ in range(WLT_UPRES):
        plugins.advectSemiLagrange(flags=xfl, vel=xV, grid=xD, order=2)
    xl.sync()
    return dict(energy=grid_to_soa(E), xl_weight=grid_to_soa(xW), xl_vel=grid_to_soa(xV), xl_dens=grid_to_soa(xD))
