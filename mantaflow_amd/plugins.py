"""Python-callable operators ("plugins") of the hot path, same names / parameter order / defaults as the reference's
PYTHON() declarations; orchestration mirrors the reference function by function, arithmetic happens in the C ABI.

  advectSemiLagrange                         source/plugin/advection.cpp:443-461
  computePressureRhs / solvePressureSystem / correctVelocity / solvePressure   source/plugin/pressure.cpp:277-523
  mapPartsToMAC / mapMACToParts / flipVelocityUpdate / mapPartsToGrid(+Vec3) / mapGridToParts(+Vec3)
                                             source/plugin/flip.cpp:637-742
  setWallBcs / addBuoyancy / addGravity      source/plugin/extforces.cpp (SURVEY 8f-1 glue)
"""
import ctypes
import functools
import time

import numpy as np
import torch

from . import core
from .core import (FlagGrid, Grid, GridBase, LevelsetGrid, MACGrid, VecGrid, _ptr, _to_vec3, vec3)

PcNone, PcMIC, PcMGDynamic, PcMGStatic = 0, 1, 2, 3
IntEuler, IntRK2, IntRK4 = 0, 1, 2

_UNIVERSAL = ("notiming", "parent", "name", "nocheck", "solver")   # codegen_python.cpp:37, pconvert.cpp:461,477
_timings = {}
_last_cg = {}


def _coerce(v, default):
    """fromPy<bool> / fromPy<int>, pconvert.cpp:170-181, 212-215: a bool parameter takes a Python bool only; an int parameter
    takes an int, or a float within 1e-5 of an integer"""
    if isinstance(default, bool):
        if not isinstance(v, bool):
            raise RuntimeError("argument is not a boolean")
        return v
    if isinstance(default, int):
        if isinstance(v, (int, np.integer)):
            return int(v)
        if isinstance(v, (float, np.floating)):
            a = float(v)
            if abs(a - np.floor(a + 0.5)) > 1e-5:
                raise RuntimeError("argument is not an int")
            return int(a + 0.5)
        raise RuntimeError("argument is not an int")
    return v


def plugin(fn):
    """Wrapper every PYTHON() symbol gets in the reference (codegen_python.cpp:32-58): universal kwargs, per-plugin
    wall timer (pclass.cpp:36-41), unknown arguments -> RuntimeError (pconvert.cpp:460-474), bool / int argument conversion
    rules (typed by the parameter's default value)."""
    import inspect
    params = list(inspect.signature(fn).parameters.values())
    typed = {p.name: p.default for p in params if isinstance(p.default, (bool, int)) and p.default is not None}
    pos_typed = [(i, p.default) for i, p in enumerate(params) if p.name in typed]

    @functools.wraps(fn)
    def w(*args, **kw):
        notiming = kw.pop("notiming", False)
        for k in _UNIVERSAL[1:]:
            kw.pop(k, None)
        if typed:
            if any(i < len(args) for i, _ in pos_typed):
                args = list(args)
                for i, d in pos_typed:
                    if i < len(args):
                        args[i] = _coerce(args[i], d)
            for k in kw:
                if k in typed:
                    kw[k] = _coerce(kw[k], typed[k])
        t0 = time.time()
        try:
            r = fn(*args, **kw)
        except TypeError as e:
            msg = str(e)
            if "unexpected keyword argument" in msg:
                raise RuntimeError("Argument %s unknown" % msg.split("argument")[-1].strip())
            raise RuntimeError(msg)
        if not notiming:
            rec = _timings.setdefault(fn.__name__, [0, 0.0])
            rec[0] += 1
            rec[1] += time.time() - t0
        return r
    return w


class Timings(object):
    """TimingData, timing.{h,cpp}: per-plugin wall time (host clock; kernels are asynchronous on the stream)."""
    def display(self):
        for k, (n, t) in sorted(_timings.items(), key=lambda kv: -kv[1][1]):
            print("[%8.3fs] %s (%d calls)" % (t, k, n))
    def saveMean(self, filename):
        with open(filename, "w") as f:
            for k, (n, t) in _timings.items():
                f.write("%s: %f\n" % (k, t / max(n, 1)))
    def step(self): pass


def _chk(obj, cls, what):
    if not isinstance(obj, cls):
        raise RuntimeError("can't convert argument to %s*" % what)
    return obj


def _opt(obj, cls, what):
    if obj is None or (isinstance(obj, int) and obj == 0):   # None or int 0 is NULL (pclass.cpp:128-134)
        return None
    return _chk(obj, cls, what)


# =========================================================================================================
# advection
# =========================================================================================================
@plugin
def advectSemiLagrange(flags, vel, grid, order=1, strength=1.0, orderSpace=1, openBounds=False, boundaryWidth=-1,
                       clampMode=2, orderTrace=1):
    _chk(flags, FlagGrid, "FlagGrid")
    _chk(vel, MACGrid, "MACGrid")
    _chk(grid, GridBase, "GridBase")
    if order not in (1, 2):
        raise RuntimeError("AdvectSemiLagrange: Only order 1 (regular SL) and 2 (MacCormack) supported")
    if orderSpace not in (1, 2):
        raise RuntimeError("Unknown interpolation order %s" % orderSpace)      # getInterpolatedHi, grid.h:157
    s = flags.parent
    lib, st = s.lib, s.stream
    sx, sy, sz = flags.dims
    dt = s.getDt()
    t = grid.getType()
    if t & GridBase.TypeMAC:
        # fnAdvectSemiLagrange<MACGrid>, advection.cpp:407-437
        # (the three temp grids are written in every cell by their kernels -- the semi-Lagrange steps put the zeros of a fresh grid
        # on the border themselves -- so they come from the pool without the constructor's clear)
        fwd = _scratch_grid(s, MACGrid)
        lib.call("mf_semi_lagrange_mac", sx, sy, sz, vel.ptr, fwd.ptr, grid.ptr, dt, int(orderTrace), int(orderSpace), st)
        if order == 1:
            _apply_outflow_bc(flags, fwd, grid, dt)
            grid.swap(fwd)
        else:
            bwd, newg = _scratch_grid(s, MACGrid), _scratch_grid(s, MACGrid)
            lib.call("mf_semi_lagrange_mac", sx, sy, sz, vel.ptr, bwd.ptr, fwd.ptr, -dt, int(orderTrace), int(orderSpace), st)
            # MacCormackCorrectMAC + MacCormackClampMAC, fused (the clamp reads the corrected value of its own cell only)
            lib.call("mf_maccormack_correct_clamp_mac", sx, sy, sz, flags.ptr, vel.ptr, newg.ptr, grid.ptr, fwd.ptr, bwd.ptr, float(strength),
                     dt, int(clampMode), st)
            _apply_outflow_bc(flags, newg, grid, dt)
            grid.swap(newg)
    elif t & (GridBase.TypeReal | GridBase.TypeVec3):
        # fnAdvectSemiLagrange<GridType>, advection.cpp:293-322
        ncomp = 1 if (t & GridBase.TypeReal) else 3
        G = type(grid)
        sl = "mf_semi_lagrange_real" if ncomp == 1 else "mf_semi_lagrange_vec3"
        fwd = _scratch_grid(s, G)
        lib.call(sl, sx, sy, sz, vel.ptr, fwd.ptr, grid.ptr, dt, int(orderTrace), int(orderSpace), st)
        if order == 1:
            grid.swap(fwd)
        else:
            bwd, newg = _scratch_grid(s, G), _scratch_grid(s, G)
            lib.call(sl, sx, sy, sz, vel.ptr, bwd.ptr, fwd.ptr, -dt, int(orderTrace), int(orderSpace), st)
            lib.call("mf_maccormack_correct_clamp", sx, sy, sz, ncomp, flags.ptr, vel.ptr, newg.ptr, grid.ptr, fwd.ptr, bwd.ptr,
                     float(strength), dt, int(clampMode), st)
            grid.swap(newg)
    else:
        raise RuntimeError("AdvectSemiLagrange: Grid Type is not supported (only Real, Vec3, MAC, Levelset)")


def _apply_outflow_bc(flags, vel, velPrev, dt):
    """applyOutflowBC, advection.cpp:388-392 (temp MAC grid so vel is not overwritten while it is read)"""
    s = flags.parent
    if not getattr(flags, "_may_have_outflow", True):
        return
    velDst = MACGrid(s)
    s.lib.call("mf_apply_outflow_bc", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, velPrev.ptr, velDst.ptr, float(dt), s.stream)


# =========================================================================================================
# pressure projection
# =========================================================================================================
@plugin
def computePressureRhs(rhs, vel, pressure, flags, cgAccuracy=1e-3, phi=None, perCellCorr=None, fractions=None, obvel=None,
                       gfClamp=1e-04, cgMaxIterFac=1.5, precondition=True, preconditioner=PcMIC,
                       enforceCompatibility=False, useL2Norm=False, zeroPressureFixing=False, curv=None, surfTens=0.):
    _chk(rhs, Grid, "Grid<Real>"); _chk(vel, MACGrid, "MACGrid"); _chk(flags, FlagGrid, "FlagGrid")
    phi, perCellCorr, curv = _opt(phi, Grid, "Grid<Real>"), _opt(perCellCorr, Grid, "Grid<Real>"), _opt(curv, Grid, "Grid<Real>")
    fractions, obvel = _opt(fractions, MACGrid, "MACGrid"), _opt(obvel, MACGrid, "MACGrid")
    s = flags.parent
    cnt, sm = ctypes.c_int32(0), ctypes.c_double(0.0)
    need = bool(enforceCompatibility)
    s.lib.call("mf_make_rhs", flags.sx, flags.sy, flags.sz, flags.ptr, rhs.ptr, vel.ptr,
               None if perCellCorr is None else perCellCorr.ptr, None if fractions is None else fractions.ptr,
               None if obvel is None else obvel.ptr, None if phi is None else phi.ptr, None if curv is None else curv.ptr,
               float(surfTens), float(gfClamp), ctypes.byref(cnt) if need else None, ctypes.byref(sm) if need else None, s.stream)
    if enforceCompatibility:
        # rhs += (Real)(-kernMakeRhs.sum / (Real)kernMakeRhs.cnt), pressure.cpp:297-298
        corr = np.float32(-sm.value / float(np.float32(cnt.value)))
        rhs.addConst(float(corr))


@plugin
def solvePressureSystem(rhs, vel, pressure, flags, cgAccuracy=1e-3, phi=None, perCellCorr=None, fractions=None, gfClamp=1e-04,
                        cgMaxIterFac=1.5, precondition=True, preconditioner=PcMIC, enforceCompatibility=False,
                        useL2Norm=False, zeroPressureFixing=False, curv=None, surfTens=0.):
    _chk(rhs, Grid, "Grid<Real>"); _chk(vel, MACGrid, "MACGrid"); _chk(pressure, Grid, "Grid<Real>"); _chk(flags, FlagGrid, "FlagGrid")
    phi = _opt(phi, Grid, "Grid<Real>")
    fractions = _opt(fractions, MACGrid, "MACGrid")
    if precondition is False:
        preconditioner = PcNone
    s = flags.parent
    lib, st = s.lib, s.stream
    sx, sy, sz = flags.dims
    # reserve temp grids, pressure.cpp:332-338
    residual, search, A0, Ai, Aj, Ak, tmp = (Grid(s) for _ in range(7))
    lib.call("mf_make_laplace_matrix", sx, sy, sz, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr,
             None if fractions is None else fractions.ptr, st)
    if phi is not None:
        lib.call("mf_apply_ghost_fluid_diagonal", sx, sy, sz, A0.ptr, flags.ptr, phi.ptr, float(gfClamp), st)
    if zeroPressureFixing or cgAccuracy < 1e-07:
        _fix_pressure(flags, rhs, A0, Ai, Aj, Ak)
    if preconditioner in (PcNone, PcMIC):
        gmax = max(sx, sy, sz)
        maxIter = int(np.float32(cgMaxIterFac) * np.float32(gmax)) * (1 if flags.is3D() else 4)   # pressure.cpp:410
        pca0 = Grid(s)
        pca1, pca2, pca3 = Grid(s), Grid(s), Grid(s)   # allocated (and zeroed) by the reference as well, :412-415
        if preconditioner == PcNone:
            # setICPreconditioner(PC_None, ...) asserts in the reference (conjugategrad.cpp:312); keep the behaviour
            raise RuntimeError("GridCg<APPLYMAT>::setICPreconditioner: Invalid method specified.")
        pc = 2   # PC_mICP ; 2-D degrades to PC_None inside the solver (conjugategrad.cpp:315-321)
    elif preconditioner in (PcMGDynamic, PcMGStatic):
        raise RuntimeError("solvePressure: multigrid preconditioners (source/multigrid.cpp) are outside the MI355X hot path")
    else:
        maxIter, pc, pca0 = 0, 0, Grid(s)
    out = (ctypes.c_float * 3)()
    lib.call("mf_cg_solve", sx, sy, sz, flags.ptr, pressure.ptr, rhs.ptr, residual.ptr, search.ptr, tmp.ptr,
             A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, pca0.ptr, pc, float(cgAccuracy), int(maxIter), int(bool(useL2Norm)), out, st)
    _last_cg["iterations"], _last_cg["residual"] = int(out[0]), float(out[1])


@plugin
def cgSolveDiffusion(flags, grid, alpha=0.25, cgMaxIterFac=1.0, cgAccuracy=1e-4):
    """conjugategrad.cpp:350-423: (I + alpha*L) u = grid, unpreconditioned GridCg<ApplyMatrix/2D>; Vec3 / MAC grids component by
    component (2 components in 2-D)"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(grid, GridBase, "GridBase")
    s = flags.parent
    lib, st = s.lib, s.stream
    sx, sy, sz = flags.dims
    rhs, residual, search, tmp, A0, Ai, Aj, Ak = (Grid(s) for _ in range(8))
    dummy = FlagGrid(s)
    dummy.setConst(core.TypeFluid)
    lib.call("mf_make_laplace_matrix", sx, sy, sz, dummy.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, None, st)
    lib.call("mf_diffusion_matrix", sx, sy, sz, flags.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, float(alpha), st)
    maxIter = int(np.float32(cgMaxIterFac) * np.float32(max(sx, sy, sz))) * (1 if flags.is3D() else 4)
    out = (ctypes.c_float * 3)()
    none = Grid(s)

    def solve(u):
        rhs.copyFrom(u)
        lib.call("mf_cg_solve", sx, sy, sz, flags.ptr, u.ptr, rhs.ptr, residual.ptr, search.ptr, tmp.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr,
                 none.ptr, PcNone, float(cgAccuracy), int(maxIter), 1, out, st)   # GridCgInterface() : mUseL2Norm(true), conjugategrad.h:31
        _last_cg["iterations"], _last_cg["residual"] = int(out[0]), float(out[1])

    t = grid.getType()
    if t & GridBase.TypeReal:
        solve(grid)
    elif t & (GridBase.TypeVec3 | GridBase.TypeMAC):
        u = Grid(s)
        for comp in range(3 if grid.is3D() else 2):
            lib.call("mf_copy_f32", u.n, u.ptr, _ptr(grid.data[comp * grid.n:]), st)
            solve(u)
            lib.call("mf_copy_f32", u.n, _ptr(grid.data[comp * grid.n:]), u.ptr, st)
    else:
        raise RuntimeError("cgSolveDiffusion: Grid Type is not supported (only Real, Vec3, MAC, or Levelset)")


def _fix_pressure(flags, rhs, A0, Ai, Aj, Ak):
    """zero-pressure fixing, pressure.cpp:349-390"""
    s = flags.parent
    ne = ctypes.c_int32(0)
    s.lib.call("mf_count_empty_cells", flags.n, flags.ptr, ctypes.byref(ne), s.stream)
    if ne.value != 0:
        return
    sx, sy, sz = flags.dims
    f = flags.data.view(sz, sy, sx)
    top = (sx // 2, sy - 1, sz // 2 if flags.is3D() else 0)
    fix = -1
    for dy in (0, 1, 2):
        i, j, k = top[0], top[1] - dy, top[2]
        if int(f[k, j, i].item()) & core.TypeFluid:
            fix = i + sx * (j + sy * k)
            break
    if fix == -1:
        inner = f[(slice(1, -1) if flags.is3D() else slice(None)), 1:-1, 1:-1]
        nz = torch.nonzero((inner & core.TypeFluid) != 0)
        if nz.numel():
            k, j, i = (int(v) for v in nz[0])
            k = k + 1 if flags.is3D() else 0
            fix = (i + 1) + sx * ((j + 1) + sy * k)
    if fix >= 0:
        s.lib.call("mf_fix_pressure", sx, sy, sz, int(fix), 0.0, rhs.ptr, A0.ptr, Ai.ptr, Aj.ptr, Ak.ptr, s.stream)


@plugin
def correctVelocity(vel, pressure, flags, cgAccuracy=1e-3, phi=None, perCellCorr=None, fractions=None, gfClamp=1e-04,
                    cgMaxIterFac=1.5, precondition=True, preconditioner=PcMIC, enforceCompatibility=False, useL2Norm=False,
                    zeroPressureFixing=False, curv=None, surfTens=0.):
    _chk(vel, MACGrid, "MACGrid"); _chk(pressure, Grid, "Grid<Real>"); _chk(flags, FlagGrid, "FlagGrid")
    phi, curv = _opt(phi, Grid, "Grid<Real>"), _opt(curv, Grid, "Grid<Real>")
    s = flags.parent
    sx, sy, sz = flags.dims
    s.lib.call("mf_correct_velocity", sx, sy, sz, flags.ptr, vel.ptr, pressure.ptr, s.stream)
    if phi is not None:
        s.lib.call("mf_correct_velocity_ghost_fluid", sx, sy, sz, vel.ptr, flags.ptr, pressure.ptr, phi.ptr, float(gfClamp),
                   None if curv is None else curv.ptr, float(surfTens), s.stream)
        s.lib.call("mf_replace_clamped_ghost_fluid_vels", sx, sy, sz, vel.ptr, flags.ptr, pressure.ptr, phi.ptr, float(gfClamp), s.stream)


@plugin
def solvePressure(vel, pressure, flags, cgAccuracy=1e-3, phi=None, perCellCorr=None, fractions=None, obvel=None, gfClamp=1e-04,
                  cgMaxIterFac=1.5, precondition=True, preconditioner=PcMIC, enforceCompatibility=False, useL2Norm=False,
                  zeroPressureFixing=False, curv=None, surfTens=0., retRhs=None):
    _chk(vel, MACGrid, "MACGrid")
    if _plain_system(vel, pressure, flags, cgAccuracy, phi, perCellCorr, fractions, obvel, precondition, preconditioner,
                     enforceCompatibility, zeroPressureFixing, curv, surfTens):
        # the plain case (every smoke scene, FLIP without ghost fluid): rhs + packed matrix in one pass, MIC factor and PCG on the
        # packed bytes -- the coefficient grids A0 / Ai / Aj / Ak and pca1..3 of pressure.cpp:332-338, 412-415 are never made
        s = flags.parent
        sx, sy, sz = flags.dims
        rhs, residual, search, tmp, pca0 = (_scratch_grid(s) for _ in range(5))
        maxIter = int(np.float32(cgMaxIterFac) * np.float32(max(sx, sy, sz)))        # pressure.cpp:410 (3D)
        out = (ctypes.c_float * 3)()
        took = True
        try:
            s.lib.call("mf_solve_pressure_fused", sx, sy, sz, flags.ptr, vel.ptr, pressure.ptr, rhs.ptr, residual.ptr, search.ptr, tmp.ptr,
                       pca0.ptr, float(cgAccuracy), int(maxIter), int(bool(useL2Norm)), out, s.stream)
        except RuntimeError as e:
            # the library declines before it touches anything (e.g. mf_set_mic_mode("levels")): the three-call path below
            if not str(e).startswith("mf_solve_pressure_fused: needs"):
                raise
            took = False
        if took:
            _last_cg["iterations"], _last_cg["residual"] = int(out[0]), float(out[1])
            correctVelocity(vel, pressure, flags, notiming=True)
            if retRhs is not None and not (isinstance(retRhs, int) and retRhs == 0):
                _chk(retRhs, Grid, "Grid<Real>").copyFrom(rhs)
            return
        del rhs, residual, search, tmp, pca0
    rhs = Grid(vel.parent)
    common = dict(cgAccuracy=cgAccuracy, phi=phi, perCellCorr=perCellCorr, fractions=fractions, gfClamp=gfClamp,
                  cgMaxIterFac=cgMaxIterFac, precondition=precondition, preconditioner=preconditioner,
                  enforceCompatibility=enforceCompatibility, useL2Norm=useL2Norm, zeroPressureFixing=zeroPressureFixing,
                  curv=curv, surfTens=surfTens, notiming=True)
    computePressureRhs(rhs, vel, pressure, flags, obvel=obvel, **common)
    solvePressureSystem(rhs, vel, pressure, flags, **common)
    correctVelocity(vel, pressure, flags, **common)
    if retRhs is not None and not (isinstance(retRhs, int) and retRhs == 0):
        _chk(retRhs, Grid, "Grid<Real>").copyFrom(rhs)


def _scratch_grid(s, cls=None):
    """a temp grid from the solver's pool WITHOUT the clear of the Grid constructor, for callees that overwrite every cell"""
    cls = cls or Grid
    g = cls.__new__(cls)
    core.PbClass.__init__(g, s, "")
    g.sx, g.sy, g.sz = s.mGridSize
    g.n = g.sx * g.sy * g.sz
    g.data = s._alloc(g._kind, zero=False)
    g._external = False
    return g


def _plain_system(vel, pressure, flags, cgAccuracy, phi, perCellCorr, fractions, obvel, precondition, preconditioner,
                  enforceCompatibility, zeroPressureFixing, curv, surfTens):
    """can mf_solve_pressure_fused take this solvePressure call?  (3D, rows of a multiple of 8 cells, MIC, MakeLaplaceMatrix system
    with the plain MakeRhs)"""
    if not (isinstance(pressure, Grid) and isinstance(flags, FlagGrid)) or not flags.is3D():
        return False
    none = lambda v: v is None or (isinstance(v, int) and not isinstance(v, bool) and v == 0)
    if not (none(phi) and none(perCellCorr) and none(fractions) and none(obvel) and none(curv)):
        return False
    if precondition is not True or preconditioner != PcMIC or enforceCompatibility or zeroPressureFixing or cgAccuracy < 1e-07:
        return False
    if flags.sx % 8 != 0 or _fused_off:
        return False
    return (vel.sx, vel.sy, vel.sz) == flags.dims == (pressure.sx, pressure.sy, pressure.sz)


_fused_off = bool(__import__("os").environ.get("MF_NO_FUSED_SETUP"))


def lastCgStats():
    """iterations / residual of the most recent solvePressureSystem (the reference prints them at debug level 2,
    pressure.cpp:442)"""
    return dict(_last_cg)


# =========================================================================================================
# FLIP transfers
# =========================================================================================================
_deterministic_p2g = True


def setDeterministicP2G(on):
    """particle->grid transfers: True (default) = sums in particle-index order, bit-identical to the reference's
    single-threaded scatter (flip.cpp:619) on every run (parallel ordered gather, p2g_ordered.hip; 2.2 ms for 3.8 M
    particles at 128^3); False = fp32 atomics with block-private LDS accumulation (1.6 ms, last bits depend on the order
    of arrival)"""
    global _deterministic_p2g
    _deterministic_p2g = bool(on)


def _pargs(parts, ptype):
    return (parts.np, parts.cap, _ptr(parts.pos), _ptr(parts.flag)), (None if ptype is None else ptype.ptr)


@plugin
def mapPartsToMAC(flags, vel, velOld, parts, partVel, weight=None, ptype=None, exclude=0):
    _chk(flags, FlagGrid, "FlagGrid"); _chk(vel, MACGrid, "MACGrid"); _chk(velOld, MACGrid, "MACGrid")
    weight = _opt(weight, VecGrid, "Grid<Vec3>")
    s = flags.parent
    w = weight if weight is not None else VecGrid(s)
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_map_parts_to_mac", flags.sx, flags.sy, flags.sz, vel.ptr, velOld.ptr, w.ptr, np_, cap, pos, pfl,
               partVel.ptr, pt, int(exclude), int(_deterministic_p2g), s.stream)


@plugin
def mapMACToParts(flags, vel, parts, partVel, ptype=None, exclude=0):
    s = flags.parent
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_map_mac_to_parts", flags.sx, flags.sy, flags.sz, vel.ptr, np_, cap, pos, pfl, partVel.ptr, pt, int(exclude), s.stream)


@plugin
def flipVelocityUpdate(flags, vel, velOld, parts, partVel, flipRatio, ptype=None, exclude=0):
    s = flags.parent
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_flip_velocity_update", flags.sx, flags.sy, flags.sz, vel.ptr, velOld.ptr, np_, cap, pos, pfl,
               partVel.ptr, float(flipRatio), pt, int(exclude), s.stream)


@plugin
def getComponent(source, target, component):
    """grid.cpp:743-746: target = source[component] (a plane copy in the SoA layout)"""
    _chk(source, VecGrid, "Grid<Vec3>"); _chk(target, Grid, "Grid<Real>")
    s = source.parent
    s.lib.call("mf_copy_f32", target.n, target.ptr, _ptr(source.data[int(component) * source.n:]), s.stream)


@plugin
def setComponent(source, target, component):
    """grid.cpp:748-751: target[component] = source"""
    _chk(source, Grid, "Grid<Real>"); _chk(target, VecGrid, "Grid<Vec3>")
    s = source.parent
    s.lib.call("mf_copy_f32", source.n, _ptr(target.data[int(component) * target.n:]), source.ptr, s.stream)


@plugin
def resetOutflow(flags, phi=None, parts=None, real=None, index=None, indexSys=None):
    """extforces.cpp:134-161 (index / indexSys only speed up the reference's particle loop; same result without)"""
    _chk(flags, FlagGrid, "FlagGrid")
    phi, real = _opt(phi, Grid, "Grid<Real>"), _opt(real, Grid, "Grid<Real>")
    s = flags.parent
    np_, cap, pos, pfl = (0, 0, None, None) if parts is None else _pargs(parts, None)[0]
    s.lib.call("mf_reset_outflow", flags.sx, flags.sy, flags.sz, flags.ptr, None if phi is None else phi.ptr,
               None if real is None else real.ptr, np_, cap, pos, pfl, s.stream)


@plugin
def apicMapPartsToMAC(flags, vel, parts, partVel, cpx, cpy, cpz, mass=None, ptype=None, exclude=0):
    """plugin/apic.cpp:92-110; the scatter is summed in particle-index order per node (= the reference's serial kernel)"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(vel, MACGrid, "MACGrid")
    mass = _opt(mass, MACGrid, "MACGrid")
    s = flags.parent
    m = mass if mass is not None else MACGrid(s)
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_apic_map_parts_to_mac", flags.sx, flags.sy, flags.sz, vel.ptr, m.ptr, np_, cap, pos, pfl, partVel.ptr,
               cpx.ptr, cpy.ptr, cpz.ptr, pt, int(exclude), s.stream)


@plugin
def apicMapMACGridToParts(partVel, cpx, cpy, cpz, parts, vel, flags, ptype=None, exclude=0):
    """plugin/apic.cpp:175-181"""
    _chk(vel, MACGrid, "MACGrid")
    s = flags.parent
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_apic_map_mac_to_parts", flags.sx, flags.sy, flags.sz, vel.ptr, np_, cap, pos, pfl, partVel.ptr,
               cpx.ptr, cpy.ptr, cpz.ptr, pt, int(exclude), s.stream)


def _map_parts_to_grid(flags, target, parts, source, ncomp):
    s = flags.parent
    tmp = Grid(s)
    (np_, cap, pos, pfl), _ = _pargs(parts, None)
    s.lib.call("mf_map_parts_to_grid", flags.sx, flags.sy, flags.sz, ncomp, target.ptr, tmp.ptr, np_, cap, pos, pfl,
               source.ptr, int(_deterministic_p2g), s.stream)


@plugin
def mapPartsToGrid(flags, target, parts, source): _map_parts_to_grid(flags, _chk(target, Grid, "Grid<Real>"), parts, source, 1)


@plugin
def mapPartsToGridVec3(flags, target, parts, source): _map_parts_to_grid(flags, _chk(target, VecGrid, "Grid<Vec3>"), parts, source, 3)


@plugin
def mapGridToParts(source, parts, target):
    s = source.parent
    (np_, cap, pos, pfl), _ = _pargs(parts, None)
    s.lib.call("mf_map_grid_to_parts", source.sx, source.sy, source.sz, 1, source.ptr, np_, cap, pos, pfl, target.ptr, s.stream)


@plugin
def mapGridToPartsVec3(source, parts, target):
    s = source.parent
    (np_, cap, pos, pfl), _ = _pargs(parts, None)
    s.lib.call("mf_map_grid_to_parts", source.sx, source.sy, source.sz, 3, source.ptr, np_, cap, pos, pfl, target.ptr, s.stream)


# =========================================================================================================
# FLIP glue
# =========================================================================================================
@plugin
def extrapolateMACSimple(flags, vel, distance=4, phiObs=None, intoObs=False):
    """fastmarch.cpp:337-376"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(vel, MACGrid, "MACGrid")
    if phiObs is not None and not (isinstance(phiObs, int) and phiObs == 0):
        raise RuntimeError("extrapolateMACSimple: the phiObs variant (knUnprojectNormalComp) is outside the hot path")
    s = flags.parent
    tmp, velTmp = core.IntGrid(s), MACGrid(s)
    s.lib.call("mf_extrapolate_mac_simple", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, int(distance), int(bool(intoObs)),
               tmp.ptr, velTmp.ptr, s.stream)


@plugin
def extrapolateMACFromWeight(vel, weight, distance=2):
    """fastmarch.cpp:415-430"""
    _chk(vel, MACGrid, "MACGrid"); _chk(weight, VecGrid, "Grid<Vec3>")
    s = vel.parent
    s.lib.call("mf_extrapolate_mac_from_weight", vel.sx, vel.sy, vel.sz, vel.ptr, weight.ptr, int(distance), s.stream)


@plugin
def markFluidCells(parts, flags, phiObs=None, ptype=None, exclude=0):
    """flip.cpp:166-188"""
    _chk(flags, FlagGrid, "FlagGrid")
    phiObs = _opt(phiObs, Grid, "Grid<Real>")
    s = flags.parent
    ftmp = core.IntGrid(s) if phiObs is not None else None
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_mark_fluid_cells", flags.sx, flags.sy, flags.sz, flags.ptr, np_, cap, pos, pfl, pt, int(exclude),
               None if phiObs is None else phiObs.ptr, None if ftmp is None else ftmp.ptr, s.stream)


@plugin
def pushOutofObs(parts, flags, phiObs, shift=0., thresh=0., ptype=None, exclude=0):
    """flip.cpp:584-602"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(phiObs, Grid, "Grid<Real>")
    s = flags.parent
    (np_, cap, pos, pfl), pt = _pargs(parts, ptype)
    s.lib.call("mf_push_out_of_obs", flags.sx, flags.sy, flags.sz, np_, cap, pos, pfl, phiObs.ptr, float(shift), float(thresh),
               pt, int(exclude), s.stream)


# =========================================================================================================
# free-surface pieces of scenes/benchmark_dam.py
# =========================================================================================================
@plugin
def gridParticleIndex(parts, indexSys, flags, index, counter=None):
    """flip.cpp:273-320: index = first slot per cell, indexSys = particle indices ordered by (cell, particle)"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(index, core.IntGrid, "Grid<int>")
    s = flags.parent
    cnt = counter if counter is not None else core.IntGrid(s)
    np_ = parts.np
    if indexSys.data.numel() < max(np_, 1):
        indexSys.data = torch.zeros(max(parts.cap, np_, 1), dtype=torch.int32, device=s.device)
    keys = torch.empty(2 * max(np_, 1), dtype=torch.int32, device=s.device)
    vals = torch.empty(2 * max(np_, 1), dtype=torch.int32, device=s.device)
    n_idx = ctypes.c_int64(0)
    s.lib.call("mf_grid_particle_index", flags.sx, flags.sy, flags.sz, np_, parts.cap, _ptr(parts.pos), _ptr(parts.flag),
               _ptr(indexSys.data), index.ptr, cnt.ptr, _ptr(keys), _ptr(vals), ctypes.byref(n_idx), s.stream)
    indexSys.np = int(n_idx.value)


@plugin
def unionParticleLevelset(parts, indexSys, flags, index, phi, radiusFactor=1., ptype=None, exclude=0):
    """flip.cpp:322-363"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(phi, Grid, "LevelsetGrid")
    s = flags.parent
    s.lib.call("mf_union_particle_levelset", flags.sx, flags.sy, flags.sz, parts.np, parts.cap, _ptr(parts.pos),
               _ptr(indexSys.data), int(indexSys.np), index.ptr, phi.ptr, float(radiusFactor),
               None if ptype is None else ptype.ptr, int(exclude), s.stream)


@plugin
def extrapolateLsSimple(phi, distance=4, inside=False, include_walls=False):
    """fastmarch.cpp:472-522"""
    _chk(phi, Grid, "Grid<Real>")
    s = phi.parent
    tmp = core.IntGrid(s)
    s.lib.call("mf_extrapolate_ls_simple", phi.sx, phi.sy, phi.sz, phi.ptr, int(distance), int(bool(inside)),
               int(bool(include_walls)), tmp.ptr, s.stream)


@plugin
def setPartType(parts, ptype, mark, stype, flags, cflag):
    """ptsplugins.cpp:56-65"""
    _chk(flags, FlagGrid, "FlagGrid")
    s = flags.parent
    s.lib.call("mf_set_part_type", flags.sx, flags.sy, flags.sz, flags.ptr, parts.np, parts.cap, _ptr(parts.pos), ptype.ptr,
               int(mark), int(stype), int(cflag), s.stream)


@plugin
def markIsolatedFluidCell(flags, mark):
    """grid.cpp:987-1011"""
    _chk(flags, FlagGrid, "FlagGrid")
    s = flags.parent
    s.lib.call("mf_mark_isolated_fluid_cell", flags.sx, flags.sy, flags.sz, flags.ptr, int(mark), s.stream)


@plugin
def addForcePvel(vel, a, dt, ptype, exclude):
    """ptsplugins.cpp:20-29"""
    a = core._to_vec3(a)
    s = vel.parent
    s.lib.call("mf_add_force_pvel", vel.size(), vel.cap, vel.ptr, float(a.x), float(a.y), float(a.z), float(dt),
               None if ptype is None else ptype.ptr, int(exclude), s.stream)


@plugin
def updateVelocityFromDeltaPos(parts, vel, x_prev, dt, ptype, exclude):
    """ptsplugins.cpp:31-41"""
    s = vel.parent
    s.lib.call("mf_update_velocity_from_delta_pos", parts.np, parts.cap, _ptr(parts.pos), vel.ptr, x_prev.ptr, float(dt),
               None if ptype is None else ptype.ptr, int(exclude), s.stream)


@plugin
def eulerStep(parts, vel, ptype, exclude):
    """ptsplugins.cpp:43-53"""
    s = vel.parent
    s.lib.call("mf_euler_step", parts.np, parts.cap, _ptr(parts.pos), vel.ptr, s.getDt(), None if ptype is None else ptype.ptr,
               int(exclude), s.stream)


# =========================================================================================================
# resampling between grids of different size (plugin/waveletturbulence.cpp:27-78)
# =========================================================================================================
def _size_factor(source, target, scale, offset, size):
    """calcGridSizeFactorMod, waveletturbulence.cpp:27-34 (fp32 Vec3 arithmetic)"""
    f32 = np.float32
    scale, offset = _to_vec3(scale), _to_vec3(offset)
    s1 = tuple(source.parent.globalGridSize())      # whole-domain sizes (a z-slab's grids hold a window of them)
    s2 = list(target.parent.globalGridSize())
    if size is not None:
        size = tuple(int(v) for v in (size if not isinstance(size, vec3) else (size.x, size.y, size.z)))
        for c in range(3):
            if size[c] > 0:
                s2[c] = size[c]
    sf = [f32(f32(f32(s1[c]) / f32(s2[c])) / f32((scale.x, scale.y, scale.z)[c])) for c in range(3)]
    off = [f32(f32(f32(-f32((offset.x, offset.y, offset.z)[c])) * sf[c]) + f32(sf[c] * f32(0.5))) for c in range(3)]
    return [float(v) for v in sf], [float(v) for v in off]


def _interpolate(target, source, scale, offset, size, orderSpace, ncomp):
    if int(orderSpace) not in (1, 2):
        raise RuntimeError("Unknown interpolation order %s" % orderSpace)
    sf, off = _size_factor(source, target, scale, offset, size)
    s = target.parent
    s.lib.call2(source.parent, "mf_interpolate_grid", target.sx, target.sy, target.sz, target.ptr, source.sx, source.sy, source.sz, source.ptr,
               ncomp, sf[0], sf[1], sf[2], off[0], off[1], off[2], int(orderSpace), s.stream)


@plugin
def interpolateGrid(target, source, scale=vec3(1.), offset=vec3(0.), size=None, orderSpace=1):
    _chk(target, Grid, "Grid<Real>"); _chk(source, Grid, "Grid<Real>")
    _interpolate(target, source, scale, offset, size, orderSpace, 1)


@plugin
def interpolateGridVec3(target, source, scale=vec3(1.), offset=vec3(0.), size=None, orderSpace=1):
    _chk(target, VecGrid, "Grid<Vec3>"); _chk(source, VecGrid, "Grid<Vec3>")
    _interpolate(target, source, scale, offset, size, orderSpace, 3)


@plugin
def interpolateMACGrid(target, source, scale=vec3(1.), offset=vec3(0.), size=None, orderSpace=1):
    _chk(target, MACGrid, "MACGrid"); _chk(source, MACGrid, "MACGrid")
    if int(orderSpace) not in (1, 2):
        raise RuntimeError("Unknown interpolation order %s" % orderSpace)
    sf, off = _size_factor(source, target, scale, offset, size)
    s = target.parent
    s.lib.call2(source.parent, "mf_interpolate_mac_grid", target.sx, target.sy, target.sz, target.ptr, source.sx, source.sy, source.sz,
               source.ptr, sf[0], sf[1], sf[2], off[0], off[1], off[2], int(orderSpace), s.stream)


# =========================================================================================================
# wavelet turbulence pieces of scenes/waveletTurbulence.py
# =========================================================================================================
@plugin
def computeEnergy(flags, vel, energy):
    """waveletturbulence.cpp:180-194"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(vel, MACGrid, "MACGrid"); _chk(energy, Grid, "Grid<Real>")
    s = flags.parent
    s.lib.call("mf_compute_energy", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, energy.ptr, s.stream)


@plugin
def computeWaveletCoeffs(input):
    """waveletturbulence.cpp:197-201 -> WaveletNoiseField::computeCoefficients"""
    _chk(input, Grid, "Grid<Real>")
    s = input.parent
    t1, t2 = Grid(s), Grid(s)
    s.lib.call("mf_compute_wavelet_coeffs", input.sx, input.sy, input.sz, input.ptr, t1.ptr, t2.ptr, s.stream)


@plugin
def vorticityConfinement(vel, flags, strength=0., strengthCell=None):
    """extforces.cpp:409-428"""
    _chk(vel, MACGrid, "MACGrid"); _chk(flags, FlagGrid, "FlagGrid")
    strengthCell = _opt(strengthCell, Grid, "Grid<Real>")
    s = flags.parent
    vc, curl, force, nrm = VecGrid(s), VecGrid(s), VecGrid(s), Grid(s)
    s.lib.call("mf_vorticity_confinement", flags.sx, flags.sy, flags.sz, vel.ptr, flags.ptr, float(strength),
               None if strengthCell is None else strengthCell.ptr, vc.ptr, curl.ptr, nrm.ptr, force.ptr, s.stream)


@plugin
def applyNoiseVec3(flags, target, noise, scale=1.0, scaleSpatial=1.0, weight=None, uv=None):
    """waveletturbulence.cpp:120-178 (weight and uv grids of any size: sampled with getInterpolated when it differs from the target's)"""
    _chk(flags, FlagGrid, "FlagGrid"); _chk(target, VecGrid, "Grid<Vec3>")
    weight = _opt(weight, Grid, "Grid<Real>")
    uv = _opt(uv, VecGrid, "Grid<Vec3>")
    if uv is not None and weight is not None and uv.dims != weight.dims:
        raise RuntimeError("UV and weight grid have to match!")
    s = flags.parent
    w = (None, 0, 0, 0) if weight is None else (weight.ptr, weight.sx, weight.sy, weight.sz)
    u = (None, 0, 0, 0) if uv is None else (uv.ptr, uv.sx, uv.sy, uv.sz)
    src = uv.parent if uv is not None else (s if weight is None else weight.parent)
    s.lib.call2(src, "mf_apply_noise_vec3", flags.sx, flags.sy, flags.sz, flags.ptr, target.ptr, _ptr(noise._tile), noise._params(),
                float(scale), float(scaleSpatial), w[0], w[1], w[2], w[3], u[0], u[1], u[2], u[3], s.stream)


@plugin
def setOpenBound(flags, bWidth, openBound="", type=16 | 4):
    """extforces.cpp:106-131 (scene set-up: flag edit on the host)"""
    _chk(flags, FlagGrid, "FlagGrid")
    if openBound == "":
        return
    lo = {c: (c in openBound) for c in "xyz"}
    up = {c: (c.upper() in openBound) for c in "xyz"}
    f = flags.to_numpy()
    sz, sy, sx = f.shape
    k, j, i = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
    bw = int(bWidth)
    loX, loY = lo["x"] & (i <= bw), lo["y"] & (j <= bw)
    upX, upY = up["x"] & (i >= sx - bw - 1), up["y"] & (j >= sy - bw - 1)
    inI, inJ = (i > bw) & (i < sx - bw - 1), (j > bw) & (j < sy - bw - 1)
    obs = (f & core.TypeObstacle) != 0
    if not flags.is3D():
        m = (loX | upX | loY | upY) & (loX | upX | inI) & (loY | upY | inJ) & obs
    else:
        loZ, upZ = lo["z"] & (k <= bw), up["z"] & (k >= sz - bw - 1)
        inK = (k > bw) & (k < sz - bw - 1)
        m = (loX | upX | loY | upY | loZ | upZ) & (loX | upX | inI) & (loY | upY | inJ) & (loZ | upZ | inK) & obs
    f[m] = int(type)
    flags.from_numpy(f)


# =========================================================================================================
# glue
# =========================================================================================================
@plugin
def setWallBcs(flags, vel, obvel=None, fractions=None, phiObs=None, boundaryWidth=0):
    obvel = _opt(obvel, MACGrid, "MACGrid")
    if phiObs is not None and fractions is not None:
        raise RuntimeError("setWallBcs: the fill-fraction variant (KnSetWallBcsFrac, extforces.cpp:240-324) is outside the hot path")
    s = flags.parent
    s.lib.call("mf_set_wall_bcs", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, None if obvel is None else obvel.ptr, s.stream)


def _f32(x): return np.float32(x)


@plugin
def addBuoyancy(flags, density, vel, gravity, coefficient=1., scale=True):
    g = _to_vec3(gravity)
    s = flags.parent
    gridScale = _f32(flags.getDx()) if scale else _f32(1)     # float gridScale = scale ? flags.getDx() : 1
    dt = _f32(s.getDt())
    f = [_f32(_f32(_f32(-_f32(c)) * dt) / gridScale) * _f32(coefficient) for c in (g.x, g.y, g.z)]   # -gravity*dt/gridScale*coefficient
    s.lib.call("mf_add_buoyancy", flags.sx, flags.sy, flags.sz, flags.ptr, density.ptr, vel.ptr, float(f[0]), float(f[1]), float(f[2]), s.stream)


@plugin
def addGravity(flags, vel, gravity, exclude=None, scale=True):
    g = _to_vec3(gravity)
    s = flags.parent
    gridScale = _f32(flags.getDx()) if scale else _f32(1)
    dt = _f32(s.getDt())
    f = [_f32(_f32(_f32(c) * dt) / gridScale) for c in (g.x, g.y, g.z)]
    exclude = _opt(exclude, Grid, "Grid<Real>")
    s.lib.call("mf_apply_force", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, float(f[0]), float(f[1]), float(f[2]),
               None if exclude is None else exclude.ptr, 1, s.stream)


@plugin
def addGravityNoScale(flags, vel, gravity, exclude=None): addGravity(flags, vel, gravity, exclude, False, notiming=True)
