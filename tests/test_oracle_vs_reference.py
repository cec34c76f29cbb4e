"""Pins the plain-C oracle restatement (oracle/manta_oracle.c) and the package's host orchestration against the
reference's own compiled C++ (oracle/_ref/libmanta_ref.so, built by oracle/ref.mk).  Bit-exact everywhere."""
import ctypes

import numpy as np
import pytest

import cases
import util
from util import assert_bitexact

pytestmark = pytest.mark.skipif(not util.have_ref(), reason="compiled reference (oracle/_ref) not present")

DIMS = cases.SIZES_3D + [cases.SIZE_2D]


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("fractions", [False, True])
def test_make_laplace_matrix(oracle, dims, fractions):
    flags, fr = cases.laplace_inputs(dims, 3, fractions)
    a = cases.run_laplace_impl(oracle, dims, flags, fr)
    b = cases.run_laplace_ref(dims, flags, fr)
    for x, y, nm in zip(a, b, ("A0", "Ai", "Aj", "Ak")):
        assert_bitexact(x, y, nm)


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("seed", [1, 2])
def test_apply_matrix(oracle, dims, seed):
    flags, A, src = cases.system_inputs(dims, seed)
    assert_bitexact(cases.run_apply_matrix_impl(oracle, dims, flags, A, src), cases.run_apply_matrix_ref(dims, flags, A, src), "ApplyMatrix")


@pytest.mark.parametrize("dims", cases.SIZES_3D)
@pytest.mark.parametrize("seed", [1, 2])
def test_mic(oracle, dims, seed):
    flags, A, src = cases.system_inputs(dims, seed)
    ap, dst = cases.run_mic_impl(oracle, dims, flags, A, src)
    ap_r, dst_r = cases.run_mic_ref(dims, flags, A, src)
    assert_bitexact(ap, ap_r, "Aprecond")
    assert_bitexact(dst, dst_r, "mic apply")


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("pc,acc,iters", [(2, 1e-3, 60), (2, 1e-1, 60), (2, 1e-9, 3), (0, 1e-3, 80)])
def test_cg_solve(oracle, dims, pc, acc, iters):
    flags, A, _ = cases.system_inputs(dims, 5)   # odd seed: empty band on top -> SPD system, CG converges
    rhs = cases.cg_rhs(dims, flags, 5)
    x, st = cases.run_cg_impl(oracle, dims, flags, A, rhs, pc, acc, iters)
    xr, str_ = cases.run_cg_ref(dims, flags, A, rhs, pc, acc, iters)
    assert st[0] == str_[0], (st, str_)
    assert_bitexact(x, xr, "cg solution")
    assert_bitexact(np.float32(st[1:]), np.float32(str_[1:]), "resNorm/sigma")


def test_cg_l2norm(oracle):
    dims = (16, 16, 16)
    flags, A, _ = cases.system_inputs(dims, 5)
    rhs = cases.cg_rhs(dims, flags, 5)
    x, st = cases.run_cg_impl(oracle, dims, flags, A, rhs, 2, 1e-4, 50, useL2=1)
    xr, str_ = cases.run_cg_ref(dims, flags, A, rhs, 2, 1e-4, 50, useL2=1)
    assert st[0] == str_[0]
    assert_bitexact(x, xr, "cg solution (L2 norm)")


@pytest.mark.parametrize("dims", DIMS)
@pytest.mark.parametrize("liquid", [False, True])
def test_solve_pressure(oracle_backend, dims, liquid):
    flags, vel, phi = cases.pressure_inputs(dims, 6, liquid)
    a = cases.run_solve_pressure_pkg(dims, flags, vel, phi)
    b = cases.run_solve_pressure_ref(dims, flags, vel, phi)
    for k in ("rhs", "pressure", "vel"):
        assert_bitexact(a[k], b[k], "solvePressure " + k)


def test_solve_pressure_options(oracle_backend):
    dims = (14, 12, 10)
    flags, vel, phi = cases.pressure_inputs(dims, 8, False)
    kw = dict(cgAccuracy=1e-5, cgMaxIterFac=3.0, enforceCompatibility=True, zeroPressureFixing=True)
    a = cases.run_solve_pressure_pkg(dims, flags, vel, phi, **kw)
    b = cases.run_solve_pressure_ref(dims, flags, vel, phi, **kw)
    for k in ("rhs", "pressure", "vel"):
        assert_bitexact(a[k], b[k], "solvePressure(options) " + k)


@pytest.mark.parametrize("terms,liquid", cases.PRESSURE_OPTIONAL_CASES)
@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
def test_solve_pressure_optional_terms(oracle_backend, dims, terms, liquid):
    """MakeRhs / MakeLaplaceMatrix with perCellCorr, fractions, obvel, curv + surfTens (pressure.cpp:32-84, 277-299): the oracle
    equals the compiled reference bit for bit"""
    flags, vel, phi = cases.pressure_inputs(dims, 12, liquid)
    extra = cases.pressure_optional_terms(dims, 13, terms)
    kw = dict(surfTens=0.7) if "curv" in terms else {}
    a = cases.run_solve_pressure_pkg(dims, flags, vel, phi, extra=extra, **kw)
    b = cases.run_solve_pressure_ref(dims, flags, vel, phi, extra=extra, **kw)
    for k in ("rhs", "pressure", "vel"):
        assert_bitexact(a[k], b[k], "solvePressure(%s) %s" % ("+".join(terms), k))


def test_solve_pressure_pcnone_raises(oracle_backend):
    """the reference asserts for PcNone / precondition=False (SURVEY intro item 1)"""
    dims = (10, 10, 10)
    flags, vel, phi = cases.pressure_inputs(dims, 8, False)
    with pytest.raises(RuntimeError, match="Invalid method specified"):
        cases.run_solve_pressure_pkg(dims, flags, vel, phi, preconditioner=0)
    with pytest.raises(RuntimeError, match="Invalid method specified"):
        cases.run_solve_pressure_pkg(dims, flags, vel, phi, precondition=False)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("order,clampMode", [(1, 2), (2, 1), (2, 2)])
@pytest.mark.parametrize("orderTrace", [1, 2])
def test_advect(oracle_backend, dims, kind, order, clampMode, orderTrace):
    sx, sy, sz = dims
    flags, vel = cases.advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
    field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
    if kind == 1 and sz == 1:
        field[2] = util.rand_real((sz, sy, sx), 12)
    kw = dict(order=order, clampMode=clampMode, orderTrace=orderTrace, strength=0.8 if order == 2 else 1.0)
    a = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    b = cases.run_advect_ref(dims, 0.9, flags, vel, field, kind, **kw)
    assert_bitexact(a, b, "advectSemiLagrange kind=%d" % kind)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("order,orderTrace", [(1, 1), (2, 1), (2, 2)])
def test_advect_cubic(oracle_backend, dims, kind, order, orderTrace):
    """orderSpace=2: cubic interpolation (util/interpolHigh.h) in the semi-Lagrangian gathers -- Real (double-precision coefficient
    expressions), Vec3 and MAC (per-product fp32 rounding), with the linear fall-back where the 4^3 neighbourhood leaves the grid"""
    sx, sy, sz = dims
    flags, vel = cases.advect_inputs(dims, 9, vmax=2.5, outflow=(kind == 2))
    field = util.rand_real((sz, sy, sx), 10) if kind == 0 else util.rand_vel(sx, sy, sz, 10)
    if kind == 1 and sz == 1:
        field[2] = util.rand_real((sz, sy, sx), 12)
    kw = dict(order=order, clampMode=2, orderTrace=orderTrace, orderSpace=2, strength=0.8 if order == 2 else 1.0)
    a = cases.run_advect_pkg(dims, 0.9, flags, vel, field, kind, **kw)
    b = cases.run_advect_ref(dims, 0.9, flags, vel, field, kind, **kw)
    assert_bitexact(a, b, "advectSemiLagrange(orderSpace=2) kind=%d" % kind)


def test_advect_selfadvection_long_traces(oracle_backend):
    """velocity advecting itself with traces leaving the domain (|v| dt up to 6 cells)"""
    dims = (16, 14, 12)
    flags, vel = cases.advect_inputs(dims, 13, vmax=6.0)
    a = cases.run_advect_pkg(dims, 1.0, flags, vel, vel, 2, order=2)
    b = cases.run_advect_ref(dims, 1.0, flags, vel, vel, 2, order=2)
    assert_bitexact(a, b, "self advection")


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
@pytest.mark.parametrize("with_ptype", [False, True])
def test_flip_transfers(oracle_backend, dims, with_ptype):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 14, empty_top=True)
    vel, velOld = util.rand_vel(sx, sy, sz, 15), util.rand_vel(sx, sy, sz, 16)
    pos, pflag, pvel = util.make_particles(flags, 3, 17)
    ptype = exclude = None
    if with_ptype:
        ptype = (np.random.default_rng(18).integers(0, 4, pos.shape[1]) * 2).astype(np.int32)
    a = cases.run_flip_pkg(dims, flags, vel, velOld, pos, pflag, pvel, ptype, 4 if with_ptype else 0)
    b = cases.run_flip_ref(dims, flags, vel, velOld, pos, pflag, pvel, ptype, 4 if with_ptype else 0)
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D, (17, 9, 11)])
def test_cg_solve_diffusion(oracle_backend, dims):
    """cgSolveDiffusion (conjugategrad.cpp:350-423): the unpreconditioned GridCg<ApplyMatrix / ApplyMatrix2D> on (I + alpha L),
    Real grid and MAC grid (component by component) -- SURVEY 8c names it as the Python-level pin of ApplyMatrix"""
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 81, obstacles=True)
    real, vel = util.rand_real((sz, sy, sx), 82), util.rand_vel(sx, sy, sz, 83)
    a, b = cases.run_diffusion_pkg(dims, flags, real, vel), cases.run_diffusion_ref(dims, flags, real, vel)
    assert a["iters"].min() > 3 and np.abs(b["real"] - real).max() > 1e-3
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
def test_reset_outflow(oracle_backend, dims):
    """resetOutflow (extforces.cpp:134-161): flags / phi / real bit for bit; the particles that survive the reference's
    kill + compaction are exactly the package's non-deleted particles (as a set: compress() reorders)"""
    inp = cases.outflow_inputs(dims, 71)
    a, b = cases.run_outflow_pkg(dims, *inp), cases.run_outflow_ref(dims, *inp)
    assert 0 < b["pos"].shape[1] < (inp[4] == 0).sum() and (b["flags"] != inp[0]).any()
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D, (17, 9, 11)])
@pytest.mark.parametrize("with_ptype", [False, True])
def test_apic_transfers(oracle_backend, dims, with_ptype):
    """apicMapPartsToMAC / apicMapMACGridToParts (plugin/apic.cpp) against the compiled reference, bit for bit"""
    flags, vel, pos, pflag, pvel, cp = cases.apic_inputs(dims, 61)
    ptype = None
    if with_ptype:
        ptype = (np.random.default_rng(18).integers(0, 4, pos.shape[1]) * 2).astype(np.int32)
    a = cases.run_apic_pkg(dims, flags, vel, pos, pflag, pvel, cp, ptype, 4 if with_ptype else 0)
    b = cases.run_apic_ref(dims, flags, vel, pos, pflag, pvel, cp, ptype, 4 if with_ptype else 0)
    assert np.abs(b["apic_vel"]).max() > 0.1 and np.abs(b["apic_cpx"]).max() > 0.01
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("deleteInObstacle,stopInObstacle", [(False, True), (True, True), (False, False), (True, False)])
def test_advect_in_grid(oracle_backend, dims, mode, deleteInObstacle, stopInObstacle):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 19, empty_top=True)
    vel = util.smooth_vel(sx, sy, sz, 20, 2.0)
    pos, pflag, _ = util.make_particles(flags, 2, 21)
    a = cases.run_advect_parts_pkg(dims, 0.8, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle)
    b = cases.run_advect_parts_ref(dims, 0.8, flags, vel, pos, pflag, mode, deleteInObstacle, stopInObstacle)
    assert_bitexact(a[1], b[1], "particle flags")
    assert_bitexact(a[0], b[0], "particle positions")


def test_advect_in_grid_ptype_skipnew(oracle_backend):
    dims = (12, 10, 9)
    flags = util.make_flags(*dims, 22, empty_top=True)
    vel = util.smooth_vel(*dims, 23, 1.5)
    pos, pflag, _ = util.make_particles(flags, 2, 24)
    ptype = (np.random.default_rng(25).integers(0, 2, pos.shape[1]) * 2).astype(np.int32)
    a = cases.run_advect_parts_pkg(dims, 1.0, flags, vel, pos, pflag, 2, False, True, True, ptype, 2)
    b = cases.run_advect_parts_ref(dims, 1.0, flags, vel, pos, pflag, 2, False, True, True, ptype, 2)
    assert_bitexact(a[1], b[1], "particle flags")
    assert_bitexact(a[0], b[0], "particle positions")


@pytest.mark.parametrize("dims", [(12, 10, 9), cases.SIZE_2D])
def test_glue(oracle_backend, dims):
    sx, sy, sz = dims
    flags = util.make_flags(sx, sy, sz, 26, empty_top=True)
    flags[flags.shape[0] // 2, 3, 3] |= util.STICK
    vel = util.rand_vel(sx, sy, sz, 27)
    density = util.rand_real((sz, sy, sx), 28)
    a = cases.run_glue_pkg(dims, 0.7, flags, vel, density)
    b = cases.run_glue_ref(dims, 0.7, flags, vel, density)
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), cases.SIZE_2D])
@pytest.mark.parametrize("with_phi", [False, True])
def test_flip_glue(oracle_backend, dims, with_phi):
    fl0, pos, pflag, pvel, vel = cases.flipglue_inputs(dims, 31)
    phi = util.rand_real((dims[2], dims[1], dims[0]), 35) if with_phi else None
    a = cases.run_flipglue_pkg(dims, fl0, pos, pflag, pvel, vel, phi)
    b = cases.run_flipglue_ref(dims, fl0, pos, pflag, pvel, vel, phi)
    assert (b["flags"] & 1).sum() > 50
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("dims", [(14, 12, 10), (9, 16, 11), cases.SIZE_2D])
def test_surface_pieces(oracle_backend, dims):
    """benchmark_dam.py's free-surface / particle maintenance calls (SURVEY 8f-2 leftovers, 8f-3)"""
    I = cases.surface_inputs(dims, 51)
    a = cases.run_surface_pkg(dims, I)
    b = cases.run_surface_ref(dims, I)
    assert b["gpi_sys"].size > 100 and (b["isolated"] != I["fiso"]).sum() > 0
    assert (b["push"] != I["pos"]).any() and (b["project"] != I["pos"]).any()
    for k in b:
        assert_bitexact(a[k], b[k], k)


@pytest.mark.parametrize("ext", [".uni", ".raw"])
def test_grid_files_interoperate_with_reference(oracle_backend, tmp_path, ext):
    """.uni / .raw files: what the package writes the reference reads back bit for bit, and vice versa"""
    from mantaflow_amd import core
    dims = (11, 7, 5)
    sx, sy, sz = dims
    s = cases._mk_solver(dims)
    rng = np.random.default_rng(3)
    for kind, cls, shape, dt in ((0, core.IntGrid, (sz, sy, sx), np.int32), (1, core.Grid, (sz, sy, sx), np.float32),
                                 (2, core.VecGrid, (sz, sy, sx, 3), np.float32), (3, core.MACGrid, (sz, sy, sx, 3), np.float32),
                                 (4, core.LevelsetGrid, (sz, sy, sx), np.float32)):
        a = (rng.normal(0, 3, shape) * (100 if kind == 0 else 1)).astype(dt)
        g = cls(s)
        g.from_numpy(a)
        f1 = str(tmp_path / ("ours%d%s" % (kind, ext)))
        assert g.save(f1) == 1
        back = np.zeros(shape, dt)
        util.refcall("ref_grid_file", sx, sy, sz, kind, 0, f1.encode(), back)
        assert_bitexact(back, a, "reference reads our %s kind %d" % (ext, kind))
        f2 = str(tmp_path / ("theirs%d%s" % (kind, ext)))
        util.refcall("ref_grid_file", sx, sy, sz, kind, 1, f2.encode(), a)
        h = cls(s)
        assert h.load(f2) == 1
        assert_bitexact(h.to_numpy(), a, "we read the reference's %s kind %d" % (ext, kind))
    if ext == ".uni":
        # wrong size / wrong type are refused like the reference does (iogrids.cpp:493-494)
        other = core.Grid(cases._mk_solver((10, 7, 5)))
        with pytest.raises(RuntimeError, match="grid dim doesn't match"):
            other.load(str(tmp_path / "ours1.uni"))
        with pytest.raises(RuntimeError, match="grid type doesn't match"):
            core.Grid(s).load(str(tmp_path / "ours2.uni"))


@pytest.mark.parametrize("orderSpace", [1, 2])
@pytest.mark.parametrize("case", range(len(cases.INTERP_CASES)))
def test_interpolate_between_grid_sizes(oracle_backend, case, orderSpace):
    """interpolateGrid / interpolateGridVec3 / interpolateMACGrid (waveletturbulence.cpp:37-78): up- and down-sampling,
    anisotropic scale, offset, explicit size, 2-D; linear and cubic (orderSpace 2, util/interpolHigh.h)"""
    sd, td, scale, offset, size = cases.INTERP_CASES[case]
    fields = {"real": util.rand_real((sd[2], sd[1], sd[0]), 71), "vec": util.rand_vel(*sd, 72)}
    a = cases.run_interp_pkg(sd, td, scale, offset, size, fields, orderSpace)
    b = cases.run_interp_ref(sd, td, scale, offset, size, fields, orderSpace)
    for k in b:
        assert_bitexact(a[k], b[k], "%s case %d orderSpace %d" % (k, case, orderSpace))


SHAPES = [  # kind, 9 floats (see ref_shape_levelset), constructor
    (0, (2.5, 3.0, 1.5, 9.25, 8.0, 6.5, 0, 0, 0), lambda sc, s: sc.Box(parent=s, p0=(2.5, 3.0, 1.5), p1=(9.25, 8.0, 6.5))),
    (1, (6.0, 5.5, 4.0, 3.25, 0, 0, 1.0, 1.5, 0.75), lambda sc, s: sc.Sphere(parent=s, center=(6.0, 5.5, 4.0), radius=3.25, scale=(1.0, 1.5, 0.75))),
    (2, (6.5, 2.0, 5.0, 2.75, 0, 0, 0, 1.5, 0), lambda sc, s: sc.Cylinder(parent=s, center=(6.5, 2.0, 5.0), radius=2.75, z=(0, 1.5, 0))),
    (2, (5.0, 5.0, 5.0, 2.0, 0, 0, 1.0, 2.0, 0.5), lambda sc, s: sc.Cylinder(parent=s, center=(5.0, 5.0, 5.0), radius=2.0, z=(1.0, 2.0, 0.5))),
]


@pytest.mark.parametrize("case", range(len(SHAPES)))
@pytest.mark.parametrize("dims", [(13, 11, 9), (16, 12, 1)])
def test_shape_levelsets(oracle_backend, case, dims):
    """Box / Sphere / Cylinder signed distance fields (shapes.cpp) -- the inputs of densityInflow"""
    from mantaflow_amd import core, scene
    kind, q, mk = SHAPES[case]
    s = cases._mk_solver(dims)
    to3 = lambda t: core.vec3(*t)
    class SC:   # constructors taking tuples
        Box = staticmethod(lambda parent, p0, p1: scene.Box(parent=parent, p0=to3(p0), p1=to3(p1)))
        Sphere = staticmethod(lambda parent, center, radius, scale: scene.Sphere(parent=parent, center=to3(center), radius=radius, scale=to3(scale)))
        Cylinder = staticmethod(lambda parent, center, radius, z: scene.Cylinder(parent=parent, center=to3(center), radius=radius, z=to3(z)))
    ours = mk(SC, s).computeLevelset().to_numpy()
    ref = np.zeros((dims[2], dims[1], dims[0]), np.float32)
    util.refcall("ref_shape_levelset", dims[0], dims[1], dims[2], kind, np.array(q, np.float32), ref)
    assert_bitexact(ours, ref, "levelset of shape %d" % case)


@pytest.mark.parametrize("case", range(len(SHAPES)))
@pytest.mark.parametrize("dims", [(13, 11, 9), (16, 12, 1)])
def test_shape_apply_to_grid(oracle_backend, case, dims):
    """Shape.applyToGrid on Real / Vec3 / MAC / int grids, with and without respectFlags: the reference's isInside tests at the
    cell centres and, for MAC grids, at the three face positions"""
    from mantaflow_amd import core, scene
    kind, q, mk = SHAPES[case]
    sx, sy, sz = dims
    s = cases._mk_solver(dims)
    to3 = lambda t: core.vec3(*t)
    class SC:
        Box = staticmethod(lambda parent, p0, p1: scene.Box(parent=parent, p0=to3(p0), p1=to3(p1)))
        Sphere = staticmethod(lambda parent, center, radius, scale: scene.Sphere(parent=parent, center=to3(center), radius=radius, scale=to3(scale)))
        Cylinder = staticmethod(lambda parent, center, radius, z: scene.Cylinder(parent=parent, center=to3(center), radius=radius, z=to3(z)))
    shape = mk(SC, s)
    flags = util.make_flags(sx, sy, sz, 88, obstacles=True)
    fl = cases.soa_to_grid(core.FlagGrid(s), flags)
    qa = np.array(q, np.float32)
    val = np.array([1.5, -2.25, 0.75], np.float32)
    hit = 0
    for gk, G, shp in ((0, core.Grid, (sz, sy, sx)), (1, core.VecGrid, (3, sz, sy, sx)), (2, core.MACGrid, (3, sz, sy, sx)), (3, core.IntGrid, (sz, sy, sx))):
        for respect in (None, fl):
            base = util.rand_real(shp, 89).astype(np.float32) if gk != 3 else np.full(shp, 7, np.int32)
            g = cases.soa_to_grid(G(s), base)
            value = float(val[0]) if gk == 0 else (3 if gk == 3 else core.vec3(*[float(v) for v in val]))
            shape.applyToGrid(grid=g, value=value, respectFlags=respect)
            ref = base.copy()
            v = np.array([3, 0, 0], np.float32) if gk == 3 else val
            util.refcall("ref_shape_apply", sx, sy, sz, kind, qa, gk, ref, v, None if respect is None else flags)
            assert_bitexact(cases.grid_to_soa(g), ref, "shape %d on grid kind %d" % (case, gk))
            hit += int((ref != base).sum())
    assert hit > 0 or sz == 1       # (the 3-D test shapes do not all reach the z = 0.5 plane of a 2-D grid)


def test_noise_tile_and_density_inflow(oracle_backend, oracle):
    """the wavelet noise tile (3 x 128^3, noisefield.cpp:95-186) and densityInflow (initplugins.cpp:27-43) with the source
    of scenes/simpleplume.py: bit-identical to the reference"""
    from mantaflow_amd import core, scene
    tile = np.zeros(3 * 128 ** 3, np.float32)
    oracle.lib.call("mf_noise_generate_tile", util.P(tile), 13322223, None)
    ref_tile = np.zeros_like(tile)
    util.refcall("ref_noise_tile", ref_tile)
    assert_bitexact(tile, ref_tile, "wavelet noise tile")
    dims = (24, 32, 20)
    sx, sy, sz = dims
    for t_total, sigma, seed in ((0.0, 0.5, -1), (3.5, 0.5, -1), (1.25, 0.0, 77), (2.0, 1.5, 5)):
        s = cases._mk_solver(dims)
        s.timeTotal = t_total
        flags = util.make_flags(sx, sy, sz, 81)
        fl = cases.soa_to_grid(core.FlagGrid(s), flags)
        d0 = (util.rand_real((sz, sy, sx), 82) * 0.3).astype(np.float32)
        dens = cases.soa_to_grid(core.Grid(s), d0)
        noise = scene.NoiseField(parent=s, fixedSeed=seed, loadFromFile=True)
        noise.posScale = core.vec3(45)
        noise.clamp, noise.clampNeg, noise.clampPos = True, 0, 1
        noise.valOffset, noise.timeAnim = 0.75, 0.2
        src = scene.Cylinder(parent=s, center=core.vec3(12, 4, 10), radius=4.5, z=core.vec3(0, 1.5, 0))
        scene.densityInflow(flags=fl, density=dens, noise=noise, shape=src, scale=1, sigma=sigma)
        ours = cases.grid_to_soa(dens)
        ref = d0.copy()
        q = np.array([12, 4, 10, 4.5, 0, 0, 0, 1.5, 0], np.float32)
        P = np.array([45, 45, 45, 0, 0, 0, 0.75, 1.0, 1, 0, 1, 0.2], np.float32)
        cf = ctypes.c_float
        util.refcall("ref_density_inflow", sx, sy, sz, cf(t_total), flags, ref, 2, q, seed, P, cf(1.0), cf(sigma))
        assert (ref != d0).sum() > 50
        assert_bitexact(ours, ref, "densityInflow t=%g sigma=%g" % (t_total, sigma))


@pytest.mark.parametrize("dims,small", [((20, 14, 12), (10, 7, 6)), ((24, 30, 1), (12, 15, 1)), ((17, 11, 9), (9, 6, 5))])
def test_wavelet_turbulence_pieces(oracle_backend, dims, small):
    """computeEnergy, computeWaveletCoeffs, vorticityConfinement, setOpenBound, applyNoiseVec3 (scenes/waveletTurbulence.py)"""
    inp = cases.turb_inputs(dims, small, 91)
    a = cases.run_turb_pkg(dims, *inp, small)
    b = cases.run_turb_ref(dims, *inp, small)
    assert np.abs(b["noise_plain"] - inp[1]).max() > 1e-3 and (b["open_Y"] != 2).sum() > 0
    for k in b:
        assert_bitexact(a[k], b[k], k)


def test_init_domain_matches_reference(oracle_backend):
    from mantaflow_amd import core
    for dims, bw, kw in [((10, 9, 8), 0, {}), ((12, 10, 9), 1, dict(open="xY", outflow="z")), ((16, 12, 1), 0, dict(inflow="y"))]:
        s = cases._mk_solver(dims)
        fl = core.FlagGrid(s)
        fl.initDomain(boundaryWidth=bw, **kw)
        fl.fillGrid()
        ref = np.zeros((dims[2], dims[1], dims[0]), np.int32)
        pad = lambda x: (x + "      ").encode()
        util.refcall("ref_init_domain", dims[0], dims[1], dims[2], ref, bw, b"xXyYzZ", pad(kw.get("open", "")),
                     pad(kw.get("inflow", "")), pad(kw.get("outflow", "")), 1)
        assert_bitexact(cases.grid_to_soa(fl), ref, "initDomain %s" % (dims,))


def test_reductions(oracle):
    n = 12345
    a = util.rand_real((n,), 30, 3.0)
    import ctypes
    r = ctypes.c_float()
    oracle.call("mf_grid_max_abs", n, oracle.dev(a), ctypes.byref(r), None)
    rr = np.zeros(1, np.float32)
    util.refcall("ref_grid_max_abs", n, a, rr)
    assert r.value == rr[0]
    d = ctypes.c_double()
    oracle.call("mf_grid_sum_sqr", n, oracle.dev(a), ctypes.byref(d), None)
    dd = np.zeros(1, np.float64)
    util.refcall("ref_grid_sum_sqr", n, a, dd)
    assert abs(d.value - dd[0]) <= 1e-12 * dd[0]


def test_sample_flags_with_particles_matches_reference_stream(oracle_backend):
    """MT19937 seed 9832 + the reference's fp32 arithmetic reproduced host-side (flip.cpp:33-58)"""
    from mantaflow_amd import core, scene
    for dims, disc in [((10, 9, 8), 2), ((16, 12, 1), 3)]:
        s = cases._mk_solver(dims)
        flags = util.make_flags(*dims, 61, empty_top=True)
        fl = cases.soa_to_grid(core.FlagGrid(s), flags)
        pp = core.BasicParticleSystem(s)
        scene.sampleFlagsWithParticles(fl, pp, disc, 0.2)
        cap = 200000
        pos = np.zeros((3, cap), np.float32)
        import ctypes
        cnt = ctypes.c_int64(0)
        util.refcall("ref_sample_flags_with_particles", dims[0], dims[1], dims[2], flags, disc, ctypes.c_float(0.2), cap, pos, ctypes.byref(cnt))
        assert cnt.value == pp.np and pp.np > 100
        assert_bitexact(pp.get_positions().T, pos[:, :cnt.value], "sampled positions")
