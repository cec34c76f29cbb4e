"""z-slab multi-rank path on CPU: world_size 2 and 3 over gloo (oracle library) against the single-rank run.
Advection must be bit-identical on every owned plane; the pressure solve uses slab-local MIC, so it is validated at
converged-solution level (same stopping rule, divergence removed, pressure close to the single-rank solution)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "slab_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(tmp_path, world, backend, dims="20x16x36"):
    out = str(tmp_path / ("w%d" % world))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    if world == 1:
        cmd = [sys.executable, WORKER, out, backend, dims]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), WORKER, out, backend, dims]
    subprocess.run(cmd, check=True, env=env, timeout=600)
    parts = [dict(np.load(out + ".%d.npz" % r)) for r in range(world)]
    cat = lambda k, ax: np.concatenate([p[k] for p in parts], axis=ax)
    return dict(dens=cat("dens", 0), vel_adv=cat("vel_adv", 1), vel=cat("vel", 1), pres=cat("pres", 0), div=cat("div", 0),
                iters=[int(p["iters"]) for p in parts], res=[float(p["res"]) for p in parts], plain_dens=[p["plain_dens"] for p in parts],
                mic_blocking=[tuple(int(v) for v in p["mic_blocking"]) for p in parts])


def check_against_single(single, multi):
    util.assert_bitexact(multi["dens"], single["dens"], "advected density")
    util.assert_bitexact(multi["vel_adv"], single["vel_adv"], "advected velocity")
    # a plain solver living next to the slab solver (window lo > 0 on the upper ranks) computes the single-device result
    for r, pd in enumerate(multi["plain_dens"]):
        util.assert_bitexact(pd, single["plain_dens"][0], "plain solver beside the slab solver of rank %d" % r)
    assert len(set(multi["iters"])) == 1, "every rank must take identical CG branches"
    assert max(multi["res"]) < 1e-4
    assert np.abs(multi["div"]).max() < 2e-3
    scale = np.abs(single["pres"]).max()
    assert np.abs(multi["pres"] - single["pres"]).max() < 5e-3 * scale
    assert np.abs(multi["vel"] - single["vel"]).max() < 5e-3 * max(np.abs(single["vel"]).max(), 1)


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    return run_world(tmp_path_factory.mktemp("slab1"), 1, "oracle")


def test_single_rank_slab_equals_plugin_path(single, oracle_backend):
    """world 1: the slab driver is the reference algorithm -> bit-identical advection, same CG iterates"""
    import cases
    from mantaflow_amd import core, plugins
    dims = (20, 16, 36)
    s = cases._mk_solver(dims, 0.9)
    flags_g = util.make_flags(*dims, 51, obstacles=True, empty_top=True)
    vel_g = util.smooth_vel(*dims, 52, 2.0); vel_g[2] *= 0.95
    fl, v, d, p = core.FlagGrid(s), core.MACGrid(s), core.Grid(s), core.Grid(s)
    cases.soa_to_grid(fl, flags_g); cases.soa_to_grid(v, vel_g); cases.soa_to_grid(d, util.rand_real((36, 16, 20), 53))
    plugins.advectSemiLagrange(fl, v, d, order=2)
    plugins.advectSemiLagrange(fl, v, v, order=2)
    plugins.setWallBcs(fl, v)
    util.assert_bitexact(cases.grid_to_soa(d), single["dens"], "density")
    util.assert_bitexact(cases.grid_to_soa(v), single["vel_adv"], "velocity")
    plugins.solvePressure(v, p, fl, cgAccuracy=1e-4)
    assert plugins.lastCgStats()["iterations"] == single["iters"][0]
    assert util.rel_err(cases.grid_to_soa(p), single["pres"]) < 1e-5


@pytest.mark.parametrize("world", [2, 3])
def test_slab_world(tmp_path, single, world):
    multi = run_world(tmp_path, world, "oracle")
    check_against_single(single, multi)


def test_slab_world_with_blocked_preconditioner(tmp_path):
    """A domain wide enough (256 x 128) that BOTH cuts of the P > 1 preconditioner are active in the slab solver (blocks of 64 rows along
    y and 128 cells along x, on top of the z-cut), together with ghost planes and the packed ApplyMatrix: 2 ranks against the undivided
    run, at converged-solution level; the iteration counts are logged."""
    dims = "256x128x20"
    single = run_world(tmp_path, 1, "oracle", dims)
    multi = run_world(tmp_path, 2, "oracle", dims)
    assert single["mic_blocking"] == [(0, 0)]
    assert multi["mic_blocking"] == [(64, 128), (64, 128)]
    check_against_single(single, multi)
    print("CG iterations 256x128x20: undivided %d, 2 slabs with 64 x 128 blocks %d" % (single["iters"][0], multi["iters"][0]))
    assert multi["iters"][0] < 2.5 * single["iters"][0]


# ---- FLIP on slabs (P2G with reverse halo, G2P / advectInGrid with particle migration) ------------------------
def _run_case(tmp_path, world, backend, dims, case):
    out = str(tmp_path / ("%s%d%s" % (case, world, backend)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    if world == 1:
        cmd = [sys.executable, WORKER, out, backend, dims, case]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), WORKER, out, backend, dims, case]
    subprocess.run(cmd, check=True, env=env, timeout=600)
    return [dict(np.load(out + ".%d.npz" % r)) for r in range(world)]


def run_flip_world(tmp_path, world, backend="oracle", dims="20x16x36"):
    parts = _run_case(tmp_path, world, backend, dims, "flip")
    o = np.argsort(np.concatenate([p["adv_pid"] for p in parts]), kind="stable")
    o2 = np.argsort(np.concatenate([p["pid"] for p in parts]), kind="stable")
    return dict(adv_pos=np.concatenate([p["adv_pos"] for p in parts], axis=1)[:, o], adv_flag=np.concatenate([p["adv_flag"] for p in parts])[o],
                adv_pid=np.concatenate([p["adv_pid"] for p in parts])[o], pvel=np.concatenate([p["pvel"] for p in parts], axis=1)[:, o2],
                p2g_vel=np.concatenate([p["p2g_vel"] for p in parts], axis=1), p2g_w=np.concatenate([p["p2g_w"] for p in parts], axis=1),
                moved=sum(int(p["moved"]) for p in parts), n0=[int(p["n0"]) for p in parts], iters=[int(p["iters"]) for p in parts])


@pytest.fixture(scope="module")
def flip_single(tmp_path_factory):
    return run_flip_world(tmp_path_factory.mktemp("flip1"), 1)


def test_flip_single_rank_slab_equals_plugin_path(flip_single, oracle_backend):
    """world 1: the slab FLIP operators are the plugin operators (bit-identical)"""
    import cases
    from mantaflow_amd import core, plugins
    dims = (20, 16, 36)
    s = cases._mk_solver(dims, 0.8)
    flags_g = util.make_flags(*dims, 61, obstacles=True, empty_top=True)
    vel_g = util.smooth_vel(*dims, 62, 2.2)
    pos, pflag, pvel = util.make_particles(flags_g, 2, 63, include_border=False)
    fl, v, vo, w = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.VecGrid(s)
    cases.soa_to_grid(fl, flags_g); cases.soa_to_grid(v, vel_g)
    pp = cases._mk_parts(s, pos, pflag)
    pv = cases._pd_vec3(s, pp, pvel)
    pp.advectInGrid(fl, v, 2, deleteInObstacle=False)
    P, F = cases._parts_get(pp)
    util.assert_bitexact(flip_single["adv_pos"], P, "advected positions")
    assert (flip_single["adv_flag"] == F).all()
    plugins.mapPartsToMAC(fl, v, vo, pp, pv, w)
    util.assert_bitexact(flip_single["p2g_vel"], cases.grid_to_soa(v), "P2G velocity")
    util.assert_bitexact(flip_single["p2g_w"], cases.grid_to_soa(w), "P2G weight")


def check_flip_against_single(flip_single, m):
    assert (m["adv_pid"] == flip_single["adv_pid"]).all() and len(m["adv_pid"]) == len(flip_single["adv_pid"])
    assert m["moved"] > 0 and min(m["n0"]) > 0, "the case must exercise migration"
    # a particle's RK4 step only reads the velocity field: bit-identical wherever it lives
    util.assert_bitexact(m["adv_pos"], flip_single["adv_pos"], "advected positions after migration")
    assert (m["adv_flag"] == flip_single["adv_flag"]).all()
    # P2G: same contributions per node; nodes fed from two ranks (and nodes whose particles changed order by migration) are
    # summed in another order -> 1e-5 relative (north_star tolerance for fp32 grid fields)
    for k in ("p2g_vel", "p2g_w"):
        assert util.rel_err(m[k], flip_single[k]) <= 1e-5, k
    same = (util.bits(m["p2g_w"]) == util.bits(flip_single["p2g_w"])).mean()
    assert same > 0.9, "most nodes see their particles in the same order: %.3f" % same
    # after the (block-Jacobi preconditioned) solve: converged-solution level, as for the smoke path
    assert len(set(m["iters"])) == 1
    scale = max(np.abs(flip_single["pvel"]).max(), 1.0)
    assert np.abs(m["pvel"] - flip_single["pvel"]).max() < 5e-3 * scale


@pytest.mark.parametrize("world", [2, 3])
def test_flip_slab_world(tmp_path, flip_single, world):
    check_flip_against_single(flip_single, run_flip_world(tmp_path, world))


# ---- the whole flip01_simple.py loop on slabs ---------------------------------------------------------------------------
def run_liquid_world(tmp_path, world, backend="oracle", dims="20x16x36"):
    parts = _run_case(tmp_path, world, backend, dims, "liquid")
    o = np.argsort(np.concatenate([p["pid"] for p in parts]), kind="stable")
    cat = lambda k, ax: np.concatenate([p[k] for p in parts], axis=ax)
    return dict(pid=cat("pid", 0)[o], pos=cat("pos", 1)[:, o], pvel=cat("pvel", 1)[:, o], flags=cat("flags", 0), flags0=cat("flags0", 0),
                vel=cat("vel", 1), vel_ext0=cat("vel_ext0", 1), moved=sum(int(p["moved"]) for p in parts), iters=[list(p["iters"]) for p in parts])


def check_liquid_against_single(one, m):
    assert (m["pid"] == one["pid"]).all() and m["moved"] > 0
    # step 1 up to the solve: flags from particles are bit-exact, extrapolated P2G velocity within the P2G tolerance
    assert (m["flags0"] == one["flags0"]).all()
    assert util.rel_err(m["vel_ext0"], one["vel_ext0"]) <= 1e-5
    # after two steps (two block-Jacobi-preconditioned solves to 1e-6): converged-solution level
    assert all(it == m["iters"][0] for it in m["iters"])
    assert (m["flags"] != one["flags"]).mean() < 2e-3
    scale = max(np.abs(one["pvel"]).max(), 1.0)
    assert np.abs(m["pvel"] - one["pvel"]).max() < 5e-3 * scale
    assert np.abs(m["pos"] - one["pos"]).max() < 5e-3
    assert np.abs(m["vel"] - one["vel"]).max() < 5e-3 * max(np.abs(one["vel"]).max(), 1.0)


@pytest.fixture(scope="module")
def liquid_single(tmp_path_factory):
    return run_liquid_world(tmp_path_factory.mktemp("liq1"), 1)


def test_liquid_single_rank_slab_equals_plugin_path(liquid_single, oracle_backend):
    """world 1: two steps of the flip01 loop through the slab operators = the same loop through the plugins, bit for bit"""
    import cases
    from mantaflow_amd import core, plugins
    dims = (20, 16, 36)
    NX, NY, NZ = dims
    s = cases._mk_solver(dims, 0.8)
    flags_g = np.full((NZ, NY, NX), 4, np.int32)
    flags_g[:, :, 0] = flags_g[:, :, -1] = flags_g[:, 0, :] = flags_g[:, -1, :] = 2
    flags_g[0] = flags_g[-1] = 2
    fluid = np.zeros_like(flags_g, bool)
    fluid[1:NZ - 1, 1:int(0.6 * NY), 1:int(0.5 * NX)] = True
    flags_g[fluid] = 1
    pos, pflag, pvel = util.make_particles(flags_g, 2, 73, vel_scale=0.6, deleted_frac=0.0, include_border=False)
    fl, v, vo, w, pr = core.FlagGrid(s), core.MACGrid(s), core.MACGrid(s), core.VecGrid(s), core.Grid(s)
    cases.soa_to_grid(fl, flags_g)
    pp = cases._mk_parts(s, pos, pflag)
    pv = cases._pd_vec3(s, pp, pvel)
    for step in range(2):
        pp.advectInGrid(fl, v, 2, deleteInObstacle=False)
        plugins.mapPartsToMAC(fl, v, vo, pp, pv, w)
        plugins.extrapolateMACFromWeight(v, w, distance=2)
        plugins.markFluidCells(pp, fl)
        plugins.addGravity(fl, v, core.vec3(0, -0.01, 0))
        plugins.setWallBcs(fl, v)
        plugins.solvePressure(v, pr, fl, cgAccuracy=1e-6)
        plugins.setWallBcs(fl, v)
        plugins.extrapolateMACSimple(fl, v, distance=4)
        plugins.flipVelocityUpdate(fl, v, vo, pp, pv, 0.97)
        s.step()
    P, F = cases._parts_get(pp)
    util.assert_bitexact(liquid_single["pos"], P, "positions after two steps")
    assert (liquid_single["flags"] == cases.grid_to_soa(fl)).all()
    assert util.rel_err(liquid_single["vel"], cases.grid_to_soa(v)) <= 1e-5
    assert util.rel_err(liquid_single["pvel"], cases._pd_get(pv, pp.np)) <= 1e-5


@pytest.mark.parametrize("world", [2, 3])
def test_liquid_slab_world(tmp_path, liquid_single, world):
    check_liquid_against_single(liquid_single, run_liquid_world(tmp_path, world))


# ---- the whole benchmark_dam.py loop on slabs (BASELINE config 4: ghost-fluid FLIP dam break) ---------------------------
DAM_RES = 16


def run_dam_world(tmp_path, world, backend="oracle", res=DAM_RES):
    parts = _run_case(tmp_path, world, backend, "%dx0x0" % res, "dam")
    o = np.argsort(np.concatenate([p["pid"] for p in parts]), kind="stable")
    cat = lambda k, ax: np.concatenate([p[k] for p in parts], axis=ax)
    return dict(pid=cat("pid", 0)[o], pos=cat("pos", 1)[:, o], pvel=cat("pvel", 1)[:, o], ptype=cat("ptype", 0)[o], flags=cat("flags", 0),
                vel=cat("vel", 1), pres=cat("pres", 0), phi=cat("phi", 0), phi_ls0=cat("phi_ls0", 0), phi_ls1=cat("phi_ls1", 0),
                moved=sum(int(p["moved"]) for p in parts), n0=[int(p["n0"]) for p in parts], iters=[list(p["iters"]) for p in parts],
                dts=[list(p["dts"]) for p in parts])


def check_dam_against_single(one, m):
    assert (m["pid"] == one["pid"]).all() and m["moved"] > 0 and sum(n > 0 for n in m["n0"]) >= 2, "the case must cross a slab face"
    # every rank takes the same branches: identical CG iteration counts and time steps (domain-wide max |v|)
    assert all(it == m["iters"][0] for it in m["iters"]) and all(dt == m["dts"][0] for dt in m["dts"])
    assert m["dts"][0] == one["dts"][0]
    # step 1, before the first solve feeds back: the particle level set (one-cell particle halo, a minimum: order independent)
    # and its extrapolation on the ghosts are bit-identical to the undivided domain
    util.assert_bitexact(m["phi_ls0"], one["phi_ls0"], "union level set + extrapolateLsSimple, step 1")
    # after the steps (block-Jacobi-preconditioned ghost-fluid solves to 1e-6): converged-solution level; index work exact
    assert (m["flags"] != one["flags"]).mean() < 2e-3 and (m["ptype"] != one["ptype"]).mean() < 2e-3
    assert np.abs(m["phi_ls1"] - one["phi_ls1"]).max() < 1e-3
    assert np.abs(m["pos"] - one["pos"]).max() < 5e-3
    for k in ("pvel", "vel", "pres"):
        assert np.abs(m[k] - one[k]).max() < 5e-3 * max(np.abs(one[k]).max(), 1.0), k


@pytest.fixture(scope="module")
def dam_single(tmp_path_factory):
    return run_dam_world(tmp_path_factory.mktemp("dam1"), 1)


def test_dam_single_rank_slab_equals_plugin_path(dam_single, oracle_backend):
    """world 1: the benchmark_dam loop through the slab operators (a slab solver next to a plain set-up solver in one process) =
    the same loop through the plugins: index work bit for bit, fields within the fp32 tolerance (the slab PCG combines its fp64
    partial sums per rank)"""
    import cases
    ref = cases.run_dam_pkg(DAM_RES, 3, zflow=True, cgacc=1e-6)
    assert dam_single["iters"][0] == ref["iters"]
    assert (dam_single["flags"] == ref["flags"]).all() and (dam_single["ptype"] == ref["ptype"]).all()
    util.assert_bitexact(dam_single["phi_ls1"], ref["rec"]["phi_ls"], "level set of the last step")
    for k in ("pos", "pvel", "vel", "pres"):
        assert util.rel_err(dam_single[k], ref[k]) <= 1e-5, k


@pytest.mark.parametrize("world", [2, 3])
def test_dam_slab_world(tmp_path, dam_single, world):
    check_dam_against_single(dam_single, run_dam_world(tmp_path, world))


# ---- the up-res loop of waveletTurbulence.py on slabs (BASELINE config 5: two solvers on the same z-ranges) -----------------
WLT_GS = (16, 24, 24)
_WLT_AXIS = dict(dens=0, vel=1, energy=0, xl_dens=0, xl_vel=1, energy0=0, xl_vel0=1, xl_dens0=0, vel_pre0=1, dens_pre0=0)


def run_wavelet_world(tmp_path, world, backend="oracle", gs=WLT_GS):
    parts = _run_case(tmp_path, world, backend, "%dx%dx%d" % gs, "wavelet")
    out = {k: np.concatenate([p[k] for p in parts], axis=a) for k, a in _WLT_AXIS.items()}
    out["iters"] = [list(p["iters"]) for p in parts]
    return out


def check_wavelet_against_single(one, m):
    assert all(it == m["iters"][0] for it in m["iters"])
    # no cross-rank sum before the first solve: advection with the split outflow BC, inflow noise, buoyancy, vorticity confinement,
    # and -- on synthetic input -- energy, the wavelet decomposition over a gathered z-window, resampling between the two slabs
    # (each under its own window), three noise octaves and the fine MacCormack advection are bit-identical to the undivided run
    for k in ("vel_pre0", "dens_pre0", "energy0", "xl_vel0", "xl_dens0"):
        util.assert_bitexact(m[k], one[k], k)
    # after two steps (block-Jacobi-preconditioned solves to 1e-6): converged-solution level; MacCormack's clamp is a selection,
    # so a few cells may pick the other candidate
    for k in ("dens", "vel", "energy", "xl_dens", "xl_vel"):
        d = np.abs(m[k] - one[k])
        scale = max(np.abs(one[k]).max(), 1e-3)
        assert (d > 1e-4 * scale).mean() < 1e-3 and d.max() < 5e-2 * scale, (k, d.max(), scale)


@pytest.fixture(scope="module")
def wavelet_single(tmp_path_factory):
    return run_wavelet_world(tmp_path_factory.mktemp("wlt1"), 1)


def test_wavelet_single_rank_slab_equals_plugin_path(wavelet_single, oracle_backend):
    """world 1: the loop through the slab operators = the loop through the plugins on two plain solvers, bit for bit"""
    import cases
    ref = cases.run_wavelet_pkg(WLT_GS, 2)
    assert wavelet_single["iters"][0] == ref["iters"]
    for k in _WLT_AXIS:
        util.assert_bitexact(wavelet_single[k], ref[k], k)


@pytest.mark.parametrize("world", [2, 3])
def test_wavelet_slab_world(tmp_path, wavelet_single, world):
    check_wavelet_against_single(wavelet_single, run_wavelet_world(tmp_path, world))
