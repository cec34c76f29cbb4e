"""-m gpu: BASELINE.json's configs 1-5 at their STATED sizes on the HIP library against the CPU checker (the plain-C oracle, which
the CPU suite pins bit for bit to the compiled reference; and the compiled reference itself where its .so travelled):
  config 1  scenes/simpleplume.py 64 x 96 x 64, 20 steps
  config 2  one 256^3 smoke step (bench.py's synthetic input): advect density + velocity (MacCormack), setWallBcs, MIC-CG 1e-3
  config 3  one 128^3 S-flip step: advectInGrid RK4, mapPartsToMAC, solvePressure, flipVelocityUpdate, 3.8 M particles
  config 4  one step of scenes/benchmark_dam.py's ghost-fluid FLIP loop on a 256^3-cell grid (res 116: 379 x 356 x 124, 8.3 M particles)
  config 5  the fine-grid pass of scenes/waveletTurbulence.py at 512^3 (interpolate -> 3 noise octaves -> 2 MacCormack advections)
The oracle runs once per test on the host cores (tens of seconds to ~2 minutes)."""
import os
import sys

import numpy as np
import pytest

import cases
import util
from util import assert_bitexact

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _on_oracle(fn, *a, **kw):
    from mantaflow_amd import _lib
    _lib.use_library(util.build_oracle(), "cpu")
    try:
        return fn(*a, **kw)
    finally:
        _lib.reset()


def _close(a, b, what, tol=1e-5):
    e = util.rel_err(a, b)
    assert e <= tol, "%s: relative error %.3e > %.0e" % (what, e, tol)


def test_config1_simpleplume_full_size(hip_backend):
    """64 x 96 x 64, 20 steps of the simpleplume.py loop: identical CG iteration counts in every step, density and velocity within
    1e-5 of the oracle's and of the compiled reference's own run of the same scene (ref_simpleplume)"""
    res, steps = 64, 20
    a = cases.run_simpleplume_pkg(res, steps)
    b = _on_oracle(cases.run_simpleplume_pkg, res, steps)
    assert a["iters"] == b["iters"] and min(b["iters"][1:]) > 3, (a["iters"], b["iters"])
    assert b["density"].max() > 0.5 and np.abs(b["vel"]).max() > 1e-2
    _close(a["density"], b["density"], "density vs oracle")
    _close(a["vel"], b["vel"], "velocity vs oracle")
    if util.have_ref():
        r = cases.run_simpleplume_ref(res, steps)
        assert_bitexact(b["density"], r["density"], "oracle vs compiled reference, density")
        _close(a["density"], r["density"], "density vs compiled reference")
        _close(a["vel"], r["vel"], "velocity vs compiled reference")


def test_config2_256_smoke_step_vs_oracle(hip_backend):
    """the bench step at 256^3: advected density and velocity bit-exact, the same number of CG iterations, pressure and projected
    velocity within 1e-5"""
    sys.path.insert(0, ROOT)
    import bench
    n = 256
    flags = bench.domain_flags(n, n, n)
    vel = bench.synthetic_velocity(n, n, n)
    dens = bench.synthetic_density(n, n, n)
    a = cases.run_smoke_step_pkg((n, n, n), 1.0, flags, vel, dens)
    b = _on_oracle(cases.run_smoke_step_pkg, (n, n, n), 1.0, flags, vel, dens)
    assert_bitexact(a["dens"], b["dens"], "advected density 256^3")
    assert_bitexact(a["vel_adv"], b["vel_adv"], "advected velocity 256^3")
    assert a["iters"] == b["iters"] and b["iters"] > 50, (a["iters"], b["iters"])
    _close(a["pressure"], b["pressure"], "pressure 256^3")
    _close(a["vel"], b["vel"], "projected velocity 256^3")


def test_config3_128_flip_step_vs_oracle(hip_backend):
    """S-flip at 128^3 with 8 particles per cell in a 0.4 x 0.6 x 1.0 liquid block: positions, flags and the P2G grids bit-exact,
    identical CG iteration count, particle velocities within 1e-5"""
    n = 128
    flags = np.full((n, n, n), 4, np.int32)
    flags[:, :, 0] = flags[:, :, -1] = flags[:, 0, :] = flags[:, -1, :] = 2
    flags[0] = flags[-1] = 2
    fluid = np.zeros_like(flags, bool)
    fluid[1:-1, 1:int(0.6 * n), 1:int(0.4 * n)] = True
    flags[fluid] = 1
    pos, pflag, pvel = util.make_particles(flags, 8, 9832, vel_scale=0.5, deleted_frac=0.0, include_border=False)
    vel = util.smooth_vel(n, n, n, 46, 1.5)
    assert pos.shape[1] > 3.5e6
    a = cases.run_flip_step_pkg((n, n, n), 0.5, flags, vel, pos, pflag, pvel)
    b = _on_oracle(cases.run_flip_step_pkg, (n, n, n), 0.5, flags, vel, pos, pflag, pvel)
    assert_bitexact(a["pos"], b["pos"], "advected positions")
    assert (a["pflag"] == b["pflag"]).all()
    assert_bitexact(a["p2g_vel"], b["p2g_vel"], "P2G velocity")
    assert_bitexact(a["p2g_w"], b["p2g_w"], "P2G weight")
    assert a["iters"] == b["iters"] and b["iters"] > 5, (a["iters"], b["iters"])
    _close(a["vel"], b["vel"], "projected velocity")
    _close(a["pvel"], b["pvel"], "particle velocities after the FLIP update")


def test_config4_dam_step_256_class_vs_oracle(hip_backend):
    """benchmark_dam.py's loop (bench.py's config4 workload: reference geometry at res 116 = 379 x 356 x 124 = 16.7 M cells, 8.3 M
    particles, ghost-fluid solve): two steps.  Step 1 starts from identical particles, so its particle level set (union + extrapolation)
    and P2G grids have to be bit-exact; CG iteration counts identical in both steps; after the two steps flags and particle types
    identical, positions / velocities / pressure / level set within 1e-5."""
    sys.path.insert(0, ROOT)
    import bench
    from mantaflow_amd import core, plugins, scene

    def run():
        sc = bench.dam_scene(core, plugins, scene, bench.DAM_RES)
        rec = {}

        def hook():
            if "phi1" not in rec:
                rec["phi1"] = cases.grid_to_soa(sc["phi"]).copy()
                rec["vel1"] = cases.grid_to_soa(sc["vel"]).copy()
        sc["state"]["hook"] = hook
        for _ in range(2):
            sc["step"]()
        sc["s"].sync()
        pp = sc["parts"]
        rec.update(pos=cases._ppos(pp), pvel=np.ascontiguousarray(sc["pvel"].to_numpy().T), ptype=sc["ptype"].data[:pp.np].cpu().numpy().copy(),
                   flags=cases.grid_to_soa(sc["flags"]), phi=cases.grid_to_soa(sc["phi"]), vel=cases.grid_to_soa(sc["vel"]),
                   pres=cases.grid_to_soa(sc["pressure"]), iters=list(sc["state"]["iters"]), gs=sc["gs"])
        return rec

    a = run()
    b = _on_oracle(run)
    assert a["gs"] == [379, 356, 124] and b["pos"].shape[1] > 8e6
    assert a["iters"] == b["iters"] and min(b["iters"]) > 10, (a["iters"], b["iters"])
    assert_bitexact(a["phi1"], b["phi1"], "particle level set of step 1")
    assert_bitexact(a["vel1"], b["vel1"], "P2G velocity + gravity of step 1")
    assert_bitexact(a["flags"], b["flags"], "flags after two steps")
    assert_bitexact(a["ptype"], b["ptype"], "particle types after two steps")
    for k in ("pos", "pvel", "vel", "pres", "phi"):
        _close(a[k], b[k], k)


def test_config5_fine_grid_pass_512_vs_oracle(hip_backend):
    """the fine-grid pass of waveletTurbulence.py:128-140 with a 256^3 coarse and a 512^3 fine grid (bench.py's config5 sizes):
    energy + wavelet decomposition, interpolateGrid / interpolateMACGrid, three octaves of applyNoiseVec3, two MacCormack advections
    of the fine density -- every field bit-exact against the oracle"""
    sys.path.insert(0, ROOT)
    import bench
    nc = 256
    v = bench.synthetic_velocity(nc, nc, nc, vmax=1.0)
    a = cases.run_upres_pass_pkg(nc, v)
    b = _on_oracle(cases.run_upres_pass_pkg, nc, v)
    assert np.abs(b["xl_vel"]).max() > 1.0 and b["xl_weight"].max() > 0.05
    for k in ("energy", "xl_weight", "xl_vel", "xl_dens"):
        assert_bitexact(a[k], b[k], k + " (512^3 fine grid)")
