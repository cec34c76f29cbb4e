"""-m gpu: BASELINE.json's configs 1-3 at their STATED sizes on the HIP library against the CPU checker (the plain-C oracle, which
the CPU suite pins bit for bit to the compiled reference; and the compiled reference itself where its .so travelled):
  config 1  scenes/simpleplume.py 64 x 96 x 64, 20 steps
  config 2  one 256^3 smoke step (bench.py's synthetic input): advect density + velocity (MacCormack), setWallBcs, MIC-CG 1e-3
  config 3  one 128^3 S-flip step: advectInGrid RK4, mapPartsToMAC, solvePressure, flipVelocityUpdate, 3.8 M particles
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
