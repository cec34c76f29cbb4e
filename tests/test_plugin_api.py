"""The Python-side contract of the plugin boundary: universal kwargs, unknown-argument errors, the reference's bool / int
argument conversion rules (pwrapper/pconvert.cpp:170-181, 212-215, 460-477), NULL pointers, introspection attributes."""
import numpy as np
import pytest

import cases
import util


def _scene(dims=(12, 10, 9)):
    from mantaflow_amd import core
    s = cases._mk_solver(dims, 0.5)
    fl = core.FlagGrid(s)
    fl.initDomain()
    fl.fillGrid()
    return s, fl, core.MACGrid(s), core.Grid(s)


def test_universal_kwargs_and_unknown_arguments(oracle_backend):
    from mantaflow_amd import api as m
    s, fl, vel, dens = _scene()
    m.advectSemiLagrange(flags=fl, vel=vel, grid=dens, order=1, notiming=True, nocheck=True, name="x")   # all accepted
    with pytest.raises(RuntimeError, match="unknown"):
        m.advectSemiLagrange(flags=fl, vel=vel, grid=dens, ordr=2)
    with pytest.raises(RuntimeError, match="can't convert argument to MACGrid"):
        m.advectSemiLagrange(flags=fl, vel=dens, grid=dens)


def test_int_and_bool_conversion_rules(oracle_backend):
    from mantaflow_amd import api as m
    s, fl, vel, dens = _scene()
    m.advectSemiLagrange(flags=fl, vel=vel, grid=dens, order=2.0)            # a float within 1e-5 of an integer is an int
    m.advectSemiLagrange(fl, vel, dens, 1.000001)
    with pytest.raises(RuntimeError, match="argument is not an int"):
        m.advectSemiLagrange(flags=fl, vel=vel, grid=dens, order=1.5)
    with pytest.raises(RuntimeError, match="argument is not an int"):
        m.advectSemiLagrange(flags=fl, vel=vel, grid=dens, order="2")
    pres = m.Grid(s)
    m.solvePressure(flags=fl, vel=vel, pressure=pres, precondition=True, useL2Norm=False)
    with pytest.raises(RuntimeError, match="argument is not a boolean"):
        m.solvePressure(flags=fl, vel=vel, pressure=pres, precondition=1)
    with pytest.raises(RuntimeError, match="argument is not a boolean"):
        m.addGravity(fl, vel, m.vec3(0, -1, 0), None, 1)
    m.addGravity(fl, vel, (0, -1e-3, 0))                                       # Vec3 from a 3-tuple (pconvert.cpp:216-226)
    m.solvePressure(flags=fl, vel=vel, pressure=pres, phi=0, fractions=None)   # None or int 0 is NULL (pclass.cpp:128-134)


def test_introspection_attributes(oracle_backend):
    """_class / _cname / _T as the reference's test helper reads them (registry.cpp:123-133, helperInclude.py:97-126)"""
    from mantaflow_amd import api as m
    s, fl, vel, dens = _scene()
    assert (s._class, type(s).__name__) == ("FluidSolver", "Solver") or s._class == "FluidSolver"
    assert (dens._class, dens._T, dens._cname) == ("Grid", "Real", "Grid<Real>")
    assert (m.VecGrid(s)._class, m.VecGrid(s)._T) == ("Grid", "Vec3")
    assert (m.IntGrid(s)._class, m.IntGrid(s)._T) == ("Grid", "int")
    assert type(vel).__name__ == "MACGrid" and type(m.LevelsetGrid(s)).__name__ == "LevelsetGrid"
    pp = s.create(m.BasicParticleSystem)
    assert pp._class == "BasicParticleSystem"
    assert (pp.create(m.PdataVec3)._class, pp.create(m.PdataVec3)._T) == ("ParticleDataImpl", "Vec3")
    assert m.Vec3 is m.vec3 and m.Real is float and m.Vec3Grid is m.VecGrid


def test_vec4_value_type():
    """manta.vec4 (pvec3.cpp:280-389): members x, y, z, t; zero / broadcast / four-value construction; partial init is an error;
    repr in the reference's format; no arithmetic"""
    from manta import vec4, Vec4
    assert Vec4 is vec4
    v = vec4()
    assert (v.x, v.y, v.z, v.t) == (0.0, 0.0, 0.0, 0.0)
    v = vec4(1.5)
    assert tuple(v) == (1.5, 1.5, 1.5, 1.5)
    v = vec4(1, 2, 3, 0.1)
    assert v.t == float(np.float32(0.1)) and repr(v) == "[+1.000000,+2.000000,+3.000000,+0.100000]"
    with pytest.raises(RuntimeError, match="Invalid partial init of vec4"):
        vec4(1, 2)
    with pytest.raises(TypeError):
        v + v
