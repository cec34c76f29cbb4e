"""Everything a scene gets from `from manta import *` (reference: module "manta", pwrapper/registry.cpp:390-421,
495-608, plus the constants of source/python/defines.py)."""
import sys

from .core import (BasicParticleSystem, FlagGrid, FluidSolver, Grid, IntGrid, LevelsetGrid, MACGrid, Mesh, ParticleIndexSystem,
                   PdataInt, PdataReal, PdataVec3, RealGrid, Solver, Vec3Grid, VecGrid, vec3, vec4)
from .plugins import (Timings, cgSolveDiffusion, getComponent, setComponent, resetOutflow, apicMapPartsToMAC, apicMapMACGridToParts, extrapolateMACFromWeight, extrapolateMACSimple, markFluidCells, addBuoyancy, addGravity, addGravityNoScale, advectSemiLagrange, computePressureRhs,
                      correctVelocity, flipVelocityUpdate, lastCgStats, mapGridToParts, mapGridToPartsVec3, mapMACToParts,
                      mapPartsToGrid, mapPartsToGridVec3, mapPartsToMAC, setDeterministicP2G, setWallBcs, solvePressure,
                      solvePressureSystem, pushOutofObs, gridParticleIndex, unionParticleLevelset, extrapolateLsSimple,
                      setPartType, markIsolatedFluidCell, addForcePvel, updateVelocityFromDeltaPos, eulerStep,
                      interpolateGrid, interpolateGridVec3, interpolateMACGrid, computeEnergy, computeWaveletCoeffs,
                      vorticityConfinement, applyNoiseVec3, setOpenBound)

from .scene import (Box, Cylinder, Gui, NoiseField, Shape, Sphere, densityInflow, sampleFlagsWithParticles,
                    sampleLevelsetWithParticles, sampleShapeWithParticles)

# module constants, registry.cpp:390-421
GUI = False
DEBUG = False
MT = True
DOUBLEPRECISION = False
CUDA = False
args = sys.argv[1:]
SCENEFILE = sys.argv[0] if sys.argv else ""

# python/defines.py:25-60
FlagFluid, FlagObstacle, FlagEmpty, FlagInflow, FlagOutflow, FlagStick, FlagReserved = 1, 2, 4, 8, 16, 64, 256
TypeFluid, TypeObstacle, TypeEmpty, TypeInflow, TypeOutflow, TypeStick, TypeReserved = 1, 2, 4, 8, 16, 64, 256
IntEuler, IntRK2, IntRK4 = 0, 1, 2
PcNone, PcMIC, PcMGDynamic, PcMGStatic = 0, 1, 2, 3
PtypeSpray, PtypeBubble, PtypeFoam, PtypeTracer = 2, 4, 8, 16
Compression_None, Compression_Zip, Compression_Blosc = 0, 1, 2

_debug_level = 1


def setDebugLevel(level=1):
    """fluidsolver.cpp:217-223"""
    global _debug_level
    _debug_level = int(level)


def mantaMsg(out, level=1):
    """fluidsolver.cpp:210-212"""
    if level <= _debug_level:
        print(out)


def printBuildInfo():
    s = "mantaflow_amd 0.1 64bit fp1 hip gfx950"
    print(s)
    return s


# python/defines.py:14-22
Real = float
false, true = False, True
Vec3 = vec3
Vec4 = vec4


def assertNumpy():
    """fluidsolver.cpp:226-232 (the numpy bridge is always there)"""


# ---- helpers of the reference's regression harness (tools/tests/helperInclude.py) --------------------------------------
# Host-side comparisons of whole grids / pdata (grid.cpp:437-478, 511-536, initplugins.cpp:297-330): not on the hot path.
def gridMaxDiff(g1, g2):
    """max |g1 - g2|, the difference formed in fp32 (grid.cpp:437-444)"""
    g1.parent.sync()
    return float((g1.data - g2.data).abs().max().item()) if g1.data.numel() else 0.0


def gridMaxDiffInt(g1, g2):
    g1.parent.sync()
    return float((g1.data.double() - g2.data.double()).abs().max().item()) if g1.data.numel() else 0.0


def gridMaxDiffVec3(g1, g2):
    """max over cells of the fp64 sum of component differences (grid.cpp:453-469)"""
    g1.parent.sync()
    n = g1.n
    d = (g1.data.double() - g2.data.double()).abs()
    return float((d[:n] + d[n:2 * n] + d[2 * n:3 * n]).max().item()) if n else 0.0


def copyMacToVec3(source, target):
    """grid.cpp:470-478 (both are SoA Vec3 storage here)"""
    target.copyFrom(source)


convertMacToVec3 = copyMacToVec3


def copyLevelsetToReal(source, target):
    """grid.cpp:511-536"""
    target.copyFrom(source)


convertLevelsetToReal = copyLevelsetToReal


def pdataMaxDiff(a, b):
    """initplugins.cpp:297-330"""
    if type(a) is not type(b):
        raise RuntimeError("pdataMaxDiff problem - different pdata types!")
    if a.size() != b.size():
        raise RuntimeError("pdataMaxDiff problem - different pdata sizes!")
    n = a.size()
    if n == 0:
        return 0.0
    a.parent.sync()
    comps = [(a.data[c * a.cap:c * a.cap + n].double() - b.data[c * b.cap:c * b.cap + n].double()).abs() for c in range(a._ncomp)]
    d = comps[0]
    for c in comps[1:]:
        d = d + c
    return float(d.max().item())
