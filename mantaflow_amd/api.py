"""Everything a scene gets from `from manta import *` (reference: module "manta", pwrapper/registry.cpp:390-421,
495-608, plus the constants of source/python/defines.py)."""
import sys

from .core import (BasicParticleSystem, FlagGrid, FluidSolver, Grid, IntGrid, LevelsetGrid, MACGrid, Mesh, ParticleIndexSystem,
                   PdataInt, PdataReal, PdataVec3, RealGrid, Solver, Vec3Grid, VecGrid, vec3)
from .plugins import (Timings, resetOutflow, apicMapPartsToMAC, apicMapMACGridToParts, extrapolateMACFromWeight, extrapolateMACSimple, markFluidCells, addBuoyancy, addGravity, addGravityNoScale, advectSemiLagrange, computePressureRhs,
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
