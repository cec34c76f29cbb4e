"""Host-side mirror of the reference's object model for the hot path.

Mirrors (names, argument meaning, defaults, error behaviour) of:
  FluidSolver / Solver           source/fluidsolver.{h,cpp}
  Grid<T>, MACGrid, FlagGrid     source/grid.{h,cpp}
  LevelsetGrid                   source/levelset.h (container + join/subtract only)
  BasicParticleSystem, Pdata*    source/particle.{h,cpp}
  vec3                           source/pwrapper/pvec3.cpp
Storage is a torch tensor on the active library's device (``cuda`` for the HIP product library); all arithmetic on
the hot path goes through the C ABI (mantaflow_amd._lib).  Grids are dense, x fastest; Vec3/MAC grids and particle
vectors are structure-of-arrays ([3][N]); the reference's AoS [z][y][x][3] view exists only at the numpy bridge.
"""
import ctypes
import math

import functools
import types

import numpy as np
import torch

from . import _lib

# ---------------------------------------------------------------------------------------------------------
# constants: FlagGrid::CellType (grid.h:306-320), particle status (particle.h:34-43), python/defines.py:25-60
# ---------------------------------------------------------------------------------------------------------
TypeNone, TypeFluid, TypeObstacle, TypeEmpty, TypeInflow, TypeOutflow, TypeOpen, TypeStick = 0, 1, 2, 4, 8, 16, 32, 64
TypeSurface, TypeReserved, TypeBandInterface, TypeTemp = 128, 256, 512, 32768
PNONE, PNEW, PSPRAY, PBUBBLE, PFOAM, PTRACER, PDELETE, PINVALID = 0, 1, 2, 4, 8, 16, 1 << 10, 1 << 30
VECTOR_EPSILON = 1e-6


class vec3(object):
    """manta.vec3 -- float[3] with component-wise + - * / against vec3 or scalar (pvec3.cpp:40-145)."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x=None, y=None, z=None):
        if x is None:
            x = y = z = 0.0
        elif isinstance(x, (vec3, tuple, list)) and y is None:
            x, y, z = (x.x, x.y, x.z) if isinstance(x, vec3) else x
        elif y is None:
            y = z = x
        elif z is None:
            raise TypeError("vec3 takes 0, 1 or 3 numbers")
        self.x, self.y, self.z = float(np.float32(x)), float(np.float32(y)), float(np.float32(z))

    @staticmethod
    def _c(o):
        return o if isinstance(o, vec3) else vec3(o) if isinstance(o, (int, float, tuple, list, np.number)) else None

    def _bin(self, o, f):
        o = vec3._c(o)
        if o is None:
            return NotImplemented
        return vec3(f(self.x, o.x), f(self.y, o.y), f(self.z, o.z))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __radd__(self, o): return vec3._c(o)._bin(self, lambda a, b: a + b)
    def __rsub__(self, o): return vec3._c(o)._bin(self, lambda a, b: a - b)
    def __rmul__(self, o): return vec3._c(o)._bin(self, lambda a, b: a * b)
    def __rtruediv__(self, o): return vec3._c(o)._bin(self, lambda a, b: a / b)
    def __neg__(self): return vec3(-self.x, -self.y, -self.z)
    def __iter__(self): return iter((self.x, self.y, self.z))
    def __getitem__(self, i): return (self.x, self.y, self.z)[i]
    def __eq__(self, o):
        o = vec3._c(o)
        return o is not None and (self.x, self.y, self.z) == (o.x, o.y, o.z)
    def __repr__(self): return "[%+4.6f,%+4.6f,%+4.6f]" % (self.x, self.y, self.z)
    def max(self): return max(self.x, self.y, self.z)


class vec4(object):
    """manta.vec4 (pvec3.cpp:280-389): float[4] with members x, y, z, t; built from nothing (zeros), one number (broadcast) or four;
    unlike vec3 it has no arithmetic (tp_as_number is NULL in the reference)"""
    __slots__ = ("x", "y", "z", "t")

    def __init__(self, x=None, y=None, z=None, t=None):
        if x is None:
            if not (y is None and z is None and t is None):
                raise RuntimeError("Invalid partial init of vec4")
            x = y = z = t = 0.0
        elif y is None and z is None and t is None:
            y = z = t = x
        elif y is None or z is None or t is None:
            raise RuntimeError("Invalid partial init of vec4")
        self.x, self.y, self.z, self.t = (float(np.float32(v)) for v in (x, y, z, t))

    def __iter__(self): return iter((self.x, self.y, self.z, self.t))
    def __getitem__(self, i): return (self.x, self.y, self.z, self.t)[i]
    def __repr__(self): return "[%+4.6f,%+4.6f,%+4.6f,%+4.6f]" % (self.x, self.y, self.z, self.t)


def _to_vec3(v, what="Vec3"):
    if isinstance(v, vec3):
        return v
    if isinstance(v, (tuple, list)) and len(v) == 3:
        return vec3(*v)
    raise RuntimeError("can't convert argument to %s" % what)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(dev):
    if dev == "cuda" or (isinstance(dev, str) and dev.startswith("cuda")):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    return None


def _method_kwargs(fn):
    """every PYTHON() member accepts the universal `notiming` / `nocheck` keywords (codegen_python.cpp:37, pconvert.cpp:461)"""
    @functools.wraps(fn)
    def w(self, *a, **kw):
        kw.pop("notiming", None)
        kw.pop("nocheck", None)
        return fn(self, *a, **kw)
    return w


class PbClass(object):
    """Base of every solver-owned object (pwrapper/pclass.h): parent solver + name."""
    _T = ""

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name, attr in list(cls.__dict__.items()):
            if isinstance(attr, types.FunctionType) and not name.startswith("_"):
                setattr(cls, name, _method_kwargs(attr))

    def __init__(self, parent, name=""):
        if parent is None:
            raise RuntimeError("New class %s: no parent given -- specify using parent=xxx !" % type(self).__name__)
        self.parent = parent
        self.name = name or ""

    def getParent(self): return self.parent
    def setName(self, n): self.name = n
    def getName(self): return self.name
    # introspection attributes of every wrapped object (registry.cpp:123-133, 320-326): C class name without / with template
    @property
    def _class(self): return type(self)._cname_cpp
    @property
    def _cname(self):
        t = getattr(type(self), "_T", "")
        return type(self)._cname_cpp + ("<%s>" % t if t else "")


class SolverLib(object):
    """The C ABI as seen by ONE solver: every call is made under that solver's z-slab window (zoff, gsz) -- (0, 0) for an
    ordinary solver whose grids are the whole domain.  The window is thread-local state of the library; binding it to the
    solver means a slab solver and a plain solver (or two slab solvers of different resolution, as in waveletTurbulence.py) can
    live in one process without inheriting each other's coordinates."""

    def __init__(self, lib, solver):
        self._lib, self._solver = lib, solver
        self.backend, self.device, self.cdll, self.path = lib.backend, lib.device, lib.cdll, lib.path

    def call(self, name, *args):
        # set on every call: the window is thread-local state of the shared object, so a cache per Library object would go
        # stale under a second host thread or a second Library on the same .so (two thread-local stores per call)
        lib, w = self._lib, self._solver._slab_window
        lib.cdll.mf_set_slab_window(w[0], w[1])
        return lib.call(name, *args)

    def call2(self, src_solver, name, *args):
        """a call that reads a grid of another solver (interpolateGrid & co.): that grid's window goes in as the source window"""
        lib, w = self._lib, src_solver._slab_window
        lib.cdll.mf_set_slab_window_source(w[0], w[1])
        return self.call(name, *args)


# ---------------------------------------------------------------------------------------------------------
# FluidSolver (python name Solver), fluidsolver.{h,cpp}
# ---------------------------------------------------------------------------------------------------------
class FluidSolver(PbClass):
    _cname_py, _cname_cpp = "Solver", "FluidSolver"

    def __init__(self, gridSize, dim=3, fourthDim=-1, name="", **kw):
        gs = _to_vec3(gridSize, "Vec3i")
        self.mGridSize = (int(gs.x), int(gs.y), int(gs.z))
        if dim not in (2, 3):
            raise RuntimeError("Only 2D and 3D solvers allowed.")
        if dim == 2 and self.mGridSize[2] != 1:
            raise RuntimeError("Trying to create 2D solver with size.z != 1")
        self.mDim = dim
        PbClass.__init__(self, self, name)
        # fluidsolver.cpp:106-109
        self.timestep = 1.0
        self.timeTotal = 0.0
        self.frame = 0
        self.mCount = 0
        self.cfl = 1000.0
        self.timestepMin = 1.0
        self.timestepMax = 1.0
        self.frameLength = 1.0
        self.timePerFrame = 0.0
        self.mLockDt = False
        self._slab_window = (0, 0)
        self.lib = SolverLib(_lib.get(), self)
        self.device = self.lib.device
        self._pool = {}    # dtype/ncomp -> list of free tensors  (GridStorage, fluidsolver.cpp:34-50)
        self._live = 0
        self.timings = {}

    # --- accessors ---
    def getGridSize(self): return vec3(*self.mGridSize)
    def is2D(self): return self.mDim == 2
    def is3D(self): return self.mDim == 3
    def getDt(self): return float(np.float32(self.timestep))
    def globalGridSize(self):
        """size of the whole domain: mGridSize, or -- for the solver of a z-slab (slab.SlabDomain) -- the undivided grid's"""
        return getattr(self, "_global_size", None) or self.mGridSize
    def getDx(self): return 1.0 / max(self.globalGridSize())   # a z-slab reports the whole domain's dx
    @property
    def ncells(self): return self.mGridSize[0] * self.mGridSize[1] * self.mGridSize[2]
    @property
    def stream(self): return _stream(self.device)

    # --- temp-grid pool: LIFO stack per element type; freshly handed grids are zeroed (grid.cpp:49-60) ---
    def _alloc(self, kind, zero=True):
        ncomp, dtype = (3, torch.float32) if kind == "vec" else ((1, torch.int32) if kind == "int" else (1, torch.float32))
        free = self._pool.setdefault(kind, [])
        if free:
            t = free.pop()
            if zero:
                t.zero_()
        else:
            if self._live > 200:
                raise RuntimeError("too many temp grids used -- are they released properly ?")
            t = torch.zeros(ncomp * self.ncells, dtype=dtype, device=self.device)
        self._live += 1
        return t

    def _release(self, kind, t):
        self._live -= 1
        self._pool.setdefault(kind, []).append(t)

    # --- time stepping, fluidsolver.cpp:143-204 ---
    def step(self, frame=-1):
        self.timePerFrame = float(np.float32(self.timePerFrame + self.timestep))
        self.timeTotal = float(np.float32(self.timeTotal + self.timestep))
        self.mCount += 1
        if (self.timePerFrame + VECTOR_EPSILON) > self.frameLength:
            self.frame += 1
            self.timeTotal = float(self.frame) * self.frameLength
            self.timePerFrame = 0.0
            self.mLockDt = False
        if frame >= 0:
            self.frame = frame

    def adaptTimestep(self, maxVel):
        f32 = np.float32
        mvt = f32(maxVel) * f32(self.timestep)
        if not self.mLockDt:
            dt = f32(f32(self.timestep) * f32(self.cfl / (float(mvt) + 1e-05)))
            dt = max(min(dt, f32(self.timestepMax)), f32(self.timestepMin))
            if (self.timePerFrame + float(dt) * 1.05) > self.frameLength:
                dt = f32((self.frameLength - self.timePerFrame) + 1e-04)
            elif (self.timePerFrame + float(dt) + self.timestepMin) > self.frameLength or \
                    (self.timePerFrame + (float(dt) * 1.25)) > self.frameLength:
                dt = f32((self.frameLength - self.timePerFrame + 1e-04) * 0.5)
                self.mLockDt = True
            self.timestep = float(dt)
        if not (self.timestep > (self.timestepMin / 2.)):
            raise RuntimeError("Invalid dt encountered! Shouldnt happen...")

    def printMemInfo(self):
        print("Allocated grids: %s live, pooled %s" % (self._live, {k: len(v) for k, v in self._pool.items()}))

    def create(self, type, name="", **kw):
        """Solver.create(Type, ...) -> object with this solver as parent (fluidsolver.cpp:129-140)."""
        if not callable(type):
            raise RuntimeError("can't convert argument to PbType")
        return type(parent=self, name=name, **kw)

    def sync(self):
        if self.device != "cpu":
            torch.cuda.current_stream().synchronize()


Solver = FluidSolver


# ---------------------------------------------------------------------------------------------------------
# grids, grid.{h,cpp}
# ---------------------------------------------------------------------------------------------------------
class GridBase(PbClass):
    TypeNone, TypeReal, TypeInt, TypeVec3, TypeMAC, TypeLevelset, TypeFlags = 0, 1, 2, 4, 8, 16, 32
    _kind, _ncomp = "real", 1

    def __init__(self, parent, show=True, name="", **kw):
        PbClass.__init__(self, parent, name)
        s = parent
        self.sx, self.sy, self.sz = s.mGridSize
        self.n = self.sx * self.sy * self.sz
        self.data = s._alloc(self._kind)       # zeroed (Grid ctor calls clear(), grid.cpp:58)
        self._external = False

    def __del__(self):
        try:
            if not self._external and self.data is not None:
                self.parent._release(self._kind, self.data)   # release stores the *current* pointer (post-swap)
        except Exception:
            pass

    # geometry
    def getSizeX(self): return self.sx
    def getSizeY(self): return self.sy
    def getSizeZ(self): return self.sz
    def getSize(self): return vec3(self.sx, self.sy, self.sz)
    def is3D(self): return self.parent.is3D()
    def is4D(self): return False
    def getDx(self): return 1.0 / max(self.parent.globalGridSize())
    def getType(self): return self._gtype
    def getGridType(self): return self._gtype
    @property
    def dims(self): return (self.sx, self.sy, self.sz)
    @property
    def ptr(self): return _ptr(self.data)
    def _call(self, fn, *args): return self.parent.lib.call(fn, *args)

    def _check_same(self, o):
        if (o.sx, o.sy, o.sz) != (self.sx, self.sy, self.sz):
            raise RuntimeError("different grid resolutions [%d,%d,%d] vs [%d,%d,%d]" % (o.sx, o.sy, o.sz, self.sx, self.sy, self.sz))

    # element-wise API of Grid<T> (grid.h:113-180, grid.cpp:228-330)
    def clear(self):
        self._call("mf_fill_f32", self.data.numel(), self.ptr, 0.0, self.parent.stream)

    def copyFrom(self, a, copyType=True):
        self._check_same(a)
        self._call("mf_copy_f32", self.data.numel(), self.ptr, a.ptr, self.parent.stream)
        return self

    def swap(self, other):
        """pointer swap, grid.cpp:100-111"""
        self._check_same(other)
        self.data, other.data = other.data, self.data

    def getDataPointer(self): return "%x" % self.data.data_ptr()

    def save(self, name):
        """Grid<T>::save, grid.cpp:157-179 (.uni, .raw, .npz)"""
        return _grid_save(self, str(name))

    def load(self, name):
        """Grid<T>::load, grid.cpp:135-155"""
        return _grid_load(self, str(name))

    # numpy bridge (plugin/numpyconvert.cpp:145-223): [z][y][x](,[c]) arrays
    def to_numpy(self):
        a = self.data.detach().cpu().numpy()
        if self._ncomp == 1:
            return a.reshape(self.sz, self.sy, self.sx).copy()
        return np.ascontiguousarray(a.reshape(3, self.sz, self.sy, self.sx).transpose(1, 2, 3, 0))

    def from_numpy(self, arr):
        arr = np.asarray(arr)
        if self._ncomp == 1:
            flat = np.ascontiguousarray(arr.reshape(self.n))
        else:
            flat = np.ascontiguousarray(arr.reshape(self.sz, self.sy, self.sx, 3).transpose(3, 0, 1, 2)).reshape(3 * self.n)
        t = torch.from_numpy(flat.astype(np.int32 if self._kind == "int" else np.float32, copy=False))
        self.data.copy_(t.to(self.data.device))
        return self


# ---------------------------------------------------------------------------------------------------------
# .uni / .raw grid files (fileio/iogrids.cpp:36-44, 255-292, 386-513): gzip stream of "MNT3" + UniHeader + the raw
# element array (x fastest; Vec3 grids as 3 floats per cell).  Data format either side of the hot path.
# ---------------------------------------------------------------------------------------------------------
_UNI_HEADER = "<6i252siQ"       # dimX dimY dimZ gridType elementType bytesPerElement info[252] dimT timestamp = 288 B


def _unify_grid_type(t):
    """unifyGridType, iogrids.cpp:213-221"""
    if t & GridBase.TypeReal: t |= GridBase.TypeLevelset
    if t & GridBase.TypeLevelset: t |= GridBase.TypeReal
    if t & GridBase.TypeVec3: t |= GridBase.TypeMAC
    if t & GridBase.TypeMAC: t |= GridBase.TypeVec3
    return t


def _grid_save(g, name):
    import gzip, struct, time as _time
    if "." not in name:
        raise RuntimeError("file '%s' does not have an extension" % name)
    ext = name[name.rfind("."):]
    raw = g.to_numpy().astype(np.int32 if g._kind == "int" else np.float32, copy=False).tobytes()
    if ext == ".raw":
        with gzip.open(name, "wb", compresslevel=1) as f:
            f.write(raw)
    elif ext == ".uni":
        et = 0 if (g._gtype & GridBase.TypeInt) else (1 if (g._gtype & GridBase.TypeReal) else 2)
        info = b"mantaflow_amd 0.1 64bit fp1 hip gfx950"
        head = struct.pack(_UNI_HEADER, g.sx, g.sy, g.sz, g._gtype, et, 4 * g._ncomp, info, 0, int(_time.time() * 1000))
        with gzip.open(name, "wb", compresslevel=1) as f:
            f.write(b"MNT3")
            f.write(head)
            f.write(raw)
    elif ext == ".npz":
        np.savez_compressed(name, arr_0=g.to_numpy())
    else:
        raise RuntimeError("file '%s' filetype not supported" % name)
    return 1


def _grid_load(g, name):
    import gzip, struct
    if "." not in name:
        raise RuntimeError("file '%s' does not have an extension" % name)
    ext = name[name.rfind("."):]
    dt = np.int32 if g._kind == "int" else np.float32
    nbytes = 4 * g._ncomp * g.n
    shape = (g.sz, g.sy, g.sx) if g._ncomp == 1 else (g.sz, g.sy, g.sx, 3)
    if ext == ".raw":
        with gzip.open(name, "rb") as f:
            raw = f.read()
        if len(raw) != nbytes:
            raise RuntimeError("can't read raw file, stream length does not match, %d vs %d" % (nbytes, len(raw)))
    elif ext == ".uni":
        with gzip.open(name, "rb") as f:
            ident = f.read(4)
            if ident != b"MNT3":
                raise RuntimeError("readGridUni: Unknown header '%s' " % ident.decode(errors="replace"))
            hb = f.read(struct.calcsize(_UNI_HEADER))
            if len(hb) != struct.calcsize(_UNI_HEADER):
                raise RuntimeError("can't read file, no header present")
            dx, dy, dz, gtype, etype, bpe, info, dimt, stamp = struct.unpack(_UNI_HEADER, hb)
            if (dx, dy, dz) != (g.sx, g.sy, g.sz):
                raise RuntimeError("grid dim doesn't match, [%d,%d,%d] vs [%d,%d,%d]" % (dx, dy, dz, g.sx, g.sy, g.sz))
            if _unify_grid_type(gtype) != _unify_grid_type(g._gtype):
                raise RuntimeError("grid type doesn't match %d vs %d" % (gtype, g._gtype))
            if bpe != 4 * g._ncomp:
                raise RuntimeError("grid element size doesn't match %d vs %d" % (bpe, 4 * g._ncomp))
            raw = f.read(nbytes)
    elif ext == ".npz":
        g.from_numpy(np.load(name)["arr_0"])
        return 1
    else:
        raise RuntimeError("file '%s' filetype not supported" % name)
    g.from_numpy(np.frombuffer(raw, dtype=dt).reshape(shape).copy())
    return 1


class Grid(GridBase):
    """Grid<Real> (python: RealGrid)."""
    _kind, _ncomp, _gtype = "real", 1, GridBase.TypeReal
    _cname_py, _cname_cpp, _T = "RealGrid", "Grid", "Real"

    def setConst(self, v): self._call("mf_fill_f32", self.n, self.ptr, float(v), self.parent.stream)
    def addConst(self, v): self._call("mf_grid_add_const", self.n, self.ptr, float(v), self.parent.stream)
    def multConst(self, v): self._call("mf_grid_mult_const", self.n, self.ptr, float(v), self.parent.stream)
    def clamp(self, lo, hi): self._call("mf_grid_clamp", self.n, self.ptr, float(lo), float(hi), self.parent.stream)
    def add(self, a): self._check_same(a); self._call("mf_grid_add", self.n, self.ptr, a.ptr, self.parent.stream)
    def sub(self, a): self._check_same(a); self._call("mf_grid_sub", self.n, self.ptr, a.ptr, self.parent.stream)
    def mult(self, a): self._check_same(a); self._call("mf_grid_mult", self.n, self.ptr, a.ptr, self.parent.stream)
    def addScaled(self, a, f): self._check_same(a); self._call("mf_grid_scaled_add", self.n, self.ptr, a.ptr, float(f), self.parent.stream)
    def safeDivide(self, a): self._check_same(a); self._call("mf_grid_safe_divide", self.n, self.ptr, a.ptr, self.parent.stream)
    def stomp(self, th): self._call("mf_grid_stomp", self.n, self.ptr, float(th), self.parent.stream)

    def setBound(self, value, boundaryWidth=1):
        """Grid::setBound -> knSetBoundary, grid.cpp:629-637"""
        self._call("mf_grid_set_bound", self.sx, self.sy, self.sz, self.ptr, float(value), int(boundaryWidth), self.parent.stream)

    def getMaxAbs(self):
        r = ctypes.c_float()
        self._call("mf_grid_max_abs", self.n, self.ptr, ctypes.byref(r), self.parent.stream)
        return r.value

    def _minmax(self):
        lo, hi = ctypes.c_float(), ctypes.c_float()
        self._call("mf_grid_min_max", self.n, self.ptr, ctypes.byref(lo), ctypes.byref(hi), self.parent.stream)
        return lo.value, hi.value

    def getMax(self): return self._minmax()[1]
    def getMin(self): return self._minmax()[0]


class IntGrid(GridBase):
    _kind, _ncomp, _gtype = "int", 1, GridBase.TypeInt
    _cname_py, _cname_cpp, _T = "IntGrid", "Grid", "int"

    def setConst(self, v): self._call("mf_fill_i32", self.n, self.ptr, int(v), self.parent.stream)
    def clear(self): self._call("mf_fill_i32", self.n, self.ptr, 0, self.parent.stream)

    # scene-side conveniences of Grid<int> (grid.cpp:370-380): exact integer reductions on the device tensor
    def _sync(self): self.parent.sync()
    def getMin(self): self._sync(); return float(self.data.min().item())
    def getMax(self): self._sync(); return float(self.data.max().item())
    def getMaxAbs(self): return max(abs(self.getMin()), abs(self.getMax()))
    def sub(self, a): self._check_same(a); self._sync(); self.data.sub_(a.data)
    def add(self, a): self._check_same(a); self._sync(); self.data.add_(a.data)
    def addConst(self, v): self._sync(); self.data.add_(int(v))
    def multConst(self, v): self._sync(); self.data.mul_(int(v))
    def addScaled(self, a, f): self._check_same(a); self._sync(); self.data.add_(a.data * int(f))


class VecGrid(GridBase):
    """Grid<Vec3> (python: VecGrid / Vec3Grid), SoA storage."""
    _kind, _ncomp, _gtype = "vec", 3, GridBase.TypeVec3
    _cname_py, _cname_cpp, _T = "VecGrid", "Grid", "Vec3"

    def setConst(self, v):
        v = _to_vec3(v)
        for c, x in enumerate((v.x, v.y, v.z)):
            self._call("mf_fill_f32", self.n, _ptr(self.data[c * self.n:]), float(x), self.parent.stream)

    def multConst(self, v):
        v = _to_vec3(v)
        for c, x in enumerate((v.x, v.y, v.z)):
            self._call("mf_grid_mult_const", self.n, _ptr(self.data[c * self.n:]), float(x), self.parent.stream)

    def addConst(self, v):
        v = _to_vec3(v)
        for c, x in enumerate((v.x, v.y, v.z)):
            self._call("mf_grid_add_const", self.n, _ptr(self.data[c * self.n:]), float(x), self.parent.stream)

    def addScaled(self, a, f):
        """Grid<Vec3>::addScaled(a, Vec3 factor): me += a * factor, component-wise (grid.cpp:283-285)"""
        self._check_same(a)
        f = _to_vec3(f)
        for c, x in enumerate((f.x, f.y, f.z)):
            self._call("mf_grid_scaled_add", self.n, _ptr(self.data[c * self.n:]), _ptr(a.data[c * a.n:]), float(x), self.parent.stream)

    def add(self, a): self._check_same(a); self._call("mf_grid_add", 3 * self.n, self.ptr, a.ptr, self.parent.stream)
    def sub(self, a): self._check_same(a); self._call("mf_grid_sub", 3 * self.n, self.ptr, a.ptr, self.parent.stream)
    def mult(self, a): self._check_same(a); self._call("mf_grid_mult", 3 * self.n, self.ptr, a.ptr, self.parent.stream)
    def safeDivide(self, a): self._check_same(a); self._call("mf_grid_safe_divide", 3 * self.n, self.ptr, a.ptr, self.parent.stream)

    def stomp(self, th):
        th = _to_vec3(th)
        for c, x in enumerate((th.x, th.y, th.z)):
            self._call("mf_grid_stomp", self.n, _ptr(self.data[c * self.n:]), float(x), self.parent.stream)

    def getMaxAbs(self):
        r = ctypes.c_float()
        self._call("mf_grid_max_abs_vec3", self.n, self.ptr, ctypes.byref(r), self.parent.stream)
        return r.value

    getMax = getMaxAbs

    def getMin(self):
        """sqrt(CompMinVec), grid.cpp:364-366: smallest normSquare (x*x + y*y + z*z in fp32)"""
        self.parent.sync()
        d, n = self.data, self.n
        return float((d[:n] * d[:n] + d[n:2 * n] * d[n:2 * n] + d[2 * n:3 * n] * d[2 * n:3 * n]).min().sqrt().item())

    def setBound(self, value, boundaryWidth=1):
        """Grid<Vec3>::setBound -> knSetBoundary, grid.cpp:629-637"""
        v = _to_vec3(value)
        for c, x in enumerate((v.x, v.y, v.z)):
            self._call("mf_grid_set_bound", self.sx, self.sy, self.sz, _ptr(self.data[c * self.n:]), float(x), int(boundaryWidth), self.parent.stream)


Vec3Grid = VecGrid


class MACGrid(VecGrid):
    _gtype = GridBase.TypeMAC | GridBase.TypeVec3
    _cname_py, _cname_cpp, _T = "MACGrid", "MACGrid", ""


class LevelsetGrid(Grid):
    _gtype = GridBase.TypeLevelset | GridBase.TypeReal
    _cname_py, _cname_cpp, _T = "LevelsetGrid", "LevelsetGrid", ""

    @staticmethod
    def invalidTimeValue(): return -1000.0   # levelset.h:45

    def join(self, o):
        """LevelsetGrid::join -> KnJoin, levelset.cpp:107-111"""
        self._check_same(o)
        self._call("mf_levelset_join", self.n, self.ptr, o.ptr, self.parent.stream)

    def subtract(self, o, flags=None, subtractType=0):
        """LevelsetGrid::subtract -> KnSubtract, levelset.cpp:113-118 (this = -o where o < 0)"""
        self._check_same(o)
        self._call("mf_levelset_subtract", self.n, self.ptr, o.ptr, None if flags is None else flags.ptr,
                   int(subtractType), self.parent.stream)

    def createMesh(self, mesh):
        """marching-cubes surface extraction is GUI/mesh output, outside the solver hot path: accepted and ignored"""
        return None


class FlagGrid(IntGrid):
    _gtype = GridBase.TypeFlags | GridBase.TypeInt
    _cname_py, _cname_cpp, _T = "FlagGrid", "FlagGrid", ""
    TypeNone, TypeFluid, TypeObstacle, TypeEmpty, TypeInflow, TypeOutflow, TypeOpen, TypeStick = 0, 1, 2, 4, 8, 16, 32, 64

    def __init__(self, parent, dim=3, show=True, name="", **kw):
        IntGrid.__init__(self, parent, show=show, name=name)

    def _view(self):
        return self.data.view(self.sz, self.sy, self.sx)

    def initDomain(self, boundaryWidth=0, wall="xXyYzZ", open="      ", inflow="      ", outflow="      ", phiWalls=None):
        """FlagGrid::initDomain + initBoundaries, grid.cpp:798-908 (index ops on the device, bit-exact)."""
        types, isset = [0] * 6, [False] * 6
        wall, open, inflow, outflow = (s + "      " for s in (wall, open, inflow, outflow))
        faces = "xXyYzZ"
        for i in range(6):
            for f in range(6 if self.is3D() else 4):
                if isset[f]:
                    continue
                ch = faces[f]
                if open[i] == ch:
                    types[f], isset[f] = TypeOpen, True
                elif inflow[i] == ch:
                    types[f], isset[f] = TypeInflow, True
                elif outflow[i] == ch:
                    types[f], isset[f] = TypeOutflow, True
                elif wall[i] == ch:
                    types[f], isset[f] = TypeObstacle, True
        w = int(boundaryWidth)
        v = self._view()
        v.fill_(TypeEmpty)
        # same overwrite order as initBoundaries: x-, x+, y-, y+, z-, z+
        v[:, :, :w + 1] = types[0]
        v[:, :, max(self.sx - 1 - w, 0):] = types[1]
        v[:, :w + 1, :] = types[2]
        v[:, max(self.sy - 1 - w, 0):, :] = types[3]
        if self.is3D():
            v[:w + 1, :, :] = types[4]
            v[max(self.sz - 1 - w, 0):, :, :] = types[5]
        if phiWalls is not None:
            self._init_phi_walls(phiWalls, w, types, isset, wall)

    def _init_phi_walls(self, phi, w, types, isset, wall):
        # InitMin/MaxXWall etc., grid.cpp:750-796: phi = min(dist - 0.5 - w, phi) for every wall side
        dev = self.data.device
        kk, jj, ii = torch.meshgrid(torch.arange(self.sz, device=dev), torch.arange(self.sy, device=dev),
                                    torch.arange(self.sx, device=dev), indexing="ij")
        p = torch.full((self.sz, self.sy, self.sx), 1000000000.0, dtype=torch.float64, device=dev)
        sides = [(ii - 0.5 - w), (self.sx - ii - 1.5 - w), (jj - 0.5 - w), (self.sy - jj - 1.5 - w),
                 (kk - 0.5 - w), (self.sz - kk - 1.5 - w)]
        for f in range(6 if self.is3D() else 4):
            if types[f] == TypeObstacle:
                p = torch.minimum(p, sides[f].to(torch.float64))
        phi.data.copy_(p.to(torch.float32).reshape(-1))

    def fillGrid(self, type=TypeFluid):
        """grid.cpp:922-927"""
        d = self.data
        keep = (d & (TypeObstacle | TypeInflow | TypeOutflow | TypeOpen)) != 0
        d.copy_(torch.where(keep, d, (d & ~(TypeEmpty | TypeFluid)) | int(type)))

    def updateFromLevelset(self, levelset):
        """grid.cpp:910-920"""
        d, phi = self.data, levelset.data
        upd = ((d & (TypeObstacle | TypeOutflow)) == 0) & (phi > LevelsetGrid.invalidTimeValue())
        newv = (d & ~(TypeEmpty | TypeFluid)) | torch.where(phi <= 0, TypeFluid, TypeEmpty).to(torch.int32)
        d.copy_(torch.where(upd, newv, d))

    def countCells(self, flag, bnd=0, mask=None):
        v = self._view()
        if bnd > 0:
            v = v[(slice(bnd, -bnd) if self.is3D() else slice(None)), bnd:-bnd, bnd:-bnd]
        return int(((v & int(flag)) != 0).sum().item())


RealGrid = Grid


# ---------------------------------------------------------------------------------------------------------
# particles, particle.{h,cpp}: BasicParticleSystem (pos + flag; the fork's pos0 is not on the hot path)
# ---------------------------------------------------------------------------------------------------------
class ParticleDataImpl(PbClass):
    _ncomp, _dtype = 1, torch.float32

    def __init__(self, parent, name="", **kw):
        PbClass.__init__(self, parent, name)
        self.sys = None
        self.data = torch.zeros(0, dtype=self._dtype, device=parent.device)
        self.cap = 0      # component stride (== pstride of the ABI)

    def _attach(self, sys):
        self.sys = sys
        self.resize(sys.np, sys.cap)

    def resize(self, np_, cap=None):
        cap = max(cap if cap is not None else np_, np_)
        if cap != self.cap:
            new = torch.zeros(self._ncomp * cap, dtype=self._dtype, device=self.parent.device)
            keep = min(self.cap, cap)
            for c in range(self._ncomp):
                new[c * cap:c * cap + keep] = self.data[c * self.cap:c * self.cap + keep]
            self.data, self.cap = new, cap

    pyResize = resize
    @property
    def ptr(self): return _ptr(self.data)
    def size(self): return self.sys.np if self.sys else 0
    def clear(self): self.data.zero_()

    def to_numpy(self):
        n = self.size()
        a = self.data.detach().cpu().numpy()
        if self._ncomp == 1:
            return a[:n].copy()
        return np.stack([a[c * self.cap:c * self.cap + n] for c in range(3)], axis=1)

    def from_numpy(self, arr):
        arr = np.asarray(arr)
        n = self.size()
        if self._ncomp == 1:
            self.data[:n] = torch.from_numpy(np.ascontiguousarray(arr.reshape(n))).to(self.data.device, self.data.dtype)
        else:
            arr = arr.reshape(n, 3)
            for c in range(3):
                self.data[c * self.cap:c * self.cap + n] = torch.from_numpy(np.ascontiguousarray(arr[:, c])).to(self.data.device, self.data.dtype)
        return self

    def copyFrom(self, o):
        self.data.copy_(o.data)
        return self


class PdataReal(ParticleDataImpl):
    _cname_py, _cname_cpp, _T = "PdataReal", "ParticleDataImpl", "Real"
    def setConst(self, v): self.data.fill_(float(v))


class PdataInt(ParticleDataImpl):
    _dtype = torch.int32
    _cname_py, _cname_cpp, _T = "PdataInt", "ParticleDataImpl", "int"
    def setConst(self, v): self.data.fill_(int(v))

    def setConstRange(self, s, begin, end):
        """ParticleDataImpl::setConstRange, particle.cpp: [begin, end)"""
        self.data[int(begin):int(end)] = int(s)


class PdataVec3(ParticleDataImpl):
    _ncomp = 3
    _cname_py, _cname_cpp, _T = "PdataVec3", "ParticleDataImpl", "Vec3"

    def setConst(self, v):
        v = _to_vec3(v)
        for c, x in enumerate((v.x, v.y, v.z)):
            self.data[c * self.cap:(c + 1) * self.cap] = float(x)


class ParticleIndexSystem(PbClass):
    """ParticleSystem<ParticleIndexData> (particle.h:228-240): sourceIndex per slot, filled by gridParticleIndex"""
    _cname_py, _cname_cpp, _T = "ParticleIndexSystem", "ParticleIndexSystem", ""

    def __init__(self, parent, name="", **kw):
        PbClass.__init__(self, parent, name)
        self.data = torch.zeros(0, dtype=torch.int32, device=parent.device)
        self.np = 0

    def size(self): return self.np
    pySize = size

    def to_numpy(self): return self.data[:self.np].detach().cpu().numpy().copy()


class Mesh(PbClass):
    """placeholder so that scenes which create a Mesh for the GUI keep running; meshes are outside the hot path"""
    _cname_py, _cname_cpp, _T = "Mesh", "Mesh", ""

    def __init__(self, parent, name="", **kw):
        PbClass.__init__(self, parent, name)


class BasicParticleSystem(PbClass):
    _cname_py, _cname_cpp, _T = "BasicParticleSystem", "BasicParticleSystem", ""

    def __init__(self, parent, name="", **kw):
        PbClass.__init__(self, parent, name)
        dev = parent.device
        self.np, self.cap = 0, 0
        self.pos = torch.zeros(0, dtype=torch.float32, device=dev)     # SoA [3][cap]
        self.flag = torch.zeros(0, dtype=torch.int32, device=dev)
        self.pdata = []

    def create(self, type, name="", **kw):
        """ParticleBase::create, particle.cpp:77-101: new pdata field sized like the system"""
        pd = type(parent=self.parent, name=name)
        self.registerPdata(pd)
        return pd

    def registerPdata(self, pd):
        self.pdata.append(pd)
        pd._attach(self)

    def pySize(self): return self.np
    def size(self): return self.np
    def getSizeSlow(self): return self.np

    def resizeAll(self, n, cap=None):
        cap = max(cap if cap is not None else n, n)
        if cap != self.cap:
            dev = self.parent.device
            newp = torch.zeros(3 * cap, dtype=torch.float32, device=dev)
            newf = torch.zeros(cap, dtype=torch.int32, device=dev)
            keep = min(self.cap, cap)
            for c in range(3):
                newp[c * cap:c * cap + keep] = self.pos[c * self.cap:c * self.cap + keep]
            newf[:keep] = self.flag[:keep]
            self.pos, self.flag, self.cap = newp, newf, cap
        self.np = n
        for pd in self.pdata:
            pd.resize(n, self.cap)

    def clear(self): self.resizeAll(0)

    # numpy bridge: positions [np][3], flags [np]
    def set_positions(self, arr, flags=None):
        arr = np.asarray(arr, dtype=np.float32).reshape(-1, 3)
        self.resizeAll(arr.shape[0])
        for c in range(3):
            self.pos[c * self.cap:c * self.cap + self.np] = torch.from_numpy(np.ascontiguousarray(arr[:, c])).to(self.pos.device)
        if flags is None:
            self.flag[:self.np] = 0
        else:
            self.flag[:self.np] = torch.from_numpy(np.asarray(flags, dtype=np.int32)).to(self.flag.device)

    def get_positions(self):
        a = self.pos.detach().cpu().numpy()
        return np.stack([a[c * self.cap:c * self.cap + self.np] for c in range(3)], axis=1)

    def get_flags(self): return self.flag[:self.np].detach().cpu().numpy().copy()

    def getPosPdata(self, target): target.data.copy_(self.pos)
    def setPosPdata(self, source): self.pos.copy_(source.data)

    def projectOutOfBnd(self, flags, bnd, plane="xXyYzZ", ptype=None, exclude=0):
        """ParticleSystem::projectOutOfBnd, particle.h:592-604"""
        if self.np == 0:
            return
        axis = sum(1 << i for i, ch in enumerate("xXyYzZ") if ch in plane)
        s = self.parent
        s.lib.call("mf_project_out_of_bnd", flags.sx, flags.sy, flags.sz, self.np, self.cap, _ptr(self.pos), _ptr(self.flag),
                   float(bnd), axis, None if ptype is None else ptype.ptr, int(exclude), s.stream)

    def addParticle(self, pos):
        p = _to_vec3(pos)
        old = self.get_positions()
        fl = self.get_flags()
        self.set_positions(np.concatenate([old, np.array([[p.x, p.y, p.z]], np.float32)]), np.concatenate([fl, [0]]))

    def advectInGrid(self, flags, vel, integrationMode, deleteInObstacle=True, stopInObstacle=True, skipNew=False,
                     ptype=None, exclude=0):
        """ParticleSystem::advectInGrid, particle.h:526-550"""
        s = self.parent
        if self.np == 0:
            return
        scratch = None
        if s.lib.backend != "hip":      # CPU implementations of the ABI need x0/u/uTotal scratch
            scratch = torch.zeros(9 * self.cap, dtype=torch.float32, device=s.device)
        s.lib.call("mf_advect_in_grid", flags.sx, flags.sy, flags.sz, flags.ptr, vel.ptr, self.np, self.cap,
                   _ptr(self.pos), _ptr(self.flag), s.getDt(), int(integrationMode), int(bool(deleteInObstacle)),
                   int(bool(stopInObstacle)), int(bool(skipNew)), None if ptype is None else ptype.ptr, int(exclude),
                   _ptr(scratch), s.stream)
