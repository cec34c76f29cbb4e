"""Scene-setup helpers that run once (or are cheap) and therefore stay host-side (shapes / noise / particle
sampling are OUT OF SCOPE for the HIP hot path but needed for scenes to run unchanged).  torch / numpy only.

  Box / Sphere / Cylinder            source/shapes.{h,cpp}: analytic SDFs (shapes.cpp:178-229, 240-242, 367-385)
  sampleFlagsWithParticles           plugin/flip.cpp:33-58, with the reference's RandomStream (MT19937 seed 9832,
                                     util/randomstream.h) reproduced through numpy's legacy-seeded MT19937
  sampleLevelsetWithParticles        plugin/flip.cpp:64-110 (reset/refillEmpty/particleFlag)
  NoiseField / densityInflow         plugin/initplugins.cpp:27-43 with a SMOOTH VALUE NOISE standing in for the
                                     reference's wavelet noise tile (noisefield.cpp) -- an inflow texture, not on the
                                     parity path; documented in DESIGN.md section 5.
"""
import math

import numpy as np
import torch

from . import core
from .core import FlagGrid, Grid, GridBase, LevelsetGrid, MACGrid, PbClass, VecGrid, _to_vec3, vec3
from .plugins import _chk, plugin

f32 = np.float32


def _cell_centres(s):
    sx, sy, sz = s.mGridSize
    k, j, i = np.meshgrid(np.arange(sz, dtype=f32), np.arange(sy, dtype=f32), np.arange(sx, dtype=f32), indexing="ij")
    return i + f32(0.5), j + f32(0.5), k + f32(0.5)


class Shape(PbClass):
    _cname_py, _cname_cpp = "Shape", "Shape"

    def __init__(self, parent, name="", **kw):
        PbClass.__init__(self, parent, name)

    def _sdf(self):
        raise NotImplementedError

    def computeLevelset(self):
        """Shape::computeLevelset, shapes.cpp:28-34: returns a new LevelsetGrid"""
        phi = LevelsetGrid(self.parent)
        phi.from_numpy(self._sdf().astype(f32))
        return phi

    def generateLevelset(self, phi):
        phi.from_numpy(self._sdf().astype(f32))

    def _inside_centres(self):
        x, y, z = _cell_centres(self.parent)
        return self._inside(x, y, z)

    def applyToGrid(self, grid, value=None, respectFlags=None):
        """ApplyShapeToGrid / ApplyShapeToMACGrid, shapes.cpp:40-69"""
        s = self.parent
        sx, sy, sz = s.mGridSize
        keep = None
        if respectFlags is not None:
            keep = (respectFlags.to_numpy() & core.TypeObstacle) != 0
        t = grid.getType()
        if t & GridBase.TypeMAC:
            v = _to_vec3(value)
            a = grid.to_numpy()
            k, j, i = np.meshgrid(np.arange(sz, dtype=f32), np.arange(sy, dtype=f32), np.arange(sx, dtype=f32), indexing="ij")
            h = f32(0.5)
            for c, (px, py, pz, val) in enumerate(((i, j + h, k + h, v.x), (i + h, j, k + h, v.y), (i + h, j + h, k, v.z))):
                m = self._inside(px, py, pz)
                if keep is not None:
                    m &= ~keep
                a[..., c][m] = val
            grid.from_numpy(a)
        else:
            m = self._inside_centres()
            if keep is not None:
                m &= ~keep
            a = grid.to_numpy()
            if t & GridBase.TypeVec3:
                v = _to_vec3(value)
                a[m] = (v.x, v.y, v.z)
            else:
                a[m] = value
            grid.from_numpy(a)

    def applyToGridSmooth(self, grid, sigma=1.0, shift=0, value=None, respectFlags=None):
        phi = self._sdf() - f32(shift)
        a = grid.to_numpy()
        keep = np.zeros(phi.shape, bool) if respectFlags is None else (respectFlags.to_numpy() & core.TypeObstacle) != 0
        m1 = (phi < -sigma) & ~keep
        m2 = (phi >= -sigma) & (phi < sigma) & ~keep
        a[m1] = value
        a[m2] = (f32(value) * (f32(0.5) * (f32(1.0) - phi / f32(sigma))))[m2]
        grid.from_numpy(a)

    def isInside(self, pos):
        p = _to_vec3(pos)
        return bool(self._inside(np.array(f32(p.x)), np.array(f32(p.y)), np.array(f32(p.z))))


class Box(Shape):
    _cname_py = _cname_cpp = "Box"

    def __init__(self, parent, center=None, p0=None, p1=None, size=None, name="", **kw):
        Shape.__init__(self, parent, name)
        # Box::Box, shapes.cpp:137-152
        if center is not None and size is not None:
            c, sz = _to_vec3(center), _to_vec3(size)
            self.p0, self.p1 = c - sz, c + sz
        elif p0 is not None and p1 is not None:
            self.p0, self.p1 = _to_vec3(p0), _to_vec3(p1)
        else:
            raise RuntimeError("Box: specify either p0,p1 or size,center")

    def _inside(self, x, y, z):
        a, b = self.p0, self.p1
        m = (x >= f32(a.x)) & (y >= f32(a.y)) & (x <= f32(b.x)) & (y <= f32(b.y))
        if self.parent.is3D():
            m &= (z >= f32(a.z)) & (z <= f32(b.z))
        return m

    def _sdf(self):
        """BoxSDF, shapes.cpp:178-229 (same case split)"""
        x, y, z = _cell_centres(self.parent)
        p1, p2 = self.p0, self.p1
        x1, y1, z1, x2, y2, z2 = (f32(v) for v in (p1.x, p1.y, p1.z, p2.x, p2.y, p2.z))
        inx, iny, inz = (x <= x2) & (x >= x1), (y <= y2) & (y >= y1), (z <= z2) & (z >= z1)
        mx, my, mz = np.maximum(x - x2, x1 - x), np.maximum(y - y2, y1 - y), np.maximum(z - z2, z1 - z)
        if not self.parent.is3D():
            mz_in = mx
        else:
            mz_in = mz
        sq = lambda a: a * a
        def mins(*a):
            r = a[0]
            for v in a[1:]:
                r = np.minimum(r, v)
            return r
        lines_x = mins(np.sqrt(sq(y1 - y) + sq(z1 - z)), np.sqrt(sq(y2 - y) + sq(z1 - z)), np.sqrt(sq(y1 - y) + sq(z2 - z)), np.sqrt(sq(y2 - y) + sq(z2 - z)))
        lines_y = mins(np.sqrt(sq(x1 - x) + sq(z1 - z)), np.sqrt(sq(x2 - x) + sq(z1 - z)), np.sqrt(sq(x1 - x) + sq(z2 - z)), np.sqrt(sq(x2 - x) + sq(z2 - z)))
        lines_z = mins(np.sqrt(sq(y1 - y) + sq(x1 - x)), np.sqrt(sq(y2 - y) + sq(x1 - x)), np.sqrt(sq(y1 - y) + sq(x2 - x)), np.sqrt(sq(y2 - y) + sq(x2 - x)))
        pts = None
        for cx in (x1, x2):
            for cy in (y1, y2):
                for cz in (z1, z2):
                    dd = np.sqrt(sq(x - cx) + sq(y - cy) + sq(z - cz))
                    pts = dd if pts is None else np.minimum(pts, dd)
        conds = [inx & iny & inz, iny & inz, inx & inz, inx & iny, (x > x1) & (x < x2), (y > y1) & (y < y2), (z > x1) & (z < z2)]
        vals = [np.maximum(mx, np.maximum(my, mz_in)), mx, my, mz, lines_x, lines_y, lines_z]
        return np.select(conds, vals, default=pts).astype(f32)


class Sphere(Shape):
    _cname_py = _cname_cpp = "Sphere"

    def __init__(self, parent, center, radius, scale=None, name="", **kw):
        Shape.__init__(self, parent, name)
        self.center, self.radius = _to_vec3(center), float(radius)
        self.scale = _to_vec3(scale) if scale is not None else vec3(1, 1, 1)

    def _inside(self, x, y, z):
        c, s = self.center, self.scale
        q = ((x - f32(c.x)) / f32(s.x)) ** 2 + ((y - f32(c.y)) / f32(s.y)) ** 2 + ((z - f32(c.z)) / f32(s.z)) ** 2
        return q <= f32(self.radius * self.radius)

    def _sdf(self):
        """SphereSDF, shapes.cpp:303-307: norm((p - center)/scale) - radius"""
        x, y, z = _cell_centres(self.parent)
        c, s = self.center, self.scale
        return (np.sqrt(((x - f32(c.x)) / f32(s.x)) ** 2 + ((y - f32(c.y)) / f32(s.y)) ** 2 + ((z - f32(c.z)) / f32(s.z)) ** 2) - f32(self.radius)).astype(f32)


class Cylinder(Shape):
    _cname_py = _cname_cpp = "Cylinder"

    def __init__(self, parent, center, radius, z, name="", **kw):
        Shape.__init__(self, parent, name)
        # Cylinder::Cylinder, shapes.cpp:312-318: mZ = normalize(mZDir)
        self.center, self.radius = _to_vec3(center), float(radius)
        zd = _to_vec3(z)
        ln = math.sqrt(zd.x ** 2 + zd.y ** 2 + zd.z ** 2)
        self.zlen = ln
        self.zdir = vec3(zd.x / ln, zd.y / ln, zd.z / ln) if ln > 0 else vec3(0, 0, 0)

    def _pz_r(self, x, y, z):
        c, a = self.center, self.zdir
        px, py, pz = x - f32(c.x), y - f32(c.y), z - f32(c.z)
        zz = np.abs(px * f32(a.x) + py * f32(a.y) + pz * f32(a.z))
        r = np.sqrt(np.maximum(px * px + py * py + pz * pz - zz * zz, 0))
        return zz, r

    def _inside(self, x, y, z):
        zz, r = self._pz_r(x, y, z)
        return (zz <= f32(self.zlen)) & (r <= f32(self.radius))

    def _sdf(self):
        """CylinderSDF, shapes.cpp:367-385"""
        x, y, z = _cell_centres(self.parent)
        zz, r = self._pz_r(x, y, z)
        R, Z = f32(self.radius), f32(self.zlen)
        return np.where(zz < Z, np.where(r < R, np.maximum(r - R, zz - Z), r - R),
                        np.where(r < R, np.abs(zz - Z), np.sqrt((zz - Z) ** 2 + (r - R) ** 2))).astype(f32)


# ---------------------------------------------------------------------------------------------------------
# particle sampling with the reference's random stream
# ---------------------------------------------------------------------------------------------------------
class RandomStream(object):
    """util/randomstream.h: MTRand (MT19937, init_genrand seeding) ; getReal() = float(randInt() * (1/4294967295))"""

    def __init__(self, seed):
        self.bg = np.random.MT19937()
        self.bg._legacy_seeding(int(seed))

    def reals(self, n):
        raw = self.bg.random_raw(n).astype(np.float64)
        return (raw * (1.0 / 4294967295.0)).astype(f32)


def _sample_cells(s, cells_ijk, discretization, randomness, rs):
    """positions for every (cell, dk, dj, di) in the reference's loop order (flip.cpp:41-55)"""
    is3d = s.is3D()
    D = int(discretization)
    jlen = f32(f32(randomness) / f32(D))
    disp = f32(1.0 / D)
    nk = D if is3d else 1
    sub = np.array([(di, dj, dk) for dk in range(nk) for dj in range(D) for di in range(D)], np.int64)    # di fastest
    ncell, nsub = cells_ijk.shape[0], sub.shape[0]
    off = (f32(0.5) + sub.astype(f32))                                   # Vec3(0.5+di, ...) as float
    base = np.repeat(cells_ijk.astype(f32), nsub, axis=0)                # pos
    offs = np.tile(off, (ncell, 1))
    subpos = base + disp * offs                                          # pos + disp * Vec3(...)
    r = rs.reals(3 * ncell * nsub).reshape(-1, 3)                        # getVec3: x, y, z drawn in this order
    jit = jlen * (f32(1.0) - (f32(2.0) * r).astype(f32))                 # jlen * (Vec3(1) - 2.0*rand)
    subpos = (subpos + jit).astype(f32)
    if not is3d:
        subpos[:, 2] = f32(0.5)
    return subpos


@plugin
def sampleFlagsWithParticles(flags, parts, discretization, randomness):
    """plugin/flip.cpp:33-58"""
    _chk(flags, FlagGrid, "FlagGrid")
    s = flags.parent
    f = flags.to_numpy()
    kk, jj, ii = np.nonzero(((f & core.TypeObstacle) == 0) & ((f & core.TypeFluid) != 0))    # k, j, i ascending = FOR_IJK order
    cells = np.stack([ii, jj, kk], axis=1)
    pos = _sample_cells(s, cells, discretization, randomness, RandomStream(9832))
    old = parts.get_positions() if parts.np else np.zeros((0, 3), f32)
    oldf = parts.get_flags() if parts.np else np.zeros(0, np.int32)
    # insertBufferedParticles marks new particles PNEW (particle.h:636-663)
    parts.set_positions(np.concatenate([old, pos]), np.concatenate([oldf, np.full(pos.shape[0], core.PNEW, np.int32)]))


@plugin
def sampleShapeWithParticles(shape, flags, parts, discretization, randomness, reset=False, refillEmpty=False, exclude=None):
    """plugin/flip.cpp:109-140: every non-obstacle cell is visited (and draws its random numbers), sub-positions inside
    the shape are kept"""
    _chk(flags, FlagGrid, "FlagGrid")
    if exclude is not None:
        raise RuntimeError("sampleShapeWithParticles: the exclude levelset is outside the hot path")
    s = flags.parent
    if reset:
        parts.clear()
    f = flags.to_numpy()
    m = (f & core.TypeObstacle) == 0
    if refillEmpty:
        m &= (f & core.TypeFluid) == 0
    kk, jj, ii = np.nonzero(m)
    cells = np.stack([ii, jj, kk], axis=1)
    pos = _sample_cells(s, cells, discretization, randomness, RandomStream(9832))
    if pos.shape[0]:
        pos = pos[shape._inside(pos[:, 0], pos[:, 1], pos[:, 2])]
    old = parts.get_positions() if parts.np else np.zeros((0, 3), f32)
    oldf = parts.get_flags() if parts.np else np.zeros(0, np.int32)
    parts.set_positions(np.concatenate([old, pos]), np.concatenate([oldf, np.full(pos.shape[0], core.PNEW, np.int32)]))


class Gui(object):
    """no-op stand-in for the Qt GUI (gui/*): scenes that open a window when `guion` is set keep running headless"""
    def __init__(self, *a, **kw): pass
    def show(self, *a, **kw): pass
    def pause(self, *a, **kw): pass
    def update(self, *a, **kw): pass
    def screenshot(self, *a, **kw): pass
    def setBackgroundMesh(self, *a, **kw): pass
    def setCamPos(self, *a, **kw): pass
    def setCamRot(self, *a, **kw): pass
    def nextRealGrid(self, *a, **kw): pass
    def nextVec3Grid(self, *a, **kw): pass
    def nextMeshDisplay(self, *a, **kw): pass
    def nextPartDisplay(self, *a, **kw): pass
    def nextParts(self, *a, **kw): pass
    def toggleHideGrids(self, *a, **kw): pass
    def windowSize(self, *a, **kw): pass


@plugin
def sampleLevelsetWithParticles(phi, flags, parts, discretization, randomness, reset=False, refillEmpty=False, particleFlag=-1):
    """plugin/flip.cpp:64-110"""
    s = flags.parent
    if reset:
        parts.clear()
    f, ph = flags.to_numpy(), phi.to_numpy()
    m = ((f & core.TypeObstacle) == 0) & (ph < f32(1.733))
    if refillEmpty:
        m &= (f & core.TypeFluid) == 0
    kk, jj, ii = np.nonzero(m)
    cells = np.stack([ii, jj, kk], axis=1)
    pos = _sample_cells(s, cells, discretization, randomness, RandomStream(9832))
    # keep sub-positions with phi.getInterpolated(subpos) <= 0 (trilinear, interpol.h:71-82)
    if pos.shape[0]:
        from . import plugins
        tmp = core.BasicParticleSystem(s)
        tmp.set_positions(pos)
        pr = tmp.create(core.PdataReal)
        plugins.mapGridToParts(phi, tmp, pr, notiming=True)
        pos = pos[pr.to_numpy() <= 0]
    fl = np.full(pos.shape[0], core.PNEW | (0 if particleFlag < 0 else int(particleFlag)), np.int32)
    old = parts.get_positions() if parts.np else np.zeros((0, 3), f32)
    oldf = parts.get_flags() if parts.np else np.zeros(0, np.int32)
    parts.set_positions(np.concatenate([old, pos]), np.concatenate([oldf, fl]))


# ---------------------------------------------------------------------------------------------------------
# inflow texture (approximation of the reference's wavelet noise; scene decoration, not on the parity path)
# ---------------------------------------------------------------------------------------------------------
class NoiseField(PbClass):
    _cname_py, _cname_cpp = "NoiseField", "WaveletNoiseField"

    def __init__(self, parent, fixedSeed=-1, loadFromFile=False, name="", **kw):
        PbClass.__init__(self, parent, name)
        self.posOffset, self.posScale = vec3(0.), vec3(1.)
        self.valOffset, self.valScale = 0.0, 1.0
        self.clamp, self.clampNeg, self.clampPos = False, 0.0, 1.0
        self.timeAnim = 0.0
        self._tile = np.random.default_rng(13322223 if fixedSeed < 0 else fixedSeed).standard_normal((32, 32, 32)).astype(f32) * f32(0.35)

    def evaluate_grid(self):
        """value at Vec3(i,j,k) for every cell (noisefield.h:118-137: pos scaled by posScale * (1/gridSize) ...)"""
        s = self.parent
        sx, sy, sz = s.mGridSize
        k, j, i = np.meshgrid(np.arange(sz, dtype=f32), np.arange(sy, dtype=f32), np.arange(sx, dtype=f32), indexing="ij")
        inv = f32(1.0 / max(sx, sy, sz))
        t = f32(s.timeTotal * self.timeAnim)
        ps, po = _to_vec3(self.posScale), _to_vec3(self.posOffset)
        px = (i * inv + f32(po.x) + t) * f32(ps.x)
        py = (j * inv + f32(po.y) + t) * f32(ps.y)
        pz = (k * inv + f32(po.z) + t) * f32(ps.z)
        T = self._tile
        n = T.shape[0]

        def tri(x, y, z):
            x0, y0, z0 = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64), np.floor(z).astype(np.int64)
            fx, fy, fz = x - x0, y - y0, z - z0
            r = 0
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy) * (fz if dz else 1 - fz)
                        r = r + w * T[(z0 + dz) % n, (y0 + dy) % n, (x0 + dx) % n]
            return r
        v = tri(px, py, pz).astype(f32)
        v = v * f32(self.valScale) + f32(self.valOffset)
        if self.clamp:
            v = np.clip(v, f32(self.clampNeg), f32(self.clampPos))
        return v.astype(f32)


@plugin
def densityInflow(flags, density, noise, shape, scale=1.0, sigma=0):
    """plugin/initplugins.cpp:27-43 (KnApplyNoiseInfl) with the approximate NoiseField above"""
    sdf = shape._sdf()
    f = flags.to_numpy()
    d = density.to_numpy()
    sg = f32(sigma)
    m = ((f & core.TypeFluid) != 0) & ~(sdf > sg)
    with np.errstate(divide="ignore", invalid="ignore"):
        factor = np.clip(1.0 - 0.5 / np.float64(sg) * (sdf.astype(np.float64) + np.float64(sg)), 0.0, 1.0).astype(f32) if sg != 0 else \
            np.where(sdf + sg > 0, f32(0.0), f32(1.0)).astype(f32)
    target = noise.evaluate_grid() * f32(scale) * factor
    upd = m & (d < target)
    d[upd] = target[upd]
    density.from_numpy(d)
