// surface.hip -- the free-surface / particle-maintenance pieces that scenes/benchmark_dam.py runs between the FLIP
// transfers (SURVEY 8f-2 leftovers and 8f-3): projectOutOfBnd, pushOutofObs, gridParticleIndex, unionParticleLevelset,
// extrapolateLsSimple, setPartType, markIsolatedFluidCell, the three ptsplugins one-liners and the levelset set ops.
// Reference: source/plugin/flip.cpp, plugin/ptsplugins.cpp, fastmarch.cpp, particle.h, grid.cpp, levelset.cpp.
#include "common.h"
#include <hipcub/hipcub.hpp>

using namespace mf;

static inline unsigned nblk_n(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK > 0 ? (n + BLOCK - 1) / BLOCK : 1); }
__device__ __forceinline__ bool in_bounds(const Dim& d, int i, int j, int k) {
	return i >= 0 && j >= 0 && k >= 0 && i < d.sx && j < d.sy && k < d.sz;
}
__device__ __forceinline__ int64_t cidx(const Dim& d, int i, int j, int k) { return (int64_t)i + d.Y * j + d.Z * k; }
// Particle positions are GLOBAL grid coordinates; a z-slab window (mf_set_slab_window) holds planes [zoff, zoff + sz) of gsz.
// cell_of: cell of a position, k as the plane inside the window; false when the cell is outside the domain or the window
// (a particle of another slab).  Without a window this is the reference's isInBounds(toVec3i(pos)).
__device__ __forceinline__ bool cell_of(const Dim& d, float x, float y, float z, int& i, int& j, int& k) {
	i = (int)x;
	j = (int)y;
	const int kg = (int)z;
	k = kg - d.zoff;
	return i >= 0 && j >= 0 && kg >= 0 && i < d.sx && j < d.sy && kg < d.gsz && k >= 0 && k < d.sz;
}
// domain-boundary test in global planes (a slab's outer ghost plane is not a domain wall)
__device__ __forceinline__ bool z_wall(const Dim& d, int k, int w) { return d.is3d && (k + d.zoff <= w || k + d.zoff >= d.gsz - 1 - w); }
#define CELL_IJK(d)                                                               \
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;                \
	if (idx >= (d).n) return;                                                     \
	const int i = (int)(idx % (d).sx);                                            \
	const int j = (int)((idx / (d).sx) % (d).sy);                                 \
	const int k = (int)(idx / ((int64_t)(d).sx * (d).sy));                        \
	(void)i; (void)j; (void)k;
#define INTERIOR_B(d, b) (i >= (b) && i < (d).sx - (b) && j >= (b) && j < (d).sy - (b) && (!(d).is3d || (k >= (b) && k < (d).sz - (b))))

// KnProjectOutOfBnd, particle.h:579-590 (std::max(pos, bnd) / std::min(pos, size - bnd))
__global__ void __launch_bounds__(BLOCK)
k_project_out_of_bnd(Dim d, int64_t np, int64_t ps, float* __restrict__ pos, const int32_t* __restrict__ pflag, float bnd, int axis,
                     const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) return;
	float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
	if (axis & 1) x = x > bnd ? x : bnd;
	if (axis & 2) { const float hi = (float)d.sx - bnd; x = hi < x ? hi : x; }
	if (axis & 4) y = y > bnd ? y : bnd;
	if (axis & 8) { const float hi = (float)d.sy - bnd; y = hi < y ? hi : y; }
	if (d.is3d) {
		if (axis & 16) z = z > bnd ? z : bnd;
		if (axis & 32) { const float hi = (float)d.gsz - bnd; z = hi < z ? hi : z; }
	}
	pos[p] = x;
	pos[ps + p] = y;
	pos[2 * ps + p] = z;
}

// knPushOutofObs, plugin/flip.cpp:584-596; getGradient grid.h:556-573; normalize vectorbase.h:421-434
__global__ void __launch_bounds__(BLOCK)
k_push_out_of_obs(Dim d, int64_t np, int64_t ps, float* __restrict__ pos, const int32_t* __restrict__ pflag, const float* __restrict__ phi,
                  float shift, float thresh, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) return;
	const float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
	int i, j, k;
	if (!cell_of(d, x, y, z, i, j, k)) return;
	const float v = interpol1(d, phi, x, y, z);
	if (!(v < thresh)) return;
	if (i > d.sx - 2) i = d.sx - 2;
	if (j > d.sy - 2) j = d.sy - 2;
	if (i < 1) i = 1;
	if (j < 1) j = 1;
	float gx = phi[cidx(d, i + 1, j, k)] - phi[cidx(d, i - 1, j, k)];
	float gy = phi[cidx(d, i, j + 1, k)] - phi[cidx(d, i, j - 1, k)];
	float gz = 0.f;
	if (d.is3d) {
		// getGradient clamps to [1, size - 2] of the whole domain; inside a slab window additionally stay addressable
		int kg = k + d.zoff;
		if (kg > d.gsz - 2) kg = d.gsz - 2;
		if (kg < 1) kg = 1;
		k = kg - d.zoff;
		if (k > d.sz - 2) k = d.sz - 2;
		if (k < 1) k = 1;
		gz = phi[cidx(d, i, j, k + 1)] - phi[cidx(d, i, j, k - 1)];
	}
	const float l = gx * gx + gy * gy + gz * gz;
	const float eps2 = 1e-6f * 1e-6f;
	float nrm;
	if (fabs((double)l - 1.) < (double)eps2) {
		nrm = 1.f;
	} else if (l > eps2) {
		nrm = sqrtf(l);
		const float fac = (float)(1. / (double)nrm);
		gx *= fac;
		gy *= fac;
		gz *= fac;
	} else {
		gx = gy = gz = 0.f;
		nrm = 0.f;
	}
	if (nrm < 1e-6f) return;
	const float f = thresh - v + shift;
	pos[p] = x + gx * f;
	pos[ps + p] = y + gy * f;
	pos[2 * ps + p] = z + gz * f;
}

// ---------------------------------------------------------------------------------------------------------
// gridParticleIndex, plugin/flip.cpp:273-320.  The reference's serial counting sort orders indexSys by (cell, particle
// index); here: per-cell histogram (integer atomics: exact), exclusive scan -> index, stable radix sort of
// (cell key, particle index) pairs -> indexSys.  Skipped particles get the key n (sorted to the end).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK)
k_gpi_keys(Dim d, int64_t np, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ pflag, int32_t* __restrict__ keys,
           int32_t* __restrict__ vals, int32_t* __restrict__ counter) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	int key = (int)d.n;
	if (!(pflag[p] & MF_PDELETE)) {
		int i, j, k;
		if (cell_of(d, pos[p], pos[ps + p], pos[2 * ps + p], i, j, k)) {
			key = (int)cidx(d, i, j, k);
			atomicAdd(&counter[key], 1);
		}
	}
	keys[p] = key;
	vals[p] = (int)p;
}
struct SortScratch {
	void* tmp = nullptr;
	size_t cap = 0;
};
static SortScratch g_sort[16];
static int sort_scratch(size_t need, void** out) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	SortScratch& s = g_sort[dev];
	if (need > s.cap) {
		MF_HIP(hipDeviceSynchronize());
		if (s.tmp) MF_HIP(hipFree(s.tmp));
		MF_HIP(hipMalloc(&s.tmp, need));
		s.cap = need;
	}
	*out = s.tmp;
	return 0;
}

// ComputeUnionLevelsetPindex, plugin/flip.cpp:322-353 (+ setBound(0.5, 0) fused: the boundary test of knSetBoundary with w=0)
__global__ void __launch_bounds__(BLOCK)
k_union_levelset(Dim d, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ isys, int64_t n_indexed,
                 const int32_t* __restrict__ index, float* __restrict__ phi, float radius, const int32_t* __restrict__ ptype, int exclude) {
	CELL_IJK(d)
	const bool bnd = (i <= 0 || i >= d.sx - 1 || j <= 0 || j >= d.sy - 1 || z_wall(d, k, 0));
	if (bnd) {
		phi[idx] = 0.5f;
		return;
	}
	const float gx = (float)i + 0.5f, gy = (float)j + 0.5f, gz = (float)(k + d.zoff) + 0.5f;
	float phiv = (float)((double)radius * 1.0);
	const int r = (int)radius + 1, rZ = d.is3d ? r : 0;
	const float eps2 = 1e-6f * 1e-6f;
	// the cells xj = i-r .. i+r of a row (yj, zj) are consecutive entries of the particle index: one [start, end) range per row -- two index
	// loads instead of 2 (2r + 1), the particles in the reference's order (cell by cell, slot by slot).  Most cells of a liquid scene
	// are far from every particle and only ever read the index.
	const int xlo = i - r < 0 ? 0 : i - r, xhi = i + r > d.sx - 1 ? d.sx - 1 : i + r;
	for (int zj = k - rZ; zj <= k + rZ; zj++)
		for (int yj = j - r; yj <= j + r; yj++) {
				if (!in_bounds(d, xlo, yj, zj)) continue;
				const int64_t c0 = cidx(d, xlo, yj, zj), c1 = cidx(d, xhi, yj, zj);
				const int64_t pStart = index[c0];
				const int64_t pEnd = (c1 + 1 < d.n) ? (int64_t)index[c1 + 1] : n_indexed;
				for (int64_t q = pStart; q < pEnd; q++) {
					const int psrc = isys[q];
					if (ptype && (ptype[psrc] & exclude)) continue;
					const float dx = gx - pos[psrc], dy = gy - pos[ps + psrc], dz = gz - pos[2 * ps + psrc];
					const float l = dx * dx + dy * dy + dz * dz;
					float nr;   // norm(), vectorbase.h:385-389
					if (l <= eps2) nr = 0.f;
					else nr = (fabs((double)l - 1.) < (double)eps2) ? 1.f : sqrtf(l);
					const float cand = fabsf(nr) - radius;
					phiv = cand < phiv ? cand : phiv;
				}
			}
	phi[idx] = phiv;
}

// knSetBoundary, grid.cpp:629-633
__global__ void __launch_bounds__(BLOCK) k_set_bound(Dim d, float* __restrict__ g, float value, int w) {
	CELL_IJK(d)
	const bool bnd = (i <= w || i >= d.sx - 1 - w || j <= w || j >= d.sy - 1 - w || z_wall(d, k, w));
	if (bnd) g[idx] = value;
}

// ---- shape level sets, shapes.cpp:178-229 (Box), 303-307 (Sphere), 367-385 (Cylinder); fp32, left to right ----
#define HD __device__ __forceinline__
#define FMAX(a, b) ((a) > (b) ? (a) : (b))
#define FMIN(a, b) ((a) < (b) ? (a) : (b))
#define SQRT(a) sqrtf(a)
#define FABS(a) fabsf(a)

static HD float shape_sdf_box(int is3d, const float* q, float x, float y, float z) {
	const float x1 = q[0], y1 = q[1], z1 = q[2], x2 = q[3], y2 = q[4], z2 = q[5];
	const int inx = (x <= x2) && (x >= x1), iny = (y <= y2) && (y >= y1), inz = (z <= z2) && (z >= z1);
	const float mx = FMAX(x - x2, x1 - x), my = FMAX(y - y2, y1 - y), mz = FMAX(z - z2, z1 - z);
	if (inx && iny && inz) return FMAX(mx, FMAX(my, is3d ? mz : mx));
	if (iny && inz) return mx;
	if (inx && inz) return my;
	if (inx && iny) return mz;
#define SQ(a) ((a) * (a))
	if (x > x1 && x < x2) {
		const float a = SQRT(SQ(y1 - y) + SQ(z1 - z)), b = SQRT(SQ(y2 - y) + SQ(z1 - z)), c = SQRT(SQ(y1 - y) + SQ(z2 - z)), d = SQRT(SQ(y2 - y) + SQ(z2 - z));
		return FMIN(FMIN(FMIN(a, b), c), d);
	}
	if (y > y1 && y < y2) {
		const float a = SQRT(SQ(x1 - x) + SQ(z1 - z)), b = SQRT(SQ(x2 - x) + SQ(z1 - z)), c = SQRT(SQ(x1 - x) + SQ(z2 - z)), d = SQRT(SQ(x2 - x) + SQ(z2 - z));
		return FMIN(FMIN(FMIN(a, b), c), d);
	}
	if (z > x1 && z < z2) { /* sic: the reference tests z against x1 (shapes.cpp:214) */
		const float a = SQRT(SQ(y1 - y) + SQ(x1 - x)), b = SQRT(SQ(y2 - y) + SQ(x1 - x)), c = SQRT(SQ(y1 - y) + SQ(x2 - x)), d = SQRT(SQ(y2 - y) + SQ(x2 - x));
		return FMIN(FMIN(FMIN(a, b), c), d);
	}
	float best = 0.f;
	int first = 1;
	for (int ix = 0; ix < 2; ix++)
		for (int iy = 0; iy < 2; iy++)
			for (int iz = 0; iz < 2; iz++) {
				const float cx = ix ? x2 : x1, cy = iy ? y2 : y1, cz = iz ? z2 : z1;
				const float dd = SQRT(SQ(x - cx) + SQ(y - cy) + SQ(z - cz));
				best = first ? dd : FMIN(best, dd);
				first = 0;
			}
	return best;
}
static HD float shape_sdf_sphere(const float* q, float x, float y, float z) {
	const float a = (x - q[0]) / q[4], b = (y - q[1]) / q[5], c = (z - q[2]) / q[6];
	return SQRT(a * a + b * b + c * c) - q[3];
}
static HD float shape_sdf_cylinder(const float* q, float x, float y, float z) {
	const float px = x - q[0], py = y - q[1], pz = z - q[2];
	const float zz = FABS(px * q[4] + py * q[5] + pz * q[6]);
	const float r = SQRT(px * px + py * py + pz * pz - zz * zz); /* NaN when rounding makes the difference negative */
	const float R = q[3], Z = q[7];
	if (zz < Z) {
		if (r < R) return FMAX(r - R, zz - Z);
		return r - R;
	}
	if (r < R) return FABS(zz - Z);
	return SQRT(SQ(zz - Z) + SQ(r - R));
#undef SQ
}
#undef HD
#undef FMAX
#undef FMIN
#undef SQRT
#undef FABS
#define HD __device__ __forceinline__
#define FABS(a) fabsf(a)

/* Box::isInside :151-154 (z is tested in 2-D as well), Sphere::isInside :240-242, Cylinder::isInside :324-329 (r^2 < R^2, no root) */
static HD int shape_inside(int kind, const float* q, float x, float y, float z) {
	if (kind == 0) return x >= q[0] && y >= q[1] && z >= q[2] && x <= q[3] && y <= q[4] && z <= q[5];
	if (kind == 1) {
		const float a = (x - q[0]) / q[4], b = (y - q[1]) / q[5], c = (z - q[2]) / q[6];
		return (a * a + b * b + c * c) <= q[3] * q[3];
	}
	const float px = x - q[0], py = y - q[1], pz = z - q[2];
	const float zz = px * q[4] + py * q[5] + pz * q[6];
	if (FABS(zz) > q[7]) return 0;
	const float r2 = (px * px + py * py + pz * pz) - zz * zz;
	return r2 < q[3] * q[3];
}
#undef HD
#undef FABS
struct ShapeParams {
	float q[12];
};
__global__ void __launch_bounds__(BLOCK) k_shape_levelset(Dim d, int kind, ShapeParams P, float* __restrict__ phi) {
	CELL_IJK(d)
	const float x = (float)i + 0.5f, y = (float)j + 0.5f, z = (float)(k + d.zoff) + 0.5f;
	phi[idx] = kind == 0 ? shape_sdf_box(d.is3d, P.q, x, y, z) : (kind == 1 ? shape_sdf_sphere(P.q, x, y, z) : shape_sdf_cylinder(P.q, x, y, z));
}

// resetOutflow, extforces.cpp:134-161
__global__ void __launch_bounds__(BLOCK) k_reset_outflow_parts(Dim d, const int32_t* __restrict__ flags, int64_t np, int64_t ps,
                                                               const float* __restrict__ pos, int32_t* __restrict__ pflag) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	const int pf = pflag[p];
	if (pf & MF_PDELETE) return;
	const int i = (int)pos[p], j = (int)pos[ps + p], k = (int)pos[2 * ps + p];
	if (i < 0 || j < 0 || k < 0 || i >= d.sx || j >= d.sy || k >= d.sz) return;
	if (flags[i + d.Y * j + d.Z * k] & MF_OUTFLOW) pflag[p] = pf | MF_PDELETE;
}
__global__ void __launch_bounds__(BLOCK) k_reset_outflow(int64_t n, int32_t* __restrict__ flags, float* __restrict__ phi, float* __restrict__ real) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= n) return;
	const int f = flags[idx];
	if (!(f & MF_OUTFLOW)) return;
	flags[idx] = (f | MF_EMPTY) & ~MF_FLUID;
	if (phi) phi[idx] = 0.5f;
	if (real) real[idx] = 0.f;
}

// extrapolateLsSimple, fastmarch.cpp:472-522
__global__ void __launch_bounds__(BLOCK) k_els_mark(Dim d, const float* __restrict__ phi, int32_t* __restrict__ tmp, int inside, int b) {
	CELL_IJK(d)
	int t = 0;
	// the b-cell border of the DOMAIN stays unmarked; the outer ghost plane of a slab window is an ordinary cell
	if (i >= b && i < d.sx - b && j >= b && j < d.sy - b && (!d.is3d || (k + d.zoff >= b && k + d.zoff < d.gsz - b))) {
		if (!inside) t = (phi[idx] < 0.) ? 1 : 0;
		else t = (phi[idx] > 0.) ? 1 : 0;
	}
	tmp[idx] = t;
}
// first layer: the reference sweeps serially in place, but it only tests for the value 1 and only writes 2
__global__ void __launch_bounds__(BLOCK) k_els_first(Dim d, int32_t* __restrict__ tmp) {
	CELL_IJK(d)
	if (!INTERIOR_B(d, 1)) return;
	if (tmp[idx]) return;
	bool hit = tmp[idx + 1] == 1 || tmp[idx - 1] == 1 || tmp[idx + d.Y] == 1 || tmp[idx - d.Y] == 1;
	if (d.is3d) hit = hit || tmp[idx + d.Z] == 1 || tmp[idx - d.Z] == 1;
	if (hit) tmp[idx] = 2;
}
// knExtrapolateLsSimple: cells written in pass dd get dd+1 and a new value; the pass only reads cells marked dd
__global__ void __launch_bounds__(BLOCK) k_els_pass(Dim d, float* __restrict__ phi, int32_t* __restrict__ tmp, int dd, float direction) {
	CELL_IJK(d)
	if (!INTERIOR_B(d, 1)) return;
	if (tmp[idx] != 0) return;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	const int cnt = d.is3d ? 6 : 4;
	int nbs = 0;
	float avg = 0.f;
	for (int n = 0; n < cnt; n++)
		if (tmp[idx + nb[n]] == dd) {
			avg += phi[idx + nb[n]];
			nbs++;
		}
	if (nbs > 0) {
		tmp[idx] = dd + 1;
		phi[idx] = avg / nbs + direction;
	}
}
__global__ void __launch_bounds__(BLOCK) k_els_rest(Dim d, float* __restrict__ phi, const int32_t* __restrict__ tmp, float value) {
	CELL_IJK(d)
	if (!INTERIOR_B(d, 1)) return;
	if (tmp[idx] == 0) phi[idx] = value;
}

// KnSetPartType, ptsplugins.cpp:56-59
__global__ void __launch_bounds__(BLOCK)
k_set_part_type(Dim d, const int32_t* __restrict__ flags, int64_t np, int64_t ps, const float* __restrict__ pos, int32_t* __restrict__ ptype,
                int mark, int stype, int cflag) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	const int i = (int)pos[p], j = (int)pos[ps + p], kg = (int)pos[2 * ps + p], k = kg - d.zoff;
	bool in = i >= 0 && j >= 0 && i < d.sx && j < d.sy;
	in = in && (d.is3d ? (kg >= 0 && kg < d.gsz && k >= 0 && k < d.sz) : (kg == 0));
	if (!in) return;
	if ((flags[cidx(d, i, j, k)] & cflag) && (ptype[p] & stype)) ptype[p] = mark;
}

// knMarkIsolatedFluidCell, grid.cpp:987-1005.  In place like the reference: a cell that gets marked has no fluid
// neighbour, so no other cell's decision depends on it.
__global__ void __launch_bounds__(BLOCK) k_mark_isolated(Dim d, int32_t* __restrict__ flags, int mark) {
	CELL_IJK(d)
	if (!(flags[idx] & MF_FLUID)) return;
	auto fl = [&](int64_t o) { return (idx + o >= 0 && idx + o < d.n) ? (flags[idx + o] & MF_FLUID) : 0; };
	if (fl(-1) || fl(1) || fl(-d.Y) || fl(d.Y)) return;
	if (d.is3d && (fl(-d.Z) || fl(d.Z))) return;
	flags[idx] = mark;
}

// ptsplugins.cpp:20-53
__global__ void __launch_bounds__(BLOCK)
k_add_force_pvel(int64_t np, int64_t ps, float* __restrict__ v, float dx, float dy, float dz, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (ptype && (ptype[p] & exclude)) return;
	v[p] += dx;
	v[ps + p] += dy;
	v[2 * ps + p] += dz;
}
__global__ void __launch_bounds__(BLOCK)
k_vel_from_delta_pos(int64_t np, int64_t ps, const float* __restrict__ pos, float* __restrict__ v, const float* __restrict__ xp, float over_dt,
                     const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (ptype && (ptype[p] & exclude)) return;
#pragma unroll
	for (int c = 0; c < 3; c++) v[c * ps + p] = (pos[c * ps + p] - xp[c * ps + p]) * over_dt;
}
__global__ void __launch_bounds__(BLOCK)
k_euler_step(int64_t np, int64_t ps, float* __restrict__ pos, const float* __restrict__ v, float dt, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (ptype && (ptype[p] & exclude)) return;
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c * ps + p] += v[c * ps + p] * dt;
}
// KnJoin / KnSubtract, levelset.cpp:107-118
__global__ void __launch_bounds__(BLOCK) k_ls_join(int64_t n, float* __restrict__ a, const float* __restrict__ b) {
	const int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (i < n) a[i] = b[i] < a[i] ? b[i] : a[i];
}
__global__ void __launch_bounds__(BLOCK)
k_ls_subtract(int64_t n, float* __restrict__ a, const float* __restrict__ b, const int32_t* __restrict__ flags, int stype) {
	const int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (i >= n) return;
	if (flags && (flags[i] & stype) == 0) return;
	if (b[i] < 0.) a[i] = b[i] * -1.f;
}

// knInterpolateGridTempl (grid.h:576-581) / KnInterpolateMACGrid (waveletturbulence.cpp:59-71): one thread per target cell
template <bool MAC, int OS>
__global__ void __launch_bounds__(BLOCK)
k_interpolate_grid(Dim t, float* __restrict__ target, Dim s, const float* __restrict__ source, int ncomp, float sfx, float sfy, float sfz,
                   float ox, float oy, float oz) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= t.n) return;
	const int i = (int)(idx % t.sx), j = (int)((idx / t.sx) % t.sy), k = (int)(idx / ((int64_t)t.sx * t.sy));
	const float px = (float)i * sfx + ox, py = (float)j * sfy + oy;
	float pz = (float)(k + t.zoff) * sfz + oz;       // global plane of the target cell; the source samplers take global positions
	if (MAC && OS == 2) {
		// getInterpolatedHi(pos - 0.5 e_c, 2)[c] = interpolCubicMAC(..)[c] = interpolCubic<Vec3>((pos - 0.5 e_c) + 0.5 e_c)[c]
		target[idx] = interpol_cubic_mac<0>(s, source, px - 0.5f, py, pz);
		target[t.n + idx] = interpol_cubic_mac<1>(s, source, px, py - 0.5f, pz);
		target[2 * t.n + idx] = s.is3d ? interpol_cubic_mac<2>(s, source, px, py, pz - 0.5f) : 0.f;
	} else if (MAC) {
		// MACGrid::getInterpolatedHi -> interpolMAC (grid.h:269-275); one component of each evaluation is kept
		float vx, vy, vz;
		interpol_mac(s, source, px - 0.5f, py, pz, vx, vy, vz);
		target[idx] = vx;
		interpol_mac(s, source, px, py - 0.5f, pz, vx, vy, vz);
		target[t.n + idx] = vy;
		if (s.is3d) {
			interpol_mac(s, source, px, py, pz - 0.5f, vx, vy, vz);
			target[2 * t.n + idx] = vz;
		} else {
			target[2 * t.n + idx] = 0.f;
		}
	} else {
		if (!s.is3d) pz = 0.f;
		for (int c = 0; c < ncomp; c++) {
			if (OS == 2) target[c * t.n + idx] = (ncomp == 3) ? interpol_cubic<true>(s, source + c * s.n, px, py, pz) : interpol_cubic<false>(s, source + c * s.n, px, py, pz);
			else target[c * t.n + idx] = interpol1(s, source + c * s.n, px, py, pz);
		}
	}
}

extern "C" {

int mf_interpolate_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source, int ncomp,
                        float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* stream) {
	MF_TRY(check_dim(tsx, tsy, tsz));
	MF_TRY(check_dim(ssx, ssy, ssz));
	if (ncomp != 1 && ncomp != 3) return fail("ncomp must be 1 or 3");
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order %d", orderSpace);
	const Dim t = mkdim(tsx, tsy, tsz), s = mkdim_src(ssx, ssy, ssz);   // each grid under its own slab window
	if (orderSpace == 2)
		hipLaunchKernelGGL((k_interpolate_grid<false, 2>), dim3(nblk_n(t.n)), dim3(BLOCK), 0, (hipStream_t)stream, t, target, s, source, ncomp, sfx, sfy, sfz, ox, oy, oz);
	else
		hipLaunchKernelGGL((k_interpolate_grid<false, 1>), dim3(nblk_n(t.n)), dim3(BLOCK), 0, (hipStream_t)stream, t, target, s, source, ncomp, sfx, sfy, sfz, ox, oy, oz);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_interpolate_mac_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source,
                            float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* stream) {
	MF_TRY(check_dim(tsx, tsy, tsz));
	MF_TRY(check_dim(ssx, ssy, ssz));
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order %d", orderSpace);
	const Dim t = mkdim(tsx, tsy, tsz), s = mkdim_src(ssx, ssy, ssz);   // each grid under its own slab window
	if (orderSpace == 2)
		hipLaunchKernelGGL((k_interpolate_grid<true, 2>), dim3(nblk_n(t.n)), dim3(BLOCK), 0, (hipStream_t)stream, t, target, s, source, 3, sfx, sfy, sfz, ox, oy, oz);
	else
		hipLaunchKernelGGL((k_interpolate_grid<true, 1>), dim3(nblk_n(t.n)), dim3(BLOCK), 0, (hipStream_t)stream, t, target, s, source, 3, sfx, sfy, sfz, ox, oy, oz);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_project_out_of_bnd(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag, float bnd, int axis,
                          const int32_t* ptype, int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_project_out_of_bnd, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, np, pstride, pos, pflag, bnd, axis, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_push_out_of_obs(int sx, int sy, int sz, int64_t np, int64_t pstride, float* pos, const int32_t* pflag, const float* phiObs,
                       float shift, float thresh, const int32_t* ptype, int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_push_out_of_obs, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, np, pstride, pos, pflag, phiObs, shift, thresh, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_grid_particle_index(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos, const int32_t* pflag,
                           int32_t* indexSys, int32_t* index, int32_t* counter, int32_t* keys, int32_t* vals,
                           int64_t* n_indexed_host, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np >= ((int64_t)1 << 31)) return fail("gridParticleIndex: too many particles for 32-bit slots");
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(counter, 0, sizeof(int32_t) * d.n, st));
	if (np > 0) {
		hipLaunchKernelGGL(k_gpi_keys, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, np, pstride, pos, pflag, keys, vals, counter);
		MF_LAUNCH_CHECK();
	}
	// index = exclusive prefix sum of the per-cell counts
	size_t scan_bytes = 0, sort_bytes = 0;
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, counter, index, (int)d.n, st));
	int end_bit = 1;
	while (end_bit < 31 && (((int64_t)1 << end_bit) <= d.n)) end_bit++;
	if (np > 0) MF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys, keys + np, vals, vals + np, (int)np, 0, end_bit, st));
	void* tmp = nullptr;
	MF_TRY(sort_scratch((scan_bytes > sort_bytes ? scan_bytes : sort_bytes) + 256, &tmp));
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, scan_bytes, counter, index, (int)d.n, st));
	if (np > 0) {
		MF_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, sort_bytes, keys, keys + np, vals, vals + np, (int)np, 0, end_bit, st));
		// the skipped particles (key n) sort to the tail; indexSys takes the whole array, only [0, n_indexed) is meaningful
		MF_HIP(hipMemcpyAsync(indexSys, vals + np, sizeof(int32_t) * np, hipMemcpyDeviceToDevice, st));
	}
	if (n_indexed_host) {
		int32_t last_idx = 0, last_cnt = 0;
		MF_HIP(hipMemcpyAsync(&last_idx, index + d.n - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
		MF_HIP(hipMemcpyAsync(&last_cnt, counter + d.n - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
		MF_HIP(hipStreamSynchronize(st));
		*n_indexed_host = (int64_t)last_idx + last_cnt;
	}
	return 0;
}

int mf_union_particle_levelset(int sx, int sy, int sz, int64_t np, int64_t pstride, const float* pos, const int32_t* indexSys,
                               int64_t n_indexed, const int32_t* index, float* phi, float radiusFactor, const int32_t* ptype,
                               int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	(void)np;
	const Dim d = mkdim(sx, sy, sz);
	// calculateRadiusFactor (flip.cpp:198-200) in double, returned as Real; radius = 0.5 * that, rounded to Real
	const float rf = (float)((d.is3d ? sqrt(3.) : sqrt(2.)) * ((double)radiusFactor + .01));
	const float radius = (float)(0.5 * (double)rf);
	hipLaunchKernelGGL(k_union_levelset, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, pstride, pos, indexSys, n_indexed, index, phi, radius, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_grid_set_bound(int sx, int sy, int sz, float* grid, float value, int boundaryWidth, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_set_bound, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, grid, value, boundaryWidth);
	MF_LAUNCH_CHECK();
	return 0;
}

__global__ void __launch_bounds__(BLOCK) k_shape_apply(Dim d, int kind, ShapeParams P, int gridkind, float* __restrict__ gf, float v0, float v1,
                                                       float v2, const int32_t* __restrict__ respect) {
	CELL_IJK(d)
	if (respect && (respect[idx] & MF_OBSTACLE)) return;
	const float x = (float)i, y = (float)j, z = (float)(k + d.zoff);
	if (gridkind == 2) {
		if (shape_inside(kind, P.q, x, y + 0.5f, z + 0.5f)) gf[idx] = v0;
		if (shape_inside(kind, P.q, x + 0.5f, y, z + 0.5f)) gf[d.n + idx] = v1;
		if (shape_inside(kind, P.q, x + 0.5f, y + 0.5f, z)) gf[2 * d.n + idx] = v2;
	} else if (shape_inside(kind, P.q, x + 0.5f, y + 0.5f, z + 0.5f)) {
		if (gridkind == 0) gf[idx] = v0;
		else if (gridkind == 3) ((int32_t*)gf)[idx] = (int32_t)v0;
		else {
			gf[idx] = v0;
			gf[d.n + idx] = v1;
			gf[2 * d.n + idx] = v2;
		}
	}
}
int mf_shape_apply_to_grid(int sx, int sy, int sz, int kind, const float* params_host, int gridkind, void* grid, const float* value_host,
                           const int32_t* respectFlags, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (kind < 0 || kind > 2 || gridkind < 0 || gridkind > 3) return fail("mf_shape_apply_to_grid: unknown shape or grid kind");
	const Dim d = mkdim(sx, sy, sz);
	ShapeParams P;
	for (int q = 0; q < 12; q++) P.q[q] = params_host[q];
	hipLaunchKernelGGL(k_shape_apply, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, kind, P, gridkind, (float*)grid, value_host[0], value_host[1],
	                   value_host[2], respectFlags);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_shape_levelset(int sx, int sy, int sz, int kind, const float* params_host, float* phi, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (kind < 0 || kind > 2) return fail("mf_shape_levelset: unknown shape kind");
	const Dim d = mkdim(sx, sy, sz);
	ShapeParams P;
	for (int q = 0; q < 12; q++) P.q[q] = params_host[q];
	hipLaunchKernelGGL(k_shape_levelset, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, kind, P, phi);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_reset_outflow(int sx, int sy, int sz, int32_t* flags, float* phi, float* real, int64_t np, int64_t ps, const float* pos,
                     int32_t* pflag, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	if (np > 0 && pos && pflag)
		hipLaunchKernelGGL(k_reset_outflow_parts, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, flags, np, ps, pos, pflag);
	hipLaunchKernelGGL(k_reset_outflow, dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d.n, flags, phi, real);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_extrapolate_ls_simple(int sx, int sy, int sz, float* phi, int distance, int inside, int include_walls, int32_t* tmp,
                             void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	const unsigned nb = nblk_n(d.n);
	const float direction = inside ? -1.f : 1.f;
	hipLaunchKernelGGL(k_els_mark, dim3(nb), dim3(BLOCK), 0, st, d, phi, tmp, inside, (inside && include_walls) ? 0 : 1);
	hipLaunchKernelGGL(k_els_first, dim3(nb), dim3(BLOCK), 0, st, d, tmp);
	for (int dd = 2; dd < 1 + distance; dd++) hipLaunchKernelGGL(k_els_pass, dim3(nb), dim3(BLOCK), 0, st, d, phi, tmp, dd, direction);
	hipLaunchKernelGGL(k_els_rest, dim3(nb), dim3(BLOCK), 0, st, d, phi, tmp, (float)(direction * (distance + 2)));
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_set_part_type(int sx, int sy, int sz, const int32_t* flags, int64_t np, int64_t pstride, const float* pos, int32_t* ptype,
                     int mark, int stype, int cflag, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_set_part_type, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, np, pstride, pos, ptype, mark, stype, cflag);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_mark_isolated_fluid_cell(int sx, int sy, int sz, int32_t* flags, int mark, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_mark_isolated, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, mark);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_add_force_pvel(int64_t np, int64_t pstride, float* pvel, float ax, float ay, float az, float dt, const int32_t* ptype,
                      int exclude, void* stream) {
	if (np <= 0) return 0;
	const float dx = ax * dt, dy = ay * dt, dz = az * dt;
	hipLaunchKernelGGL(k_add_force_pvel, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, np, pstride, pvel, dx, dy, dz, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_update_velocity_from_delta_pos(int64_t np, int64_t pstride, const float* pos, float* pvel, const float* xprev, float dt,
                                      const int32_t* ptype, int exclude, void* stream) {
	if (np <= 0) return 0;
	const float over_dt = (float)(1.0 / (double)dt);
	hipLaunchKernelGGL(k_vel_from_delta_pos, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, np, pstride, pos, pvel, xprev, over_dt, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_euler_step(int64_t np, int64_t pstride, float* pos, const float* pvel, float dt, const int32_t* ptype, int exclude,
                  void* stream) {
	if (np <= 0) return 0;
	hipLaunchKernelGGL(k_euler_step, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, np, pstride, pos, pvel, dt, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_levelset_join(int64_t n, float* phi, const float* other, void* stream) {
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_ls_join, dim3(nblk_n(n)), dim3(BLOCK), 0, (hipStream_t)stream, n, phi, other);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_levelset_subtract(int64_t n, float* phi, const float* other, const int32_t* flags, int subtractType, void* stream) {
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_ls_subtract, dim3(nblk_n(n)), dim3(BLOCK), 0, (hipStream_t)stream, n, phi, other, flags, subtractType);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
