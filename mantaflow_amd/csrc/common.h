// common.h -- shared host/device helpers of libmanta_hip.so (gfx950 only).
//
// Numerics contract: every kernel keeps the reference's evaluation order and its float/double promotion
// points (cited per kernel), and the library is compiled with -ffp-contract=off so that no FMA is formed
// where the reference (gcc, x86-64, no -march) rounds twice.  IEEE fp32 divide/sqrt are hipcc's defaults.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/manta_hip.h"
#include <float.h>

namespace mf {

// ---- error plumbing ---------------------------------------------------------------------------
extern thread_local char g_err[512];
int fail(const char* fmt, ...);
#define MF_HIP(call)                                                                                  \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) return mf::fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)
#define MF_LAUNCH_CHECK() MF_HIP(hipGetLastError())
#define MF_TRY(expr)           \
	do {                       \
		int r_ = (expr);       \
		if (r_) return r_;     \
	} while (0)

// ---- per-device workspace (reduction partials, device scalars, pinned readback) ---------------------
struct Workspace {
	double* partials;   // [MAX_BLOCKS * 4]
	float* fpartials;   // [MAX_BLOCKS * 4]
	void* scalars;      // device scalar block, 4 KiB: CgScalars / reduction results at 0, mf_cg_solve's liquid-scene flags at WS_PCG_FLAGS
	void* host;         // pinned host mirror, 4 KiB
	int* tilework;      // counters for work-queue kernels
};
constexpr int MAX_BLOCKS = 16384;
constexpr int WS_PCG_FLAGS = 1024;   // byte offset in Workspace::scalars of {outside_bad, x-range lo, x-range hi} (k_cg_outside_zero, k_pack_xrange)
int get_workspace(Workspace** ws);

// ---- grid geometry ---------------------------------------------------------------------------------
struct Dim {
	int sx, sy, sz;
	int is3d;
	int zoff, gsz;    // z-slab window: this grid holds planes [zoff, zoff+sz) of a global grid of gsz planes
	int64_t Y, Z, n;  // strides (X == 1); Z == 0 in 2-D (reference grid.cpp:56)
};
extern thread_local int g_slab_zoff, g_slab_gsz;  // set by mf_set_slab_window; (0,0) = the grid is the whole domain
static inline Dim mkdim(int sx, int sy, int sz) {
	Dim d;
	d.sx = sx;
	d.sy = sy;
	d.sz = sz;
	d.is3d = sz > 1;
	d.zoff = g_slab_gsz > 0 ? g_slab_zoff : 0;
	d.gsz = g_slab_gsz > 0 ? g_slab_gsz : sz;
	d.Y = sx;
	d.Z = d.is3d ? (int64_t)sx * sy : 0;
	d.n = (int64_t)sx * sy * sz;
	return d;
}
// the same for the SOURCE grid of a call that reads a grid of another resolution (interpolateGrid & co.): its own window,
// mf_set_slab_window_source
extern thread_local int g_slab_src_zoff, g_slab_src_gsz;
static inline Dim mkdim_src(int sx, int sy, int sz) {
	Dim d = mkdim(sx, sy, sz);
	d.zoff = g_slab_src_gsz > 0 ? g_slab_src_zoff : 0;
	d.gsz = g_slab_src_gsz > 0 ? g_slab_src_gsz : sz;
	return d;
}
// global plane index -> index inside the local window, kept addressable for the ghost fringe (whose results are
// discarded); the identity when the grid is the whole domain
__device__ __forceinline__ int local_z(const Dim& d, int zi, int hi_off) {
	if (d.sz <= 1) return zi;
	zi -= d.zoff;
	const int hi = d.sz - 1 - hi_off;
	return zi < 0 ? 0 : (zi > hi ? hi : zi);
}
static inline int check_dim(int sx, int sy, int sz) {
	if (sx < 2 || sy < 2 || sz < 1) return fail("invalid grid size %dx%dx%d", sx, sy, sz);
	if ((int64_t)sx * sy * sz >= (int64_t)1 << 31) return fail("grid too large for 32-bit cell indices");
	return 0;
}

// p2g_ordered.hip: parallel particle->grid transfers with the reference's serial summation order (bit-exact)
int p2g_ordered_mac(const Dim& d, float* vel, float* weight, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                    const float* pvel, const int32_t* ptype, int exclude, hipStream_t st);
int p2g_ordered_cell(const Dim& d, int ncomp, float* target, float* wsum, int64_t np, int64_t ps, const float* pos,
                     const int32_t* pflag, const float* psrc, hipStream_t st);

int p2g_ordered_apic(const Dim& d, float* vel, float* mass, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                     const float* pvel, const float* cpx, const float* cpy, const float* cpz, const int32_t* ptype, int exclude,
                     hipStream_t st);

constexpr int BLOCK = 256;
static inline int blocks_for(int64_t n, int per_block, int cap = MAX_BLOCKS) {
	int64_t b = (n + per_block - 1) / per_block;
	if (b < 1) b = 1;
	if (b > cap) b = cap;
	return (int)b;
}

// XCD-aware block remap: blocks are dealt round-robin over the 8 XCDs (blockIdx % 8 share an L2), so give
// each XCD one contiguous range of work items; speed only, never correctness.
__device__ __forceinline__ int xcd_swizzle(int b, int nb) {
	if (nb & 7) return b;
	return (b & 7) * (nb >> 3) + (b >> 3);
}

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
	return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}
// block-wide sum for blockDim.x == BLOCK (4 waves); result valid in thread 0
__device__ __forceinline__ double block_sum(double v) {
	__shared__ double sh[BLOCK / 64];
	v = wave_sum(v);
	const int w = threadIdx.x >> 6;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[w] = v;
	__syncthreads();
	if (threadIdx.x == 0) {
		v = sh[0];
		for (int i = 1; i < (int)(blockDim.x >> 6); i++) v += sh[i];
	}
	return v;
}
__device__ __forceinline__ void block_minmax(float& lo, float& hi) {
	__shared__ float shl[BLOCK / 64], shh[BLOCK / 64];
	lo = wave_min(lo);
	hi = wave_max(hi);
	const int w = threadIdx.x >> 6;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) {
		shl[w] = lo;
		shh[w] = hi;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		lo = shl[0];
		hi = shh[0];
		for (int i = 1; i < (int)(blockDim.x >> 6); i++) {
			lo = fminf(lo, shl[i]);
			hi = fmaxf(hi, shh[i]);
		}
	}
}

// ---- partials that another workgroup of the SAME launch folds (the beta step as the tail of the backward MIC sweep): written with
// one agent-scope (sc1, write-through) store whose completion the writer waits for before its workgroup reports in, read with
// agent-scope (sc1) loads -- the protocol of the sweeps' face granules, no fence
__device__ __forceinline__ void part_store(double* p, double v) {
	__hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ double part_load(const double* p) {
	return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// strided_sum + block_sum of a 256-thread block, executed by threads 0..255 of a workgroup of any size (every thread calls: barriers);
// result valid in thread 0.  SC1: the partials are of this launch (part_store)
template <bool SC1>
__device__ __forceinline__ double tail_sum256(const double* p, int nb) {
	__shared__ double sh_t[4];
	const int tid = threadIdx.x;
	double acc = 0.0;
	if (tid < 256) {
		int i = tid;
		const int S = 256;
		for (; i + 7 * S < nb; i += 8 * S) {
			double v[8];
#pragma unroll
			for (int q = 0; q < 8; q++) v[q] = SC1 ? part_load(p + i + q * S) : p[i + q * S];
#pragma unroll
			for (int q = 0; q < 8; q++) acc += v[q];
		}
		for (; i < nb; i += S) acc += SC1 ? part_load(p + i) : p[i];
		acc = wave_sum(acc);
	}
	__syncthreads();
	if (tid < 256 && (tid & 63) == 0) sh_t[tid >> 6] = acc;
	__syncthreads();
	if (tid == 0) {
		acc = sh_t[0];
		for (int q = 1; q < 4; q++) acc += sh_t[q];
	}
	return acc;
}
// the min / max fold of block_minmax over fpart[2 * i], fpart[2 * i + 1] as a 256-thread block does it; valid in thread 0
__device__ __forceinline__ void tail_minmax256(const float* fpart, int nb, float& lo, float& hi) {
	__shared__ float shl_t[4], shh_t[4];
	const int tid = threadIdx.x;
	lo = FLT_MAX;
	hi = -FLT_MAX;
	if (tid < 256) {
		for (int i = tid; i < nb; i += 256) {
			lo = fminf(lo, fpart[2 * i]);
			hi = fmaxf(hi, fpart[2 * i + 1]);
		}
		lo = wave_min(lo);
		hi = wave_max(hi);
	}
	__syncthreads();
	if (tid < 256 && (tid & 63) == 0) {
		shl_t[tid >> 6] = lo;
		shh_t[tid >> 6] = hi;
	}
	__syncthreads();
	if (tid == 0) {
		lo = shl_t[0];
		hi = shh_t[0];
		for (int q = 1; q < 4; q++) {
			lo = fminf(lo, shl_t[q]);
			hi = fmaxf(hi, shh_t[q]);
		}
	}
}

// per-thread strided sum over per-block partials with 8 loads in flight (a plain loop serialises one
// memory round trip per element); fixed order -> deterministic
__device__ __forceinline__ double strided_sum(const double* __restrict__ p, int nb) {
	double acc = 0.0;
	int i = threadIdx.x;
	const int S = blockDim.x;
	for (; i + 7 * S < nb; i += 8 * S) {
		const double v0 = p[i], v1 = p[i + S], v2 = p[i + 2 * S], v3 = p[i + 3 * S];
		const double v4 = p[i + 4 * S], v5 = p[i + 5 * S], v6 = p[i + 6 * S], v7 = p[i + 7 * S];
		acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
	}
	for (; i < nb; i += S) acc += p[i];
	return acc;
}

// Non-temporal 16-byte accesses for data a kernel touches exactly once (streams that nothing re-reads from the caches): on gfx950 they
// leave L2 / MALL to the operands that are re-read (stencil neighbours) -- ApplyMatrix 91 -> 69 us at 256^3 with A0 / Ai / flags loaded
// and dst stored this way.
__device__ __forceinline__ float4 ld_nt4(const float* p) {
	typedef float v4 __attribute__((ext_vector_type(4)));
	const v4 v = __builtin_nontemporal_load((const v4*)p);
	return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ld_nt4i(const int32_t* p) {
	typedef int v4 __attribute__((ext_vector_type(4)));
	const v4 v = __builtin_nontemporal_load((const v4*)p);
	return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt4(float* p, float4 a) {
	typedef float v4 __attribute__((ext_vector_type(4)));
	v4 o;
	o.x = a.x; o.y = a.y; o.z = a.z; o.w = a.w;
	__builtin_nontemporal_store(o, (v4*)p);
}

// ---- interpolation primitives, reference util/interpol.h -------------------------------------------------
struct Bi {
	int xi, yi, zi;
	float s0, s1, t0, t1, f0, f1;
};
// Positions are GLOBAL grid coordinates (z includes the slab offset), so a slab computes bit-identical weights to
// the undivided domain.  BUILD_INDEX, interpol.h:52-69.  `1.-s1` is an fp64 subtraction rounded to fp32 in the reference; for
// s1 in [0,1) that equals the fp32 subtraction (exact in fp64 when s1 >= 2^-29, both give 1 below), and
// outside that range the clamps overwrite the weights.  The fork clamps the upper side on px, not xi.
__device__ __forceinline__ Bi build_index(const Dim& d, float x, float y, float z) {
	Bi b;
	const float px = x - 0.5f, py = y - 0.5f, pz = z - 0.5f;
	b.xi = (int)px;
	b.yi = (int)py;
	b.zi = (int)pz;
	b.s1 = px - (float)b.xi;
	b.s0 = (float)(1. - (double)b.s1);
	b.t1 = py - (float)b.yi;
	b.t0 = (float)(1. - (double)b.t1);
	b.f1 = pz - (float)b.zi;
	b.f0 = (float)(1. - (double)b.f1);
	if (px < 0.f) { b.xi = 0; b.s0 = 1.f; b.s1 = 0.f; }
	if (py < 0.f) { b.yi = 0; b.t0 = 1.f; b.t1 = 0.f; }
	if (pz < 0.f) { b.zi = 0; b.f0 = 1.f; b.f1 = 0.f; }
	if (px >= (float)(d.sx - 1)) { b.xi = d.sx - 2; b.s0 = 0.f; b.s1 = 1.f; }
	if (py >= (float)(d.sy - 1)) { b.yi = d.sy - 2; b.t0 = 0.f; b.t1 = 1.f; }
	if (d.gsz > 1) { if (pz >= (float)(d.gsz - 1)) { b.zi = d.gsz - 2; b.f0 = 0.f; b.f1 = 1.f; } }
	b.zi = local_z(d, b.zi, 1);
	return b;
}
// shifted half of BUILD_INDEX_SHIFT, interpol.h:116-129 (upper clamp on the integer index)
__device__ __forceinline__ Bi build_index_shift(const Dim& d, float x, float y, float z) {
	Bi b;
	b.xi = (int)x;
	b.yi = (int)y;
	b.zi = (int)z;
	b.s1 = x - (float)b.xi;
	b.s0 = (float)(1. - (double)b.s1);
	b.t1 = y - (float)b.yi;
	b.t0 = (float)(1. - (double)b.t1);
	b.f1 = z - (float)b.zi;
	b.f0 = (float)(1. - (double)b.f1);
	if (x < 0.f) { b.xi = 0; b.s0 = 1.f; b.s1 = 0.f; }
	if (y < 0.f) { b.yi = 0; b.t0 = 1.f; b.t1 = 0.f; }
	if (z < 0.f) { b.zi = 0; b.f0 = 1.f; b.f1 = 0.f; }
	if (b.xi >= d.sx - 1) { b.xi = d.sx - 2; b.s0 = 0.f; b.s1 = 1.f; }
	if (b.yi >= d.sy - 1) { b.yi = d.sy - 2; b.t0 = 0.f; b.t1 = 1.f; }
	if (d.gsz > 1) { if (b.zi >= d.gsz - 1) { b.zi = d.gsz - 2; b.f0 = 0.f; b.f1 = 1.f; } }
	b.zi = local_z(d, b.zi, 1);
	return b;
}
// 8-corner gather with the reference's association order, interpol.h:77-80 / 90-93
__device__ __forceinline__ float tri8(const float* __restrict__ r, int64_t Y, int64_t Z, float t0, float t1, float s0,
                                      float s1, float f0, float f1) {
	const float a = (r[0] * t0 + r[Y] * t1) * s0 + (r[1] * t0 + r[1 + Y] * t1) * s1;
	const float b = (r[Z] * t0 + r[Y + Z] * t1) * s0 + (r[1 + Z] * t0 + r[1 + Y + Z] * t1) * s1;
	return a * f0 + b * f1;
}
// interpol<T> / interpolComponent<c> on one scalar plane
__device__ __forceinline__ float interpol1(const Dim& d, const float* __restrict__ data, float x, float y, float z) {
	const Bi b = build_index(d, x, y, z);
	const int64_t idx = (int64_t)b.xi + d.Y * b.yi + d.Z * b.zi;
	return tri8(data + idx, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, b.f0, b.f1);
}
// ---- cubic interpolation, util/interpolHigh.h (orderSpace = 2 of getInterpolatedHi, grid.h:153-159 / 271-286) -------------------
// cubicInterp<T> :22-39.  VEC = false: T = Real -- a2 and a3 are double expressions (3.0 * deltak - 2.0 * d0 - d1) rounded once;
// VEC = true: one component of T = Vec3 -- every scalar * vector product is rounded to fp32 (vectorbase.h:277-284) before the fp32
// vector sums.  The polynomial is fp32 either way, left to right.
template <bool VEC>
__device__ __forceinline__ float cubic_interp(float t, float p0, float p1, float p2, float p3) {
	const float d0 = (float)((double)(p2 - p0) * 0.5), d1 = (float)((double)(p3 - p1) * 0.5);
	const float dk = p2 - p1;
	float a2, a3;
	if (!VEC) {
		a2 = (float)(3.0 * (double)dk - 2.0 * (double)d0 - (double)d1);
		a3 = (float)(-2.0 * (double)dk + (double)d0 + (double)d1);
	} else {
		a2 = ((float)(3.0 * (double)dk) - (float)(2.0 * (double)d0)) - d1;
		a3 = ((float)(-2.0 * (double)dk) + d0) + d1;
	}
	const float sq = t * t, cu = sq * t;
	return a3 * cu + a2 * sq + d0 * t + p1;
}
// interpolCubic<T> / interpolCubic2D<T> :42-167 on one scalar plane: 4 x 4 (x 4) points around the cell of pos - 0.5; where that
// neighbourhood leaves the grid the reference falls back to the linear interpol().  Positions are global (z-slab window).
template <bool VEC>
__device__ __forceinline__ float interpol_cubic(const Dim& d, const float* __restrict__ data, float x, float y, float z) {
	const float px = x - 0.5f, py = y - 0.5f, pz = z - 0.5f;
	const int x1 = (int)px, y1 = (int)py, z1 = (int)pz;
	const int x0 = x1 - 1, x3 = x1 + 2, y0 = y1 - 1, y3 = y1 + 2, z0 = z1 - 1, z3 = z1 + 2;
	if (x0 < 0 || y0 < 0 || x3 >= d.sx || y3 >= d.sy || (d.is3d && (z0 < 0 || z3 >= d.gsz))) return interpol1(d, data, x, y, z);
	const float xi = px - (float)x1, yi = py - (float)y1, zi = pz - (float)z1;
	float planes[4];
	const int nzp = d.is3d ? 4 : 1;
#pragma unroll
	for (int c = 0; c < 4; c++) {
		if (c < nzp) {
			const float* base = data + (d.is3d ? d.Z * (int64_t)(z0 + c - d.zoff) : 0) + x0;
			float rows[4];
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const float* r = base + d.Y * (int64_t)(y0 + b);
				rows[b] = cubic_interp<VEC>(xi, r[0], r[1], r[2], r[3]);
			}
			planes[c] = cubic_interp<VEC>(yi, rows[0], rows[1], rows[2], rows[3]);
		}
	}
	return d.is3d ? cubic_interp<VEC>(zi, planes[0], planes[1], planes[2], planes[3]) : planes[0];
}
// interpolCubicMAC :169-176, component C: interpolCubic<Vec3>(pos + 0.5 e_C)[C]; 0 for the z component of a 2-D grid
template <int C>
__device__ __forceinline__ float interpol_cubic_mac(const Dim& d, const float* __restrict__ vel, float x, float y, float z) {
	if (C == 2 && !d.is3d) return 0.f;
	return interpol_cubic<true>(d, vel + (int64_t)C * d.n, C == 0 ? x + 0.5f : x, C == 1 ? y + 0.5f : y, C == 2 ? z + 0.5f : z);
}
// interpolMAC, interpol.h:131-164 (vel is SoA: x plane, y plane, z plane)
__device__ __forceinline__ void interpol_mac(const Dim& d, const float* __restrict__ vel, float x, float y, float z,
                                             float& ox, float& oy, float& oz) {
	const Bi b = build_index(d, x, y, z), s = build_index_shift(d, x, y, z);
	ox = tri8(vel + (((int64_t)b.zi * d.sy + b.yi) * d.sx + s.xi), d.Y, d.Z, b.t0, b.t1, s.s0, s.s1, b.f0, b.f1);
	oy = tri8(vel + d.n + (((int64_t)b.zi * d.sy + s.yi) * d.sx + b.xi), d.Y, d.Z, s.t0, s.t1, b.s0, b.s1, b.f0, b.f1);
	oz = tri8(vel + 2 * d.n + (((int64_t)s.zi * d.sy + b.yi) * d.sx + b.xi), d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, s.f0, s.f1);
}
// MACGrid samplers, grid.h:460-506.  0.5*/0.25* of an fp32 sum is exact, so fp32 arithmetic matches the
// reference's double-literal products.
__device__ __forceinline__ void get_centered(const Dim& d, const float* __restrict__ vel, int64_t idx, float& vx,
                                             float& vy, float& vz) {
	vx = 0.5f * (vel[idx] + vel[idx + 1]);
	vy = 0.5f * (vel[d.n + idx] + vel[d.n + idx + d.sx]);
	vz = 0.f;
	if (d.is3d) vz = 0.5f * (vel[2 * d.n + idx] + vel[2 * d.n + idx + d.Z]);
}
__device__ __forceinline__ void get_at_mac_x(const Dim& d, const float* __restrict__ vel, int64_t idx, float& vx,
                                             float& vy, float& vz) {
	const float* y = vel + d.n;
	const float* z = vel + 2 * d.n;
	vx = vel[idx];
	vy = 0.25f * (y[idx] + y[idx - 1] + y[idx + d.sx] + y[idx + d.sx - 1]);
	vz = 0.f;
	if (d.is3d) vz = 0.25f * (z[idx] + z[idx - 1] + z[idx + d.Z] + z[idx + d.Z - 1]);
}
__device__ __forceinline__ void get_at_mac_y(const Dim& d, const float* __restrict__ vel, int64_t idx, float& vx,
                                             float& vy, float& vz) {
	const float* x = vel;
	const float* z = vel + 2 * d.n;
	vx = 0.25f * (x[idx] + x[idx - d.sx] + x[idx + 1] + x[idx + 1 - d.sx]);
	vy = vel[d.n + idx];
	vz = 0.f;
	if (d.is3d) vz = 0.25f * (z[idx] + z[idx - d.sx] + z[idx + d.Z] + z[idx + d.Z - d.sx]);
}
__device__ __forceinline__ void get_at_mac_z(const Dim& d, const float* __restrict__ vel, int64_t idx, float& vx,
                                             float& vy, float& vz) {
	const float* x = vel;
	const float* y = vel + d.n;
	vx = 0.25f * (x[idx] + x[idx - d.Z] + x[idx + 1] + x[idx + 1 - d.Z]);
	vy = 0.25f * (y[idx] + y[idx - d.Z] + y[idx + d.sx] + y[idx + d.sx - d.Z]);
	vz = vel[2 * d.n + idx];
}


// ---- APIC (plugin/apic.cpp:29-33, 119-123): face index f = (IndexInt)pos, centre index c = (IndexInt)(pos - 0.5) with the
// subtraction in double (double literal) and truncation toward zero; wf = clamp(pos - f, 0, 1) in fp32,
// wc = clamp(Real(pos - c - 0.5), 0, 1) with the "- 0.5" in double.  apic_face: face COMP (0 u, 1 v, 2 w) -> flat base
// index, position of the base node, per-axis weight pairs {1 - w, w}.
struct ApicFace {
	int64_t gidx;
	float gpos[3];
	float W[3][2];
};
__device__ __forceinline__ float apic_clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
template <int COMP>
__device__ __forceinline__ ApicFace apic_face(const Dim& d, float px, float py, float pz) {
	const float P[3] = {px, py, pz};
	ApicFace a;
	int64_t b[3];
#pragma unroll
	for (int q = 0; q < 3; q++) {
		if (q == COMP) {
			const int64_t f = (int64_t)P[q];
			b[q] = f;
			a.gpos[q] = (float)f;
			const float w = apic_clamp01(P[q] - (float)f);
			a.W[q][0] = 1.f - w;
			a.W[q][1] = w;
		} else {
			const int64_t c = (int64_t)((double)P[q] - 0.5);
			b[q] = c;
			a.gpos[q] = (float)((double)c + 0.5);
			const float w = apic_clamp01((float)((double)(P[q] - (float)c) - 0.5));
			a.W[q][0] = 1.f - w;
			a.W[q][1] = w;
		}
	}
	a.gidx = b[0] + b[1] * d.Y + b[2] * d.Z;
	return a;
}
}  // namespace mf
