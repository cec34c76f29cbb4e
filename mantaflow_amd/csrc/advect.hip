// advect.hip -- semi-Lagrangian / MacCormack advection on MAC grids (gfx950).
// Reference: source/plugin/advection.cpp (cited per kernel), util/interpol.h, grid.h:460-506.
// All kernels are pure gathers (one thread per cell, coalesced along x); the velocity field and the advected
// field are SoA planes so that a wave's 64 consecutive cells read 256 contiguous bytes per plane.
#include "common.h"
#include <float.h>
#include <stdlib.h>

using namespace mf;

// blocks are dealt round-robin over the 8 XCDs: give every XCD one contiguous range of cells (a z-slab), so that the planes a
// gather reaches stay in that XCD's L2 instead of being fetched from HBM by all eight (xcd_swizzle: speed only)
#define CELL_IJK(d)                                                               \
	const int64_t idx = xcd_swizzle((int)blockIdx.x, (int)gridDim.x) * (int64_t)BLOCK + threadIdx.x; \
	if (idx >= (d).n) return;                                                     \
	const int i = (int)(idx % (d).sx);                                            \
	const int j = (int)((idx / (d).sx) % (d).sy);                                 \
	const int k = (int)(idx / ((int64_t)(d).sx * (d).sy));                        \
	(void)i; (void)j; (void)k;
#define INTERIOR(d) (i >= 1 && i < (d).sx - 1 && j >= 1 && j < (d).sy - 1 && (!(d).is3d || (k >= 1 && k < (d).sz - 1)))
static inline unsigned nblk(const Dim& d) { return (unsigned)((d.n + BLOCK - 1) / BLOCK); }

struct __attribute__((packed, aligned(4))) F2u {
	float a, b;
};
__device__ __forceinline__ F2u ld2(const float* __restrict__ p) { return *(const F2u*)p; }   // 8-byte load from a 4-byte aligned address

// SemiLagrange<T>, advection.cpp:25-42.  NCOMP scalar planes (1 = Real, 3 = cell-centred Vec3); OS = orderSpace (1 linear, 2 cubic:
// a kernel of its own, so that the common linear path keeps its registers)
template <int NCOMP, int OS>
__global__ void __launch_bounds__(BLOCK)
k_semi_lagrange(Dim d, const float* __restrict__ vel, float* __restrict__ dst, const float* __restrict__ src, float dt, int orderTrace) {
	CELL_IJK(d)
	if (!INTERIOR(d)) {
		// KERNEL(bnd=1) leaves the boundary of a fresh (cleared) temp grid alone: written here, so that the caller need not clear dst
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = 0.f;
		return;
	}
	float vx, vy, vz, px, py, pz;
	get_centered(d, vel, idx, vx, vy, vz);
	const float cx = (float)i + 0.5f, cy = (float)j + 0.5f, cz = (float)(k + d.zoff) + 0.5f;
	if (orderTrace == 1) {
		px = cx - vx * dt;
		py = cy - vy * dt;
		pz = cz - vz * dt;
	} else {
		// p1 = p0 - getCentered*dt*0.5 ; p2 = p0 - vel.getInterpolated(p1)*dt   (explicit midpoint)
		const float p1x = cx - (vx * dt) * 0.5f, p1y = cy - (vy * dt) * 0.5f, p1z = cz - (vz * dt) * 0.5f;
		float ux, uy, uz;
		interpol_mac(d, vel, p1x, p1y, p1z, ux, uy, uz);
		px = cx - ux * dt;
		py = cy - uy * dt;
		pz = cz - uz * dt;
	}
	if (OS == 2) {
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = interpol_cubic<(NCOMP == 3)>(d, src + c * d.n, px, py, pz);
		return;
	}
	const Bi b = build_index(d, px, py, pz);
	const int64_t base = (int64_t)b.xi + d.Y * b.yi + d.Z * b.zi;
#pragma unroll
	for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = tri8(src + c * d.n + base, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, b.f0, b.f1);
}

// The same, marching along z: a thread owns SLZ consecutive planes of one (i, j) column, software-pipelined -- the velocity loads of
// plane k + 1 are in flight while the eight source taps of plane k are gathered (their addresses depend on the velocity of plane k).
// The one-cell kernel above is bound by two dependent memory round trips per cell at full occupancy (56 VGPRs, 32 waves per CU:
// 2048 cells in flight per CU, ~3.4 us per cell => ~110 us at 256^3); here the two round trips of neighbouring planes overlap, and the
// +z velocity sample of plane k is the -z sample of plane k + 1.  Same arithmetic per cell.  3D, orderTrace 1, linear interpolation.
// 256^3 MacCormack (two of these + the fused correction): 457 -> 419 us.  (A three-stage version -- velocity of k + 2, taps of k + 1,
// finish k, every load unconditional -- has 8-11 loads in flight per thread and is no faster, 433 us; SLZ = 8: 422 us.)
constexpr int SLZ = 4;
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_semi_lagrange_zmarch(Dim d, const float* __restrict__ vel, float* __restrict__ dst, const float* __restrict__ src, float dt) {
	const int64_t plane = (int64_t)d.sx * d.sy;
	const int64_t q = xcd_swizzle((int)blockIdx.x, (int)gridDim.x) * (int64_t)BLOCK + threadIdx.x;
	const int64_t g = q / plane;                       // group of SLZ planes
	const int64_t ij = q - g * plane;
	const int i = (int)(ij % d.sx), j = (int)(ij / d.sx);
	const int k0 = (int)g * SLZ;
	if (k0 >= d.sz) return;
	if (i < 1 || i >= d.sx - 1 || j < 1 || j >= d.sy - 1) {
		// boundary columns: the zeros of a cleared temp grid (the caller need not clear dst)
		for (int t = 0; t < SLZ && k0 + t < d.sz; t++)
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dst[c * d.n + ij + (int64_t)(k0 + t) * d.Z] = 0.f;
		return;
	}
	const float* vx_ = vel;
	const float* vy_ = vel + d.n;
	const float* vz_ = vel + 2 * d.n;
	const float cx = (float)i + 0.5f, cy = (float)j + 0.5f;
	int64_t idx = ij + (int64_t)k0 * d.Z;
	// samples of the first plane
	float ax = vx_[idx], bx = vx_[idx + 1], ay = vy_[idx], by = vy_[idx + d.sx], az = vz_[idx], bz = (k0 + 1 < d.sz) ? vz_[idx + d.Z] : 0.f;
#pragma unroll
	for (int t = 0; t < SLZ; t++) {
		const int k = k0 + t;
		if (k >= d.sz) break;
		// next plane's samples (its -z sample is this plane's +z sample)
		const bool more = (t + 1 < SLZ) && (k + 1 < d.sz);
		const int64_t nidx = idx + d.Z;
		float nax = 0.f, nbx = 0.f, nay = 0.f, nby = 0.f, nbz = 0.f;
		if (more) {
			nax = vx_[nidx];
			nbx = vx_[nidx + 1];
			nay = vy_[nidx];
			nby = vy_[nidx + d.sx];
			nbz = (k + 2 < d.sz) ? vz_[nidx + d.Z] : 0.f;
		}
		if (k < 1 || k >= d.sz - 1) {
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = 0.f;      // first / last plane: boundary zeros
		} else {
			const float vx = 0.5f * (ax + bx), vy = 0.5f * (ay + by), vz = 0.5f * (az + bz);      // getCentered
			const float cz = (float)(k + d.zoff) + 0.5f;
			const float px = cx - vx * dt, py = cy - vy * dt, pz = cz - vz * dt;
			const Bi b = build_index(d, px, py, pz);
			const int64_t base = (int64_t)b.xi + d.Y * b.yi + d.Z * b.zi;
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = tri8(src + c * d.n + base, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, b.f0, b.f1);
		}
		ax = nax; bx = nbx; ay = nay; by = nby; az = bz; bz = nbz;
		idx = nidx;
	}
}

// MACGrid::getInterpolatedComponentHi<C>, grid.h:280-286
template <int C, int OS>
__device__ __forceinline__ float mac_component_hi(const Dim& d, const float* __restrict__ src, float x, float y, float z) {
	return OS == 1 ? interpol1(d, src + (int64_t)C * d.n, x, y, z) : interpol_cubic_mac<C>(d, src, x, y, z);
}
// SemiLagrangeMAC, advection.cpp:45-78
// (Round 3, built and dropped: a workgroup copying the velocity of its 32 x 4 x 4 tile + halo into LDS and forming getAtMACX/Y/Z from
// there -- 7 coalesced loads per cell instead of 27 neighbour loads -- is bit-exact and SLOWER, 355 vs 293 us here and 563 vs 469 us
// in the fused clamp: the neighbour loads were L1 hits all along, the kernels are bound by the traced, divergent gathers and by the
// latency of the position -> address -> value chain, and the tile costs occupancy, a barrier and index arithmetic.)
template <int OS>
__global__ void __launch_bounds__(BLOCK)
k_semi_lagrange_mac(Dim d, const float* __restrict__ vel, float* __restrict__ dst, const float* __restrict__ src, float dt, int orderTrace) {
	CELL_IJK(d)
	if (!INTERIOR(d)) {
		dst[idx] = dst[d.n + idx] = dst[2 * d.n + idx] = 0.f;      // boundary zeros of a cleared temp grid
		return;
	}
	const float cx = (float)i + 0.5f, cy = (float)j + 0.5f, cz = (float)(k + d.zoff) + 0.5f;
	const int kg = k + d.zoff;
	float vx, vy, vz, rx, ry, rz;
	if (orderTrace == 1) {
		get_at_mac_x(d, vel, idx, vx, vy, vz);
		rx = mac_component_hi<0, OS>(d, src, cx - vx * dt, cy - vy * dt, cz - vz * dt);
		get_at_mac_y(d, vel, idx, vx, vy, vz);
		ry = mac_component_hi<1, OS>(d, src, cx - vx * dt, cy - vy * dt, cz - vz * dt);
		get_at_mac_z(d, vel, idx, vx, vy, vz);
		rz = mac_component_hi<2, OS>(d, src, cx - vx * dt, cy - vy * dt, cz - vz * dt);
	} else {
		// the midpoint variant traces with `src`, not `vel` (advection.cpp:62-72)
		float ux, uy, uz;
		get_at_mac_x(d, src, idx, vx, vy, vz);
		interpol_mac(d, src, (float)i - (vx * dt) * 0.5f, cy - (vy * dt) * 0.5f, cz - (vz * dt) * 0.5f, ux, uy, uz);
		rx = mac_component_hi<0, OS>(d, src, cx - ux * dt, cy - uy * dt, cz - uz * dt);
		get_at_mac_y(d, src, idx, vx, vy, vz);
		interpol_mac(d, src, cx - (vx * dt) * 0.5f, (float)j - (vy * dt) * 0.5f, cz - (vz * dt) * 0.5f, ux, uy, uz);
		ry = mac_component_hi<1, OS>(d, src, cx - ux * dt, cy - uy * dt, cz - uz * dt);
		get_at_mac_z(d, src, idx, vx, vy, vz);
		interpol_mac(d, src, cx - (vx * dt) * 0.5f, cy - (vy * dt) * 0.5f, (float)kg - (vz * dt) * 0.5f, ux, uy, uz);
		rz = mac_component_hi<2, OS>(d, src, cx - ux * dt, cy - uy * dt, cz - uz * dt);
	}
	dst[idx] = rx;
	dst[d.n + idx] = ry;
	dst[2 * d.n + idx] = rz;
}

// MacCormackCorrect<T>, advection.cpp:82-92 (KERNEL(idx): every cell)
//   Real: `dst += strength*0.5*(old-bwd)` is an fp64 compound assignment (0.5 is a double literal)
//   Vec3: `S2 * Vector3D` rounds each product to fp32 first (vectorbase.h:282-284), then an fp32 add
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_maccormack_correct(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ old,
                     const float* __restrict__ fwd, const float* __restrict__ bwd, float strength) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= d.n) return;
	const bool fl = flags[idx] & MF_FLUID;
	const double sh = (double)strength * 0.5;
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		const int64_t q = c * d.n + idx;
		float v = fwd[q];
		if (fl) {
			const float df = old[q] - bwd[q];
			if (NCOMP == 1)
				v = (float)((double)v + sh * (double)df);
			else
				v = v + (float)(sh * (double)df);
		}
		dst[q] = v;
	}
}
// MacCormackCorrectMAC<Vec3>(isMAC = true), advection.cpp:95-116 (KERNEL(): every cell)
__global__ void __launch_bounds__(BLOCK)
k_maccormack_correct_mac(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ old,
                         const float* __restrict__ fwd, const float* __restrict__ bwd, float strength) {
	CELL_IJK(d)
	bool s0 = false, s1 = false, s2 = false;
	if (!(flags[idx] & MF_FLUID)) s0 = s1 = s2 = true;
	if ((i > 0) && !(flags[idx - 1] & MF_FLUID)) s0 = true;
	if ((j > 0) && !(flags[idx - d.Y] & MF_FLUID)) s1 = true;
	if ((k > 0) && !(flags[idx - d.Z] & MF_FLUID)) s2 = true;
	const double sh = (double)strength * 0.5;
	const bool skip[3] = {s0, s1, s2};
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const int64_t q = c * d.n + idx;
		const float f = fwd[q];
		dst[q] = skip[c] ? f : (float)((double)f + sh * (double)(old[q] - bwd[q]));
	}
}

__device__ __forceinline__ bool checkflag(int f) { return (f & (MF_FLUID | MF_EMPTY)) != 0; }  // advection.cpp:140
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// MacCormackClamp<T>, advection.cpp:242-268 + doClampComponent :145-187
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_maccormack_clamp(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ dst,
                   const float* __restrict__ orig, const float* __restrict__ fwd, float dt, int clampMode) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	float vx, vy, vz;
	get_centered(d, vel, idx, vx, vy, vz);
	vx *= dt;
	vy *= dt;
	vz *= dt;
	float dval[NCOMP], fw[NCOMP], minv[NCOMP], maxv[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		dval[c] = dst[c * d.n + idx];
		fw[c] = fwd[c * d.n + idx];
		minv[c] = FLT_MAX;
		maxv[c] = -FLT_MAX;
	}
	bool haveFl = false;
	const int gx = d.sx - 1, gy = d.sy - 1, gz = d.gsz - 1;  // gridUpper = size - 1 (global z extent)
	const int kg = k + d.zoff;
	const int numPos = (clampMode == 1) ? 2 : 1;
	for (int l = 0; l < numPos; l++) {
		const float sg = l == 0 ? -1.f : 1.f;
		const int cxp = (int)((float)i + sg * vx), cyp = (int)((float)j + sg * vy), czp = (int)((float)kg + sg * vz);
		const int i0 = clampi(cxp, 0, gx - 1), j0 = clampi(cyp, 0, gy - 1);
		const int k0 = local_z(d, clampi(czp, 0, d.is3d ? (gz - 1) : 1), 1);
		const int nz = d.is3d ? 2 : 1;
		for (int dz = 0; dz < nz; dz++)
			for (int dy = 0; dy < 2; dy++)
				for (int dx = 0; dx < 2; dx++) {
					const int64_t q = (int64_t)(i0 + dx) + d.Y * (j0 + dy) + d.Z * (k0 + dz);
					if (checkflag(flags[q])) {
#pragma unroll
						for (int c = 0; c < NCOMP; c++) {
							const float v = orig[c * d.n + q];
							if (v < minv[c]) minv[c] = v;
							if (v > maxv[c]) maxv[c] = v;
						}
						haveFl = true;
					}
				}
	}
	if (!haveFl) {
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
	} else if (clampMode == 1) {
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dval[c] = dval[c] < minv[c] ? minv[c] : (dval[c] > maxv[c] ? maxv[c] : dval[c]);
	} else {
		bool outside = false;
#pragma unroll
		for (int c = 0; c < NCOMP; c++) outside |= (dval[c] < minv[c]) | (dval[c] > maxv[c]);
		if (outside) {
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
		}
	}
	if (clampMode == 1) {
		const float cx = (float)i + 0.5f, cy = (float)j + 0.5f, cz = (float)kg + 0.5f;
		const int fx = (int)(cx - vx), fy = (int)(cy - vy), fz = (int)(cz - vz);
		const int bx = (int)(cx + vx), by = (int)(cy + vy), bz = (int)(cz + vz);
		bool bad = fx < 0 || fy < 0 || fz < 0 || bx < 0 || by < 0 || bz < 0 || fx > gx || fy > gy || ((fz > gz) && d.is3d) ||
		           bx > gx || by > gy || ((bz > gz) && d.is3d);
		if (!bad)
			bad = (flags[(int64_t)fx + d.Y * fy + d.Z * local_z(d, fz, 0)] & MF_OBSTACLE) ||
			      (flags[(int64_t)bx + d.Y * by + d.Z * local_z(d, bz, 0)] & MF_OBSTACLE);
		if (bad) {
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
		}
	}
#pragma unroll
	for (int c = 0; c < NCOMP; c++) dst[c * d.n + idx] = dval[c];
}

// doClampComponentMAC<c>, advection.cpp:192-236
__device__ __forceinline__ float clamp_component_mac(const Dim& d, int c, const int32_t* __restrict__ flags, float dstv,
                                                     const float* __restrict__ orig, float fwdv, int i, int j, int k,
                                                     float vx, float vy, float vz, int clampMode) {
	float minv = FLT_MAX, maxv = -FLT_MAX;
	const int64_t o = (int64_t)i + d.Y * j + d.Z * k;
	const int64_t nbo = o - (c == 0 ? 1 : (c == 1 ? d.Y : d.Z));
	if (clampMode == 2 && !(checkflag(flags[o]) && checkflag(flags[nbo]))) return fwdv;
	const int gx = d.sx - 1, gy = d.sy - 1, gz = d.gsz - 1;
	const int kg = k + d.zoff;
	const float* oc = orig + c * d.n;
	const int numPos = (clampMode == 1) ? 2 : 1;
	for (int l = 0; l < numPos; l++) {
		const float sg = l == 0 ? -1.f : 1.f;
		const int cxp = (int)((float)i + sg * vx), cyp = (int)((float)j + sg * vy), czp = (int)((float)kg + sg * vz);
		const int i0 = clampi(cxp, 0, gx - 1), j0 = clampi(cyp, 0, gy - 1);
		const int k0 = local_z(d, clampi(czp, 0, d.is3d ? (gz - 1) : 0), 1);
		const int nz = d.is3d ? 2 : 1;
		for (int dz = 0; dz < nz; dz++)
			for (int dy = 0; dy < 2; dy++) {
				const float* rp = oc + (int64_t)i0 + d.Y * (j0 + dy) + d.Z * (k0 + dz);
				F2u v2;
				v2.a = rp[0];
				v2.b = rp[1];
				if (v2.a < minv) minv = v2.a;
				if (v2.a > maxv) maxv = v2.a;
				if (v2.b < minv) minv = v2.b;
				if (v2.b > maxv) maxv = v2.b;
			}
	}
	if (clampMode == 1)
		dstv = dstv < minv ? minv : (dstv > maxv ? maxv : dstv);
	else if ((dstv < minv) | (dstv > maxv))
		dstv = fwdv;
	return dstv;
}
// MacCormackClampMAC, advection.cpp:271-288
__global__ void __launch_bounds__(BLOCK)
k_maccormack_clamp_mac(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ dst,
                       const float* __restrict__ orig, const float* __restrict__ fwd, float dt, int clampMode) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	float vx, vy, vz;
	get_at_mac_x(d, vel, idx, vx, vy, vz);
	const float rx = clamp_component_mac(d, 0, flags, dst[idx], orig, fwd[idx], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
	get_at_mac_y(d, vel, idx, vx, vy, vz);
	const float ry = clamp_component_mac(d, 1, flags, dst[d.n + idx], orig, fwd[d.n + idx], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
	float rz = dst[2 * d.n + idx];
	if (d.is3d) {
		get_at_mac_z(d, vel, idx, vx, vy, vz);
		rz = clamp_component_mac(d, 2, flags, rz, orig, fwd[2 * d.n + idx], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
	}
	dst[idx] = rx;
	dst[d.n + idx] = ry;
	dst[2 * d.n + idx] = rz;
}


// =========================================================================================================
// MacCormack's correction fused into its clamp: the clamp only looks at the corrected value of its own cell, so the intermediate
// grid of the reference's two kernels (and one launch) disappears -- 52 + 64 -> 88 B per cell for MAC grids, 20 + 32 -> 36 for Real.
// One cell per thread like the kernels above: a four-cells-per-thread variant of these gather kernels (16-byte velocity loads, 2-D
// index instead of the 64-bit divisions) was measured SLOWER on the MI355X (semi-Lagrange MAC 470 vs 276 us at 256^3): the gathers
// are latency-bound, and a quarter of the threads at 132 VGPRs leaves too few loads in flight.
// =========================================================================================================
// doClampComponent + the clampMode 1 trace test of MacCormackClamp<T> for one cell (advection.cpp:145-187, 250-264): dval in / out
template <int NCOMP>
__device__ __forceinline__ void clamp_cell(const Dim& d, const int32_t* __restrict__ flags, const float* __restrict__ orig, int i, int j, int k,
                                           float vx, float vy, float vz, float dval[NCOMP], const float fw[NCOMP], int clampMode) {
	float minv[NCOMP], maxv[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		minv[c] = FLT_MAX;
		maxv[c] = -FLT_MAX;
	}
	bool haveFl = false;
	const int gx = d.sx - 1, gy = d.sy - 1, gz = d.gsz - 1;  // gridUpper = size - 1 (global z extent)
	const int kg = k + d.zoff;
	const int numPos = (clampMode == 1) ? 2 : 1;
	for (int l = 0; l < numPos; l++) {
		const float sg = l == 0 ? -1.f : 1.f;
		const int cxp = (int)((float)i + sg * vx), cyp = (int)((float)j + sg * vy), czp = (int)((float)kg + sg * vz);
		const int a0 = clampi(cxp, 0, gx - 1), b0 = clampi(cyp, 0, gy - 1);
		const int c0 = local_z(d, clampi(czp, 0, d.is3d ? (gz - 1) : 1), 1);
		const int nz = d.is3d ? 2 : 1;
		for (int dz = 0; dz < nz; dz++)
			for (int dy = 0; dy < 2; dy++) {
				const int64_t q = (int64_t)a0 + d.Y * (b0 + dy) + d.Z * (c0 + dz);
				const bool ok0 = checkflag(flags[q]), ok1 = checkflag(flags[q + 1]);
#pragma unroll
				for (int c = 0; c < NCOMP; c++) {
					F2u o2;
					o2.a = orig[c * d.n + q];
					o2.b = orig[c * d.n + q + 1];
					if (ok0) {
						if (o2.a < minv[c]) minv[c] = o2.a;
						if (o2.a > maxv[c]) maxv[c] = o2.a;
					}
					if (ok1) {
						if (o2.b < minv[c]) minv[c] = o2.b;
						if (o2.b > maxv[c]) maxv[c] = o2.b;
					}
				}
				haveFl = haveFl || ok0 || ok1;
			}
	}
	if (!haveFl) {
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
	} else if (clampMode == 1) {
#pragma unroll
		for (int c = 0; c < NCOMP; c++) dval[c] = dval[c] < minv[c] ? minv[c] : (dval[c] > maxv[c] ? maxv[c] : dval[c]);
	} else {
		bool outside = false;
#pragma unroll
		for (int c = 0; c < NCOMP; c++) outside |= (dval[c] < minv[c]) | (dval[c] > maxv[c]);
		if (outside) {
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
		}
	}
	if (clampMode == 1) {
		const float cx = (float)i + 0.5f, cy = (float)j + 0.5f, cz = (float)kg + 0.5f;
		const int fx = (int)(cx - vx), fy = (int)(cy - vy), fz = (int)(cz - vz);
		const int bx = (int)(cx + vx), by = (int)(cy + vy), bz = (int)(cz + vz);
		bool bad = fx < 0 || fy < 0 || fz < 0 || bx < 0 || by < 0 || bz < 0 || fx > gx || fy > gy || ((fz > gz) && d.is3d) ||
		           bx > gx || by > gy || ((bz > gz) && d.is3d);
		if (!bad)
			bad = (flags[(int64_t)fx + d.Y * fy + d.Z * local_z(d, fz, 0)] & MF_OBSTACLE) ||
			      (flags[(int64_t)bx + d.Y * by + d.Z * local_z(d, bz, 0)] & MF_OBSTACLE);
		if (bad) {
#pragma unroll
			for (int c = 0; c < NCOMP; c++) dval[c] = fw[c];
		}
	}
}

#ifndef ADV_NT
#define ADV_NT 1
#endif
// MacCormackCorrect<T> + MacCormackClamp<T> (advection.cpp:82-92, 145-187, 242-268) in one pass over the grid
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_mc_correct_clamp(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ dst,
                   const float* __restrict__ orig, const float* __restrict__ fwd, const float* __restrict__ bwd, float strength, float dt,
                   int clampMode) {
	CELL_IJK(d)
	const bool fl = flags[idx] & MF_FLUID;
	const double sh = (double)strength * 0.5;
	float fw[NCOMP], dv[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		const int64_t q = c * d.n + idx;
		fw[c] = ADV_NT ? __builtin_nontemporal_load(fwd + q) : fwd[q];      // fwd / bwd: read here once and never again
		float v = fw[c];
		if (fl) {
			const float df = orig[q] - (ADV_NT ? __builtin_nontemporal_load(bwd + q) : bwd[q]);
			if (NCOMP == 1) v = (float)((double)v + sh * (double)df);
			else v = v + (float)(sh * (double)df);
		}
		dv[c] = v;
	}
	if (INTERIOR(d)) {
		float vx, vy, vz;
		get_centered(d, vel, idx, vx, vy, vz);
		clamp_cell<NCOMP>(d, flags, orig, i, j, k, vx * dt, vy * dt, vz * dt, dv, fw, clampMode);
	}
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		if (ADV_NT) __builtin_nontemporal_store(dv[c], dst + c * d.n + idx);
		else dst[c * d.n + idx] = dv[c];
	}
}

// MacCormackCorrectMAC + MacCormackClampMAC (advection.cpp:95-116, 192-236, 271-288) in one pass
__global__ void __launch_bounds__(BLOCK)
k_mc_correct_clamp_mac(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ dst,
                       const float* __restrict__ orig, const float* __restrict__ fwd, const float* __restrict__ bwd, float strength, float dt,
                       int clampMode) {
	CELL_IJK(d)
	bool s0 = false, s1 = false, s2 = false;
	if (!(flags[idx] & MF_FLUID)) s0 = s1 = s2 = true;
	if ((i > 0) && !(flags[idx - 1] & MF_FLUID)) s0 = true;
	if ((j > 0) && !(flags[idx - d.Y] & MF_FLUID)) s1 = true;
	if ((k > 0) && !(flags[idx - d.Z] & MF_FLUID)) s2 = true;
	const double sh = (double)strength * 0.5;
	const bool skip[3] = {s0, s1, s2};
	float fw[3], dv[3];
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const int64_t q = c * d.n + idx;
		fw[c] = ADV_NT ? __builtin_nontemporal_load(fwd + q) : fwd[q];
		dv[c] = skip[c] ? fw[c] : (float)((double)fw[c] + sh * (double)(orig[q] - (ADV_NT ? __builtin_nontemporal_load(bwd + q) : bwd[q])));
	}
	if (INTERIOR(d)) {
		float vx, vy, vz;
		get_at_mac_x(d, vel, idx, vx, vy, vz);
		dv[0] = clamp_component_mac(d, 0, flags, dv[0], orig, fw[0], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
		get_at_mac_y(d, vel, idx, vx, vy, vz);
		dv[1] = clamp_component_mac(d, 1, flags, dv[1], orig, fw[1], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
		if (d.is3d) {
			get_at_mac_z(d, vel, idx, vx, vy, vz);
			dv[2] = clamp_component_mac(d, 2, flags, dv[2], orig, fw[2], i, j, k, vx * dt, vy * dt, vz * dt, clampMode);
		}
	}
#pragma unroll
	for (int c = 0; c < 3; c++) {
		if (ADV_NT) __builtin_nontemporal_store(dv[c], dst + c * d.n + idx);
		else dst[c * d.n + idx] = dv[c];
	}
}

// extrapolateVelConvectiveBC + getBulkVel, advection.cpp:327-382 (KERNEL(): every cell; work only in outflow cells)
__device__ __forceinline__ bool inb(const Dim& d, int i, int j, int k) { return i >= 0 && j >= 0 && k >= 0 && i < d.sx && j < d.sy && k < d.sz; }
__global__ void __launch_bounds__(BLOCK)
k_outflow_extrapolate(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ velDst,
                      const float* __restrict__ velPrev, float timeStep) {
	CELL_IJK(d)
	if (!(flags[idx] & MF_OUTFLOW)) return;
	const int64_t n = d.n;
	float avg[3] = {0.f, 0.f, 0.f};
	int count = 0;
	const int nmax = d.is3d ? 1 : 0;
	for (int nn = -nmax; nn <= nmax; nn++)
		for (int m = -1; m <= 1; m++)
			for (int l = -1; l <= 1; l++)
				if (inb(d, i + l, j + m, k + nn)) {
					const int64_t q = (int64_t)(i + l) + d.Y * (j + m) + d.Z * (k + nn);
					if (flags[q] & (MF_FLUID | MF_OUTFLOW)) {
						avg[0] += vel[q];
						avg[1] += vel[n + q];
						avg[2] += vel[2 * n + q];
						count++;
					}
				}
	if (count > 0) {
		const float fc = (float)count;
		avg[0] = avg[0] / fc;
		avg[1] = avg[1] / fc;
		avg[2] = avg[2] / fc;
	}
	const int dim = d.is3d ? 3 : 2;
	int cnt = 0;
	float acc[3] = {velDst[idx], velDst[n + idx], velDst[2 * n + idx]};
	const float cur[3] = {vel[idx], vel[n + idx], vel[2 * n + idx]};
	const float prv[3] = {velPrev[idx], velPrev[n + idx], velPrev[2 * n + idx]};
	for (int c = 0; c < dim; c++) {
		int low[3] = {i, j, k}, up[3] = {i, j, k}, flLow[3] = {i, j, k}, flUp[3] = {i, j, k};
		const float factor = timeStep * (1.0f > avg[c] ? 1.0f : avg[c]);  // max((Real)1.0, bulkVel[c])
		low[c] = flLow[c] = low[c] - 1;
		up[c] = flUp[c] = up[c] + 1;
		for (int dd = 0; dd < 2; dd++) {
			const bool eL = inb(d, flLow[0], flLow[1], flLow[2]) && (flags[(int64_t)flLow[0] + d.Y * flLow[1] + d.Z * flLow[2]] & MF_FLUID);
			const bool eU = inb(d, flUp[0], flUp[1], flUp[2]) && (flags[(int64_t)flUp[0] + d.Y * flUp[1] + d.Z * flUp[2]] & MF_FLUID);
			if (eL || eU) {
				if (eL) {
					const int64_t q = (int64_t)low[0] + d.Y * low[1] + d.Z * low[2];
					for (int e = 0; e < 3; e++) acc[e] += ((cur[e] - prv[e]) / factor) + vel[e * n + q];
					cnt++;
				}
				if (eU) {
					const int64_t q = (int64_t)up[0] + d.Y * up[1] + d.Z * up[2];
					for (int e = 0; e < 3; e++) acc[e] += ((cur[e] - prv[e]) / factor) + vel[e * n + q];
					cnt++;
				}
				break;
			}
			flLow[c]--;
			flUp[c]++;
		}
	}
	if (cnt > 0) {
		const float fc = (float)cnt;
		acc[0] /= fc;
		acc[1] /= fc;
		acc[2] /= fc;
	}
	velDst[idx] = acc[0];
	velDst[n + idx] = acc[1];
	velDst[2 * n + idx] = acc[2];
}
// copyChangedVels, advection.cpp:385
__global__ void __launch_bounds__(BLOCK)
k_copy_changed_vels(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ velDst, float* __restrict__ vel) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= d.n) return;
	if (flags[idx] & MF_OUTFLOW) {
		vel[idx] = velDst[idx];
		vel[d.n + idx] = velDst[d.n + idx];
		vel[2 * d.n + idx] = velDst[2 * d.n + idx];
	}
}

static bool slz_off() {
	static const bool off = getenv("MF_SL_NOZMARCH") != nullptr;      // the one-cell-per-thread kernels, for comparison
	return off;
}

extern "C" {

int mf_semi_lagrange_real(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt, int orderTrace, int orderSpace,
                          void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (orderTrace != 1 && orderTrace != 2) return fail("Unknown backtracing order %d", orderTrace);
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order %d", orderSpace);
	const Dim d = mkdim(sx, sy, sz);
	if (orderSpace == 2)
		hipLaunchKernelGGL((k_semi_lagrange<1, 2>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	else
		if (d.is3d && orderTrace == 1 && !slz_off()) {
		const int64_t nthreads = (int64_t)d.sx * d.sy * ((d.sz + SLZ - 1) / SLZ);
		hipLaunchKernelGGL((k_semi_lagrange_zmarch<1>), dim3((unsigned)((nthreads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt);
	} else
		hipLaunchKernelGGL((k_semi_lagrange<1, 1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_semi_lagrange_vec3(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt, int orderTrace, int orderSpace,
                          void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (orderTrace != 1 && orderTrace != 2) return fail("Unknown backtracing order %d", orderTrace);
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order %d", orderSpace);
	const Dim d = mkdim(sx, sy, sz);
	if (orderSpace == 2)
		hipLaunchKernelGGL((k_semi_lagrange<3, 2>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	else
		if (d.is3d && orderTrace == 1 && !slz_off()) {
		const int64_t nthreads = (int64_t)d.sx * d.sy * ((d.sz + SLZ - 1) / SLZ);
		hipLaunchKernelGGL((k_semi_lagrange_zmarch<3>), dim3((unsigned)((nthreads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt);
	} else
		hipLaunchKernelGGL((k_semi_lagrange<3, 1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_semi_lagrange_mac(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt, int orderTrace, int orderSpace,
                         void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (orderTrace != 1 && orderTrace != 2) return fail("Unknown backtracing order %d", orderTrace);
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order %d", orderSpace);
	const Dim d = mkdim(sx, sy, sz);
	if (orderSpace == 2)
		hipLaunchKernelGGL((k_semi_lagrange_mac<2>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	else
		hipLaunchKernelGGL((k_semi_lagrange_mac<1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, dst, src, dt, orderTrace);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_correct(int sx, int sy, int sz, int ncomp, const int32_t* flags, float* dst, const float* old,
                          const float* fwd, const float* bwd, float strength, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (ncomp == 1)
		hipLaunchKernelGGL((k_maccormack_correct<1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, dst, old, fwd, bwd, strength);
	else if (ncomp == 3)
		hipLaunchKernelGGL((k_maccormack_correct<3>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, dst, old, fwd, bwd, strength);
	else
		return fail("ncomp must be 1 or 3");
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_correct_mac(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* old, const float* fwd,
                              const float* bwd, float strength, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_maccormack_correct_mac, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, dst, old, fwd, bwd, strength);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst,
                        const float* orig, const float* fwd, float dt, int clampMode, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (ncomp == 1)
		hipLaunchKernelGGL((k_maccormack_clamp<1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, dt, clampMode);
	else if (ncomp == 3)
		hipLaunchKernelGGL((k_maccormack_clamp<3>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, dt, clampMode);
	else
		return fail("ncomp must be 1 or 3");
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst, const float* orig,
                            const float* fwd, float dt, int clampMode, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_maccormack_clamp_mac, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, dt, clampMode);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_correct_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst, const float* orig,
                                const float* fwd, const float* bwd, float strength, float dt, int clampMode, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (dst == orig || dst == fwd || dst == bwd) return fail("mf_maccormack_correct_clamp: dst must not alias orig / fwd / bwd");
	if (ncomp == 1)
		hipLaunchKernelGGL((k_mc_correct_clamp<1>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, bwd, strength, dt, clampMode);
	else if (ncomp == 3)
		hipLaunchKernelGGL((k_mc_correct_clamp<3>), dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, bwd, strength, dt, clampMode);
	else
		return fail("ncomp must be 1 or 3");
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_maccormack_correct_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst, const float* orig,
                                    const float* fwd, const float* bwd, float strength, float dt, int clampMode, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (dst == orig || dst == fwd || dst == bwd) return fail("mf_maccormack_correct_clamp_mac: dst must not alias orig / fwd / bwd");
	hipLaunchKernelGGL(k_mc_correct_clamp_mac, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, dst, orig, fwd, bwd, strength, dt, clampMode);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_apply_outflow_bc(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* velPrev, float* velDst,
                        float dtIn, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	// applyOutflowBC(double timeStep) hands max(1.0, timeStep*4) to a Real parameter, advection.cpp:388-391
	const double t4 = (double)dtIn * 4;
	const float timeStep = (float)(1.0 > t4 ? 1.0 : t4);
	hipLaunchKernelGGL(k_outflow_extrapolate, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, velDst, velPrev, timeStep);
	hipLaunchKernelGGL(k_copy_changed_vels, dim3(nblk(d)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, velDst, vel);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
