// p2g_ordered.hip -- particle -> grid transfers that reproduce the reference's SERIAL scatter bit for bit, in parallel.
//
// knMapLinearVec3ToMACGrid / knMapLinear<T> are KERNEL(pts, single): one thread walks the particles in index order and
// adds w and w*val into the 8 nodes around each particle (setInterpolMAC / setInterpol, util/interpol.h:96-113, 166-213).
// Every node is therefore an independent fp32 accumulator that receives its contributions in increasing particle index.
// That is a gather: a node (i,j,k) is reached exactly by the particles whose base cell is (i-di, j-dj, k-dk), di,dj,dk in
// {0,1}, through the corner (di,dj,dk).  So:
//   1. key(p) = flat base cell of the component (skipped particles: key n), per-cell histogram, exclusive scan;
//   2. stable radix sort of (key, p): each base cell's particles, in increasing p;
//   3. one thread per node merges its (up to) 8 sorted lists by p and accumulates  acc_w += w ; acc_v += w*val  in
//      exactly the reference's order.
// Cost: every particle is visited by 8 nodes per component (position + value re-read from L2); no atomics, no order
// dependence, same bits as the reference on every run.  In 2D the two z-corners alias the same node (strideZ = 0), the
// per-particle order of those two adds follows the statement order of the reference (z-corners first, except for the
// Z component).
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <limits.h>

using namespace mf;

namespace {

static inline unsigned nblk_n(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK > 0 ? (n + BLOCK - 1) / BLOCK : 1); }

struct Corner {
	int bx, by, bz;
	float s[2], t[2], f[2];
};
// MODE 0/1/2: MAC component X/Y/Z (BUILD_INDEX_SHIFT, interpol.h:116-129); MODE 3: cell-centred (BUILD_INDEX)
template <int MODE>
__device__ __forceinline__ Corner corner_of(const Dim& d, float x, float y, float z) {
	Corner c;
	const Bi b = build_index(d, x, y, z);
	c.bx = b.xi; c.by = b.yi; c.bz = b.zi;
	c.s[0] = b.s0; c.s[1] = b.s1; c.t[0] = b.t0; c.t[1] = b.t1; c.f[0] = b.f0; c.f[1] = b.f1;
	if (MODE != 3) {
		const Bi sh = build_index_shift(d, x, y, z);
		if (MODE == 0) { c.bx = sh.xi; c.s[0] = sh.s0; c.s[1] = sh.s1; }
		if (MODE == 1) { c.by = sh.yi; c.t[0] = sh.t0; c.t[1] = sh.t1; }
		if (MODE == 2) { c.bz = sh.zi; c.f[0] = sh.f0; c.f[1] = sh.f1; }
	}
	return c;
}

template <int MODE>
__global__ void __launch_bounds__(BLOCK)
k_keys(Dim d, int64_t np, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ pflag, const int32_t* __restrict__ ptype,
       int exclude, int32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* __restrict__ counts) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	int key = (int)d.n;
	if (!((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude)))) {
		const Corner c = corner_of<MODE>(d, pos[p], pos[ps + p], pos[2 * ps + p]);
		key = (int)((int64_t)c.bx + d.sx * ((int64_t)c.by + (int64_t)d.sy * c.bz));
		atomicAdd(&counts[key], 1);
	}
	keys[p] = key;
	vals[p] = (int)p;
}

// one thread per node; NCOMP values per particle (1 for a MAC component / Real grid, 3 for a Vec3 grid)
template <int MODE, int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_gather(Dim d, int64_t ps, const float* __restrict__ pos, const float* __restrict__ pval, int64_t vstride, const int32_t* __restrict__ order,
         const int32_t* __restrict__ start, float* __restrict__ ref, int64_t rstride, float* __restrict__ sum) {
	const int64_t node = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (node >= d.n) return;
	const int i = (int)(node % d.sx), j = (int)((node / d.sx) % d.sy), k = (int)(node / ((int64_t)d.sx * d.sy));
	const int nlist = d.is3d ? 8 : 4;
	int cur[8], end[8], head[8];
#pragma unroll
	for (int q = 0; q < 8; q++) {
		const int di = q & 1, dj = (q >> 1) & 1, dk = q >> 2;
		const int bx = i - di, by = j - dj, bz = k - dk;
		const bool ok = (q < nlist) && bx >= 0 && by >= 0 && bz >= 0;
		int a = 0, e = 0;
		if (ok) {
			const int64_t c = (int64_t)bx + d.sx * ((int64_t)by + (int64_t)d.sy * bz);
			a = start[c];
			e = start[c + 1];
		}
		cur[q] = a;
		end[q] = e;
		head[q] = a < e ? order[a] : INT_MAX;
	}
	float acc_w = 0.f, acc_v[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) acc_v[c] = 0.f;
	for (;;) {
		int best = INT_MAX, bm = 0;
#pragma unroll
		for (int q = 0; q < 8; q++)
			if (head[q] < best) {
				best = head[q];
				bm = q;
			}
		if (best == INT_MAX) break;
#pragma unroll
		for (int q = 0; q < 8; q++)
			if (q == bm) {
				cur[q]++;
				head[q] = cur[q] < end[q] ? order[cur[q]] : INT_MAX;
			}
		const int p = best;
		const Corner c = corner_of<MODE>(d, pos[p], pos[ps + p], pos[2 * ps + p]);
		const int di = bm & 1, dj = (bm >> 1) & 1, dk = bm >> 2;
		const float sw = di ? c.s[1] : c.s[0], tw = dj ? c.t[1] : c.t[0];
		float v[NCOMP];
#pragma unroll
		for (int cc = 0; cc < NCOMP; cc++) v[cc] = pval[cc * vstride + p];
		if (d.is3d) {
			const float w = tw * (sw * (dk ? c.f[1] : c.f[0]));
			acc_w += w;
#pragma unroll
			for (int cc = 0; cc < NCOMP; cc++) acc_v[cc] += w * v[cc];
		} else {
			// strideZ == 0: both z-corners land on this node, in the reference's statement order
			const float wa = tw * (sw * ((MODE == 2) ? c.f[0] : c.f[1]));
			const float wb = tw * (sw * ((MODE == 2) ? c.f[1] : c.f[0]));
			acc_w += wa;
			acc_w += wb;
#pragma unroll
			for (int cc = 0; cc < NCOMP; cc++) {
				acc_v[cc] += wa * v[cc];
				acc_v[cc] += wb * v[cc];
			}
		}
	}
	sum[node] = acc_w;
#pragma unroll
	for (int cc = 0; cc < NCOMP; cc++) ref[cc * rstride + node] = acc_v[cc];
}

// ---- APIC (knApicMapLinearVec3ToMACGrid, apic.cpp:19-90): same gather, other weights.  The reference addresses the 8 nodes
// by FLAT index gidx + dX[i] + dY[j] + dZ[k] with no bounds check; the gather follows the flat arithmetic (a node n is fed by
// the base indices n - (i + j*Y + k*Z)), faces whose base index lies outside [0, n) are skipped like in the oracle.
template <int COMP>
__global__ void __launch_bounds__(BLOCK)
k_keys_apic(Dim d, int64_t np, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ pflag, const int32_t* __restrict__ ptype,
            int exclude, int32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* __restrict__ counts) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	int key = (int)d.n;
	if (!((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude)))) {
		const ApicFace a = apic_face<COMP>(d, pos[p], pos[ps + p], pos[2 * ps + p]);
		if (a.gidx >= 0 && a.gidx < d.n) {
			key = (int)a.gidx;
			atomicAdd(&counts[key], 1);
		}
	}
	keys[p] = key;
	vals[p] = (int)p;
}

// one thread per node of face COMP: per particle (increasing index)  m += w ; v += w*vel_c ; v += w*dot(cp_c, node - pos)
template <int COMP>
__global__ void __launch_bounds__(BLOCK)
k_gather_apic(Dim d, int64_t ps, const float* __restrict__ pos, const float* __restrict__ pvc, const float* __restrict__ cp,
              const int32_t* __restrict__ order, const int32_t* __restrict__ start, float* __restrict__ vel, float* __restrict__ mass) {
	const int64_t node = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (node >= d.n) return;
	const int nlist = d.is3d ? 8 : 4;
	int cur[8], end[8], head[8];
#pragma unroll
	for (int q = 0; q < 8; q++) {
		const int64_t base = node - ((q & 1) + ((q >> 1) & 1) * d.Y + (q >> 2) * d.Z);
		int a = 0, e = 0;
		if (q < nlist && base >= 0) {
			a = start[base];
			e = start[base + 1];
		}
		cur[q] = a;
		end[q] = e;
		head[q] = a < e ? order[a] : INT_MAX;
	}
	float acc_m = 0.f, acc_v = 0.f;
	for (;;) {
		int best = INT_MAX, bm = 0;
#pragma unroll
		for (int q = 0; q < 8; q++)
			if (head[q] < best) {
				best = head[q];
				bm = q;
			}
		if (best == INT_MAX) break;
#pragma unroll
		for (int q = 0; q < 8; q++)
			if (q == bm) {
				cur[q]++;
				head[q] = cur[q] < end[q] ? order[cur[q]] : INT_MAX;
			}
		const int p = best;
		const float px = pos[p], py = pos[ps + p], pz = pos[2 * ps + p];
		const ApicFace a = apic_face<COMP>(d, px, py, pz);
		const int di = bm & 1, dj = (bm >> 1) & 1;
		const float vc = pvc[p], c0 = cp[p], c1 = cp[ps + p], c2 = cp[2 * ps + p];
		const float wij = (di ? a.W[0][1] : a.W[0][0]) * (dj ? a.W[1][1] : a.W[1][0]);
		const float dx = (a.gpos[0] + (float)di) - px, dy = (a.gpos[1] + (float)dj) - py;
		// 3D: the list fixes k; 2D (strideZ == 0): k = 0 then k = 1 land on this node, in the reference's loop order
		const int k0 = d.is3d ? (bm >> 2) : 0, k1 = d.is3d ? (bm >> 2) : 1;
		for (int k = k0; k <= k1; k++) {
			const float w = wij * (k ? a.W[2][1] : a.W[2][0]);
			const float dz = (a.gpos[2] + (float)k) - pz;
			acc_m += w;
			acc_v += w * vc;
			acc_v += w * (c0 * dx + c1 * dy + c2 * dz);
		}
	}
	mass[node] = acc_m;
	vel[node] = acc_v;
}

struct Scratch {
	int32_t* keys = nullptr;    // [2 * cap_p]
	int32_t* vals = nullptr;    // [2 * cap_p]
	int32_t* counts = nullptr;  // [cap_n + 1]
	int32_t* start = nullptr;   // [cap_n + 1]
	void* tmp = nullptr;
	int64_t cap_p = 0, cap_n = 0;
	size_t cap_tmp = 0;
};
Scratch g_scratch[16];

int get_scratch(int64_t np, int64_t n, size_t tmp_bytes, Scratch** out) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	Scratch& s = g_scratch[dev];
	if (np > s.cap_p || n > s.cap_n || tmp_bytes > s.cap_tmp) MF_HIP(hipDeviceSynchronize());
	if (np > s.cap_p) {
		if (s.keys) MF_HIP(hipFree(s.keys));
		if (s.vals) MF_HIP(hipFree(s.vals));
		s.cap_p = np + np / 8 + 1024;
		MF_HIP(hipMalloc((void**)&s.keys, sizeof(int32_t) * 2 * s.cap_p));
		MF_HIP(hipMalloc((void**)&s.vals, sizeof(int32_t) * 2 * s.cap_p));
	}
	if (n > s.cap_n) {
		if (s.counts) MF_HIP(hipFree(s.counts));
		if (s.start) MF_HIP(hipFree(s.start));
		s.cap_n = n;
		MF_HIP(hipMalloc((void**)&s.counts, sizeof(int32_t) * (s.cap_n + 1)));
		MF_HIP(hipMalloc((void**)&s.start, sizeof(int32_t) * (s.cap_n + 1)));
	}
	if (tmp_bytes > s.cap_tmp) {
		if (s.tmp) MF_HIP(hipFree(s.tmp));
		s.cap_tmp = tmp_bytes + (tmp_bytes >> 2);
		MF_HIP(hipMalloc(&s.tmp, s.cap_tmp));
	}
	*out = &s;
	return 0;
}

template <int MODE, int NCOMP>
int run(const Dim& d, int64_t np, int64_t ps, const float* pos, const int32_t* pflag, const int32_t* ptype, int exclude,
        const float* pval, int64_t vstride, float* ref, int64_t rstride, float* sum, hipStream_t st) {
	if (np >= ((int64_t)1 << 31) - 1) return fail("ordered P2G: too many particles for 32-bit indices");
	size_t scan_bytes = 0, sort_bytes = 0;
	int end_bit = 1;
	while (end_bit < 31 && (((int64_t)1 << end_bit) <= d.n)) end_bit++;
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int)(d.n + 1), st));
	MF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)np, 0, end_bit, st));
	Scratch* s;
	MF_TRY(get_scratch(np, d.n, (scan_bytes > sort_bytes ? scan_bytes : sort_bytes) + 256, &s));
	MF_HIP(hipMemsetAsync(s->counts, 0, sizeof(int32_t) * (d.n + 1), st));
	hipLaunchKernelGGL((k_keys<MODE>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, np, ps, pos, pflag, ptype, exclude, s->keys, s->vals, s->counts);
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(s->tmp, scan_bytes, s->counts, s->start, (int)(d.n + 1), st));
	MF_HIP(hipcub::DeviceRadixSort::SortPairs(s->tmp, sort_bytes, s->keys, s->keys + np, s->vals, s->vals + np, (int)np, 0, end_bit, st));
	hipLaunchKernelGGL((k_gather<MODE, NCOMP>), dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d, ps, pos, pval, vstride, s->vals + np, s->start, ref, rstride, sum);
	MF_LAUNCH_CHECK();
	return 0;
}

template <int COMP>
int run_apic(const Dim& d, int64_t np, int64_t ps, const float* pos, const int32_t* pflag, const int32_t* ptype, int exclude,
             const float* pvc, const float* cp, float* vel, float* mass, hipStream_t st) {
	if (np >= ((int64_t)1 << 31) - 1) return fail("ordered P2G: too many particles for 32-bit indices");
	size_t scan_bytes = 0, sort_bytes = 0;
	int end_bit = 1;
	while (end_bit < 31 && (((int64_t)1 << end_bit) <= d.n)) end_bit++;
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int)(d.n + 1), st));
	MF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int)np, 0, end_bit, st));
	Scratch* s;
	MF_TRY(get_scratch(np, d.n, (scan_bytes > sort_bytes ? scan_bytes : sort_bytes) + 256, &s));
	MF_HIP(hipMemsetAsync(s->counts, 0, sizeof(int32_t) * (d.n + 1), st));
	hipLaunchKernelGGL((k_keys_apic<COMP>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, np, ps, pos, pflag, ptype, exclude, s->keys, s->vals, s->counts);
	MF_HIP(hipcub::DeviceScan::ExclusiveSum(s->tmp, scan_bytes, s->counts, s->start, (int)(d.n + 1), st));
	MF_HIP(hipcub::DeviceRadixSort::SortPairs(s->tmp, sort_bytes, s->keys, s->keys + np, s->vals, s->vals + np, (int)np, 0, end_bit, st));
	hipLaunchKernelGGL((k_gather_apic<COMP>), dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d, ps, pos, pvc, cp, s->vals + np, s->start, vel, mass);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // namespace

namespace mf {

// vel / weight: SoA MAC grids (already zeroed or not: every node is written)
int p2g_ordered_mac(const Dim& d, float* vel, float* weight, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                    const float* pvel, const int32_t* ptype, int exclude, hipStream_t st) {
	MF_TRY((run<0, 1>(d, np, ps, pos, pflag, ptype, exclude, pvel, ps, vel, d.n, weight, st)));
	MF_TRY((run<1, 1>(d, np, ps, pos, pflag, ptype, exclude, pvel + ps, ps, vel + d.n, d.n, weight + d.n, st)));
	MF_TRY((run<2, 1>(d, np, ps, pos, pflag, ptype, exclude, pvel + 2 * ps, ps, vel + 2 * d.n, d.n, weight + 2 * d.n, st)));
	return 0;
}
// target: ncomp planes; wsum: Real grid
int p2g_ordered_cell(const Dim& d, int ncomp, float* target, float* wsum, int64_t np, int64_t ps, const float* pos,
                     const int32_t* pflag, const float* psrc, hipStream_t st) {
	if (ncomp == 1) return run<3, 1>(d, np, ps, pos, pflag, nullptr, 0, psrc, ps, target, d.n, wsum, st);
	return run<3, 3>(d, np, ps, pos, pflag, nullptr, 0, psrc, ps, target, d.n, wsum, st);
}

// vel / mass: SoA MAC grids; in 2-D the w face is not touched by the reference (apic.cpp:73) -> the caller's zeros stay
int p2g_ordered_apic(const Dim& d, float* vel, float* mass, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                     const float* pvel, const float* cpx, const float* cpy, const float* cpz, const int32_t* ptype, int exclude,
                     hipStream_t st) {
	MF_TRY((run_apic<0>(d, np, ps, pos, pflag, ptype, exclude, pvel, cpx, vel, mass, st)));
	MF_TRY((run_apic<1>(d, np, ps, pos, pflag, ptype, exclude, pvel + ps, cpy, vel + d.n, mass + d.n, st)));
	if (d.is3d) MF_TRY((run_apic<2>(d, np, ps, pos, pflag, ptype, exclude, pvel + 2 * ps, cpz, vel + 2 * d.n, mass + 2 * d.n, st)));
	return 0;
}

}  // namespace mf
