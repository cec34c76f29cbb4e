// pressure.hip -- pressure projection on gfx950: 7-point ApplyMatrix stencil, Laplace matrix / rhs assembly,
// velocity correction (+ ghost fluid), MIC(0) preconditioner as tile wavefronts, and the device-resident PCG loop.
// Reference: source/conjugategrad.{h,cpp}, source/plugin/pressure.cpp (cited per kernel).
#include "common.h"
#include <float.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

using namespace mf;

// =========================================================================================================
// ApplyMatrix, conjugategrad.h:118-151
//   non-fluid: dst = src ; fluid: left-to-right fp32 sum of the 7 products.  28 B/cell of compulsory traffic
//   (flags, src, A0, Ai, Aj, Ak read once + dst written); the +-X/+-Y/+-Z neighbours of src and the -X/-Y/-Z
//   entries of Ai/Aj/Ak are re-reads served by L1/L2 (each XCD owns a contiguous z-range of the grid).
//   Optional fusion: per-block fp64 partial sums of dst*src (GridDotProduct(tmp, search)).
// =========================================================================================================
struct CgScalars {
	float sigma, alpha, nalpha, beta, resNorm, dp, sigmaNew, accuracy;
	int iterations, done, diverged, useL2;
};

template <bool DOT, bool IS3D>
__global__ void __launch_bounds__(BLOCK)
k_apply_matrix_v4(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ src,
                  const float* __restrict__ A0, const float* __restrict__ Ai, const float* __restrict__ Aj,
                  const float* __restrict__ Ak, double* __restrict__ partials, const CgScalars* __restrict__ sc, int qpt) {
	if (DOT && sc->done) return;
	const int64_t nq = d.n >> 2;
	const int qx = d.sx >> 2;  // quads per row (sx % 4 == 0)
	double acc = 0.0;
	// each block owns a contiguous run of qpt*BLOCK quads; with the XCD remap every XCD sweeps one contiguous
	// z-slab front to back, so the +-Z neighbour planes are still in that XCD's L2 when they are re-read
	const int vb = xcd_swizzle(blockIdx.x, gridDim.x);
	const int64_t base = (int64_t)vb * BLOCK * qpt + threadIdx.x;
	for (int t = 0; t < qpt; t++) {
		const int64_t q = base + (int64_t)t * BLOCK;
		if (q >= nq) break;
		const int64_t idx = q << 2;
		const int4 f = ((const int4*)flags)[q];
		const float4 s = ((const float4*)src)[q];
		float4 r = s;
		if ((f.x | f.y | f.z | f.w) & MF_FLUID) {
			const int64_t row = q / qx;
			const int i0 = (int)(q - row * qx) << 2;
			const int j = (int)(row % d.sy);
			const int k = (int)(row / d.sy);
			const float4 a0 = ((const float4*)A0)[q];
			const float4 ai = ((const float4*)Ai)[q];
			const float4 aj = ((const float4*)Aj)[q];
			const float sl = (idx > 0) ? src[idx - 1] : 0.f;
			const float al = (idx > 0) ? Ai[idx - 1] : 0.f;
			const float sr = (idx + 4 < d.n) ? src[idx + 4] : 0.f;
			const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
			const float4 sym = (j > 0) ? *(const float4*)(src + idx - d.Y) : z4;
			const float4 ajm = (j > 0) ? *(const float4*)(Aj + idx - d.Y) : z4;
			const float4 syp = (j < d.sy - 1) ? *(const float4*)(src + idx + d.Y) : z4;
			float4 szm = z4, akm = z4, szp = z4, ak = z4;
			if (IS3D) {
				ak = ((const float4*)Ak)[q];
				if (k > 0) {
					szm = *(const float4*)(src + idx - d.Z);
					akm = *(const float4*)(Ak + idx - d.Z);
				}
				if (k < d.sz - 1) szp = *(const float4*)(src + idx + d.Z);
			}
			(void)i0;
#define CELL(c, SL, AL, SR)                                                                       \
	if (f.c & MF_FLUID) {                                                                         \
		float v = s.c * a0.c;                                                                     \
		v = v + (SL) * (AL);                                                                      \
		v = v + (SR) * ai.c;                                                                      \
		v = v + sym.c * ajm.c;                                                                    \
		v = v + syp.c * aj.c;                                                                     \
		if (IS3D) {                                                                               \
			v = v + szm.c * akm.c;                                                                \
			v = v + szp.c * ak.c;                                                                 \
		}                                                                                         \
		r.c = v;                                                                                  \
	}
			CELL(x, sl, al, s.y)
			CELL(y, s.x, ai.x, s.z)
			CELL(z, s.y, ai.y, s.w)
			CELL(w, s.z, ai.z, sr)
#undef CELL
		}
		((float4*)dst)[q] = r;
		if (DOT) {
			const float p0 = r.x * s.x, p1 = r.y * s.y, p2 = r.z * s.z, p3 = r.w * s.w;
			acc += (double)p0;
			acc += (double)p1;
			acc += (double)p2;
			acc += (double)p3;
		}
	}
	if (DOT) {
		acc = block_sum(acc);
		if (threadIdx.x == 0) partials[blockIdx.x] = acc;
	}
}

// v5: every load is issued before anything depends on it (no flags -> operand round trip), each thread owns R
// consecutive y-rows of one x-quad so src[j+-1] / Aj[j-1] are reused from registers, and the +-X neighbours come
// from the adjacent lanes (DPP wave shift) instead of three extra 4-byte-per-lane loads.  Same arithmetic, same
// order per cell as v4.
__device__ __forceinline__ float wave_shr1(float v) {  // lane l gets lane l-1 (lane 0: unchanged)
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float v) {  // lane l gets lane l+1 (lane 63: unchanged)
	return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
template <bool DOT, bool IS3D, int R>
__global__ void __launch_bounds__(BLOCK)
k_apply_matrix_v5(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ src,
                  const float* __restrict__ A0, const float* __restrict__ Ai, const float* __restrict__ Aj,
                  const float* __restrict__ Ak, double* __restrict__ partials, const CgScalars* __restrict__ sc, int jgroups, int tpb) {
	if (DOT && sc->done) return;
	const int qx = d.sx >> 2;
	const int64_t nthr = (int64_t)qx * jgroups * d.sz;
	const int vb0 = xcd_swizzle(blockIdx.x, gridDim.x) * tpb;
	double acc = 0.0;
#pragma unroll 1
	for (int t = 0; t < tpb; t++) {
	const int64_t T0 = (int64_t)(vb0 + t) * BLOCK + threadIdx.x;
	const bool live = T0 < nthr;
	const int64_t T = live ? T0 : nthr - 1;
	const int qi = (int)(T % qx);
	const int64_t rg = T / qx;
	const int j0 = (int)(rg % jgroups) * R;
	const int k = (int)(rg / jgroups);
	const int lane = threadIdx.x & 63;
	const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
	const int64_t Y = d.Y, Z = d.Z;
	const int64_t row0 = ((int64_t)k * d.sy + j0) * d.sx + 4 * qi;  // flat index of this thread's first quad
	int4 f[R];
	float4 sv[R + 2], ajv[R + 1], a0[R], ai[R], ak[R], akm[R], szm[R], szp[R];
	float sl[R], al[R], sr[R];
	// ---- loads (addresses clamped into the grid; out-of-domain neighbours are zeroed afterwards) ----
	{
		const int jm = (j0 > 0) ? -1 : 0;
		sv[0] = *(const float4*)(src + row0 + jm * Y);
		ajv[0] = *(const float4*)(Aj + row0 + jm * Y);
	}
#pragma unroll
	for (int r = 0; r < R; r++) {
		const int jr = (j0 + r < d.sy) ? r : (d.sy - 1 - j0);
		const int64_t idx = row0 + jr * Y;
		f[r] = *(const int4*)(flags + idx);
		sv[r + 1] = *(const float4*)(src + idx);
		ajv[r + 1] = *(const float4*)(Aj + idx);
		a0[r] = *(const float4*)(A0 + idx);
		ai[r] = *(const float4*)(Ai + idx);
		if (IS3D) {
			ak[r] = *(const float4*)(Ak + idx);
			const int64_t im = (k > 0) ? idx - Z : idx, ip = (k < d.sz - 1) ? idx + Z : idx;
			akm[r] = *(const float4*)(Ak + im);
			szm[r] = *(const float4*)(src + im);
			szp[r] = *(const float4*)(src + ip);
		}
	}
	{
		const int jr = (j0 + R < d.sy) ? R : (d.sy - 1 - j0);
		sv[R + 1] = *(const float4*)(src + row0 + jr * Y);
	}
	// ---- +-X neighbours: adjacent lanes hold the adjacent quads of the same row, except at row / wave edges ----
	const bool edge_l = (lane == 0) || (qi == 0), edge_r = (lane == 63) || (qi == qx - 1);
#pragma unroll
	for (int r = 0; r < R; r++) {
		const int jr = (j0 + r < d.sy) ? r : (d.sy - 1 - j0);
		const int64_t idx = row0 + jr * Y;
		sl[r] = wave_shr1(sv[r + 1].w);
		al[r] = wave_shr1(ai[r].w);
		sr[r] = wave_shl1(sv[r + 1].x);
		if (edge_l) {
			sl[r] = (idx > 0) ? src[idx - 1] : 0.f;
			al[r] = (idx > 0) ? Ai[idx - 1] : 0.f;
		}
		if (edge_r) sr[r] = (idx + 4 < d.n) ? src[idx + 4] : 0.f;
	}
#pragma unroll
	for (int r = 0; r < R; r++) {
		const int j = j0 + r;
		if (j >= d.sy) break;
		const int4 fl = f[r];
		const float4 s = sv[r + 1];
		float4 res = s;
		if ((fl.x | fl.y | fl.z | fl.w) & MF_FLUID) {
			const float4 sym = (j > 0) ? sv[r] : z4, ajm = (j > 0) ? ajv[r] : z4;
			const float4 syp = (j < d.sy - 1) ? sv[r + 2] : z4, aj = ajv[r + 1];
			float4 zm = z4, am = z4, zp = z4, akc = z4;
			if (IS3D) {
				akc = ak[r];
				if (k > 0) {
					zm = szm[r];
					am = akm[r];
				}
				if (k < d.sz - 1) zp = szp[r];
			}
#define CELL5(c, SL, AL, SR)                                                                      \
	if (fl.c & MF_FLUID) {                                                                        \
		float v = s.c * a0[r].c;                                                                  \
		v = v + (SL) * (AL);                                                                      \
		v = v + (SR) * ai[r].c;                                                                   \
		v = v + sym.c * ajm.c;                                                                    \
		v = v + syp.c * aj.c;                                                                     \
		if (IS3D) {                                                                               \
			v = v + zm.c * am.c;                                                                  \
			v = v + zp.c * akc.c;                                                                 \
		}                                                                                         \
		res.c = v;                                                                                \
	}
			CELL5(x, sl[r], al[r], s.y)
			CELL5(y, s.x, ai[r].x, s.z)
			CELL5(z, s.y, ai[r].y, s.w)
			CELL5(w, s.z, ai[r].z, sr[r])
#undef CELL5
		}
		if (live) {
			*(float4*)(dst + row0 + r * Y) = res;
			if (DOT) {
				const float p0 = res.x * s.x, p1 = res.y * s.y, p2 = res.z * s.z, p3 = res.w * s.w;
				acc += (double)p0;
				acc += (double)p1;
				acc += (double)p2;
				acc += (double)p3;
			}
		}
	}
	}
	if (DOT) {
		acc = block_sum(acc);
		if (threadIdx.x == 0) partials[blockIdx.x] = acc;
	}
}

// generic fallback (sx % 4 != 0 or unaligned views): one cell per thread
template <bool DOT>
__global__ void __launch_bounds__(BLOCK)
k_apply_matrix_scalar(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ src,
                      const float* __restrict__ A0, const float* __restrict__ Ai, const float* __restrict__ Aj,
                      const float* __restrict__ Ak, double* __restrict__ partials, const CgScalars* __restrict__ sc) {
	if (DOT && sc->done) return;
	double acc = 0.0;
	for (int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x; idx < d.n; idx += (int64_t)gridDim.x * BLOCK) {
		const float s = src[idx];
		float v = s;
		if (flags[idx] & MF_FLUID) {
			const int64_t Y = d.Y, Z = d.Z;
			v = s * A0[idx];
			v = v + ((idx >= 1) ? src[idx - 1] * Ai[idx - 1] : 0.f);
			v = v + ((idx + 1 < d.n) ? src[idx + 1] : 0.f) * Ai[idx];
			v = v + ((idx >= Y) ? src[idx - Y] * Aj[idx - Y] : 0.f);
			v = v + ((idx + Y < d.n) ? src[idx + Y] : 0.f) * Aj[idx];
			if (d.is3d) {
				v = v + ((idx >= Z) ? src[idx - Z] * Ak[idx - Z] : 0.f);
				v = v + ((idx + Z < d.n) ? src[idx + Z] : 0.f) * Ak[idx];
			}
		}
		dst[idx] = v;
		if (DOT) {
			const float p = v * s;
			acc += (double)p;
		}
	}
	if (DOT) {
		acc = block_sum(acc);
		if (threadIdx.x == 0) partials[blockIdx.x] = acc;
	}
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// returns the number of blocks launched (== number of partials written when DOT)
template <bool DOT>
static int launch_apply_matrix(const Dim& d, const int32_t* flags, float* dst, const float* src, const float* A0,
                               const float* Ai, const float* Aj, const float* Ak, double* partials,
                               const CgScalars* sc, hipStream_t st, int* nblocks) {
	const bool vec = (d.sx % 4 == 0) && al16(flags) && al16(dst) && al16(src) && al16(A0) && al16(Ai) && al16(Aj) && al16(Ak);
	int nb;
	static const int am_rows = [] {
		const char* e = getenv("MF_AM_ROWS");
		return e ? atoi(e) : 2;
	}();
	if (vec && am_rows > 0) {
		const int R = am_rows >= 4 ? 4 : (am_rows >= 2 ? 2 : 1);
		const int jgroups = (d.sy + R - 1) / R;
		const int64_t vblocks = ((int64_t)(d.sx >> 2) * jgroups * d.sz + BLOCK - 1) / BLOCK;
		const int tpb = (int)((vblocks + MAX_BLOCKS - 1) / MAX_BLOCKS);
		nb = (int)((vblocks + tpb - 1) / tpb);
#define AM5(RR)                                                                                                                                   \
	if (d.is3d)                                                                                                                                   \
		hipLaunchKernelGGL((k_apply_matrix_v5<DOT, true, RR>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, jgroups, tpb); \
	else                                                                                                                                          \
		hipLaunchKernelGGL((k_apply_matrix_v5<DOT, false, RR>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, jgroups, tpb);
		if (R == 4) { AM5(4) } else if (R == 2) { AM5(2) } else { AM5(1) }
#undef AM5
	} else if (vec) {
		const int64_t nq = d.n >> 2;
		const int qpt = (int)((nq + (int64_t)BLOCK * MAX_BLOCKS - 1) / ((int64_t)BLOCK * MAX_BLOCKS));
		nb = (int)((nq + (int64_t)BLOCK * qpt - 1) / ((int64_t)BLOCK * qpt));
		if (nb < 1) nb = 1;
		if (d.is3d)
			hipLaunchKernelGGL((k_apply_matrix_v4<DOT, true>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, qpt);
		else
			hipLaunchKernelGGL((k_apply_matrix_v4<DOT, false>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, qpt);
	} else {
		nb = blocks_for(d.n, BLOCK, 2048);
		hipLaunchKernelGGL((k_apply_matrix_scalar<DOT>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc);
	}
	MF_LAUNCH_CHECK();
	if (nblocks) *nblocks = nb;
	return 0;
}

// =========================================================================================================
// assembly kernels (one thread per cell, bnd = 1 unless noted)
// =========================================================================================================
#define CELL_IJK(d)                                                               \
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;                \
	if (idx >= (d).n) return;                                                     \
	const int i = (int)(idx % (d).sx);                                            \
	const int j = (int)((idx / (d).sx) % (d).sy);                                 \
	const int k = (int)(idx / ((int64_t)(d).sx * (d).sy));                        \
	(void)i; (void)j; (void)k;
#define INTERIOR(d) (i >= 1 && i < (d).sx - 1 && j >= 1 && j < (d).sy - 1 && (!(d).is3d || (k >= 1 && k < (d).sz - 1)))

// MakeLaplaceMatrix, conjugategrad.h:154-187
__global__ void __launch_bounds__(BLOCK)
k_make_laplace(Dim d, const int32_t* __restrict__ flags, float* __restrict__ A0, float* __restrict__ Ai,
               float* __restrict__ Aj, float* __restrict__ Ak, const float* __restrict__ fr) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (!(flags[idx] & MF_FLUID)) return;
	const int64_t Y = d.Y, Z = d.Z;
	float a0 = A0[idx];
	if (!fr) {
		// `A0 += 1.` : fp32 value + double literal, rounded back; exact for these magnitudes
		if (!(flags[idx - 1] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (!(flags[idx + 1] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (!(flags[idx - Y] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (!(flags[idx + Y] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (d.is3d && !(flags[idx - Z] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (d.is3d && !(flags[idx + Z] & MF_OBSTACLE)) a0 = (float)((double)a0 + 1.);
		if (flags[idx + 1] & MF_FLUID) Ai[idx] = -1.f;
		if (flags[idx + Y] & MF_FLUID) Aj[idx] = -1.f;
		if (d.is3d && (flags[idx + Z] & MF_FLUID)) Ak[idx] = -1.f;
	} else {
		const float *fx = fr, *fy = fr + d.n, *fz = fr + 2 * d.n;
		a0 += fx[idx];
		a0 += fx[idx + 1];
		a0 += fy[idx];
		a0 += fy[idx + Y];
		if (d.is3d) a0 += fz[idx];
		if (d.is3d) a0 += fz[idx + Z];
		if (flags[idx + 1] & MF_FLUID) Ai[idx] = -fx[idx + 1];
		if (flags[idx + Y] & MF_FLUID) Aj[idx] = -fy[idx + Y];
		if (d.is3d && (flags[idx + Z] & MF_FLUID)) Ak[idx] = -fz[idx + Z];
	}
	A0[idx] = a0;
}

// ghost-fluid helpers, plugin/pressure.cpp:115-133, 191-196
__device__ __forceinline__ float thetaHelper(float inside, float outside) {
	const float denom = inside - outside;
	if ((double)denom > -1e-04) return 0.5f;
	const float q = inside / denom;
	const float m = q < 1.f ? q : 1.f;
	return 0.f < m ? m : 0.f;
}
__device__ __forceinline__ float ghostFluidHelper(int64_t idx, int64_t offset, const float* __restrict__ phi, float gfClamp) {
	const float alpha = thetaHelper(phi[idx], phi[idx + offset]);
	if (alpha < gfClamp) return gfClamp;
	return (float)(1. - (1. / (double)alpha));
}
__device__ __forceinline__ float surfTensHelper(int64_t idx, int64_t offset, const float* __restrict__ phi,
                                                const float* __restrict__ curv, float surfTens, float gfClamp) {
	return surfTens * (curv[idx + offset] - ghostFluidHelper(idx, offset, phi, gfClamp) * curv[idx]);
}
__device__ __forceinline__ bool ghostFluidWasClamped(int64_t idx, int64_t offset, const float* __restrict__ phi, float gfClamp) {
	return thetaHelper(phi[idx], phi[idx + offset]) < gfClamp;
}

// MakeRhs, plugin/pressure.cpp:32-84 ; partials[b] = fp64 sum of `set`, partials[MAX_BLOCKS + b] = count
__global__ void __launch_bounds__(BLOCK)
k_make_rhs(Dim d, const int32_t* __restrict__ flags, float* __restrict__ rhs, const float* __restrict__ vel,
           const float* __restrict__ pcc, const float* __restrict__ fr, const float* __restrict__ ob,
           const float* __restrict__ phi, const float* __restrict__ curv, float surfTens, float gfClamp,
           double* __restrict__ partials) {
	double mysum = 0.0, mycnt = 0.0;
	const int64_t X = 1, Y = d.Y, Z = d.Z, n = d.n;
	const float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
	for (int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * BLOCK) {
		const int i = (int)(idx % d.sx);
		const int j = (int)((idx / d.sx) % d.sy);
		const int k = (int)(idx / ((int64_t)d.sx * d.sy));
		if (!INTERIOR(d)) continue;
		if (!(flags[idx] & MF_FLUID)) {
			rhs[idx] = 0.f;
			continue;
		}
		float set;
		if (!fr) {
			set = vx[idx] - vx[idx + X] + vy[idx] - vy[idx + Y];
			if (d.is3d) set += vz[idx] - vz[idx + Z];
		} else {
			const float *fx = fr, *fy = fr + n, *fz = fr + 2 * n;
			set = fx[idx] * vx[idx] - fx[idx + X] * vx[idx + X] + fy[idx] * vy[idx] - fy[idx + Y] * vy[idx + Y];
			if (d.is3d) set += fz[idx] * vz[idx] - fz[idx + Z] * vz[idx + Z];
			if (ob) {
				const float *ox = ob, *oy = ob + n, *oz = ob + 2 * n;
				set += (1 - fx[idx]) * ox[idx] - (1 - fx[idx + X]) * ox[idx + X] + (1 - fy[idx]) * oy[idx] -
				       (1 - fy[idx + Y]) * oy[idx + Y];
				if (d.is3d) set += (1 - fz[idx]) * oz[idx] - (1 - fz[idx + Z]) * oz[idx + Z];
			}
		}
		if (phi && curv) {
			if (flags[idx - X] & MF_EMPTY) set += surfTensHelper(idx, -X, phi, curv, surfTens, gfClamp);
			if (flags[idx + X] & MF_EMPTY) set += surfTensHelper(idx, +X, phi, curv, surfTens, gfClamp);
			if (flags[idx - Y] & MF_EMPTY) set += surfTensHelper(idx, -Y, phi, curv, surfTens, gfClamp);
			if (flags[idx + Y] & MF_EMPTY) set += surfTensHelper(idx, +Y, phi, curv, surfTens, gfClamp);
			if (d.is3d) {
				if (flags[idx - Z] & MF_EMPTY) set += surfTensHelper(idx, -Z, phi, curv, surfTens, gfClamp);
				if (flags[idx + Z] & MF_EMPTY) set += surfTensHelper(idx, +Z, phi, curv, surfTens, gfClamp);
			}
		}
		if (pcc) set += pcc[idx];
		mysum += (double)set;
		mycnt += 1.0;
		rhs[idx] = set;
	}
	mysum = block_sum(mysum);
	__syncthreads();
	mycnt = block_sum(mycnt);
	if (threadIdx.x == 0) {
		partials[blockIdx.x] = mysum;
		partials[MAX_BLOCKS + blockIdx.x] = mycnt;
	}
}
// second-level reduction for grids with more blocks than one finishing block can hold in `partials`
__global__ void __launch_bounds__(BLOCK) k_sum2_finish(int nb, const double* __restrict__ p0, const double* __restrict__ p1, double* __restrict__ out) {
	double a = strided_sum(p0, nb), b = strided_sum(p1, nb);
	a = block_sum(a);
	__syncthreads();
	b = block_sum(b);
	if (threadIdx.x == 0) {
		out[0] = a;
		out[1] = b;
	}
}

// ApplyGhostFluidDiagonal, plugin/pressure.cpp:136-151
__global__ void __launch_bounds__(BLOCK)
k_ghost_fluid_diag(Dim d, float* __restrict__ A0, const int32_t* __restrict__ flags, const float* __restrict__ phi, float gfClamp) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (!(flags[idx] & MF_FLUID)) return;
	const int64_t X = 1, Y = d.Y, Z = d.Z;
	float a = A0[idx];
	if (flags[idx - X] & MF_EMPTY) a -= ghostFluidHelper(idx, -X, phi, gfClamp);
	if (flags[idx + X] & MF_EMPTY) a -= ghostFluidHelper(idx, +X, phi, gfClamp);
	if (flags[idx - Y] & MF_EMPTY) a -= ghostFluidHelper(idx, -Y, phi, gfClamp);
	if (flags[idx + Y] & MF_EMPTY) a -= ghostFluidHelper(idx, +Y, phi, gfClamp);
	if (d.is3d) {
		if (flags[idx - Z] & MF_EMPTY) a -= ghostFluidHelper(idx, -Z, phi, gfClamp);
		if (flags[idx + Z] & MF_EMPTY) a -= ghostFluidHelper(idx, +Z, phi, gfClamp);
	}
	A0[idx] = a;
}

// knCorrectVelocity, plugin/pressure.cpp:87-109
__global__ void __launch_bounds__(BLOCK)
k_correct_velocity(Dim d, const int32_t* __restrict__ flags, float* __restrict__ vel, const float* __restrict__ p) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t X = 1, Y = d.Y, Z = d.Z, n = d.n;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
	const int f = flags[idx];
	if (f & MF_FLUID) {
		const float pc = p[idx];
		float x = vx[idx], y = vy[idx], z = d.is3d ? vz[idx] : 0.f;
		const int fxm = flags[idx - X], fym = flags[idx - Y], fzm = d.is3d ? flags[idx - Z] : 0;
		if (fxm & MF_FLUID) x -= (pc - p[idx - X]);
		if (fym & MF_FLUID) y -= (pc - p[idx - Y]);
		if (d.is3d && (fzm & MF_FLUID)) z -= (pc - p[idx - Z]);
		if (fxm & MF_EMPTY) x -= pc;
		if (fym & MF_EMPTY) y -= pc;
		if (d.is3d && (fzm & MF_EMPTY)) z -= pc;
		vx[idx] = x;
		vy[idx] = y;
		if (d.is3d) vz[idx] = z;
	} else if ((f & MF_EMPTY) && !(f & MF_OUTFLOW)) {
		if (flags[idx - X] & MF_FLUID) vx[idx] += p[idx - X]; else vx[idx] = 0.f;
		if (flags[idx - Y] & MF_FLUID) vy[idx] += p[idx - Y]; else vy[idx] = 0.f;
		if (d.is3d) {
			if (flags[idx - Z] & MF_FLUID) vz[idx] += p[idx - Z]; else vz[idx] = 0.f;
		}
	}
}

// knCorrectVelocityGhostFluid, plugin/pressure.cpp:154-187
__global__ void __launch_bounds__(BLOCK)
k_correct_velocity_gf(Dim d, float* __restrict__ vel, const int32_t* __restrict__ flags, const float* __restrict__ p,
                      const float* __restrict__ phi, float gfClamp, const float* __restrict__ curv, float surfTens) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t X = 1, Y = d.Y, Z = d.Z, n = d.n;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
	const int f = flags[idx];
	const bool fl = (f & MF_FLUID) != 0, emp = (f & MF_EMPTY) && !(f & MF_OUTFLOW);
	if (fl) {
		if (flags[idx - X] & MF_EMPTY) vx[idx] += p[idx] * ghostFluidHelper(idx, -X, phi, gfClamp);
		if (flags[idx - Y] & MF_EMPTY) vy[idx] += p[idx] * ghostFluidHelper(idx, -Y, phi, gfClamp);
		if (d.is3d && (flags[idx - Z] & MF_EMPTY)) vz[idx] += p[idx] * ghostFluidHelper(idx, -Z, phi, gfClamp);
	} else if (emp) {
		if (flags[idx - X] & MF_FLUID) vx[idx] -= p[idx - X] * ghostFluidHelper(idx - X, +X, phi, gfClamp); else vx[idx] = 0.f;
		if (flags[idx - Y] & MF_FLUID) vy[idx] -= p[idx - Y] * ghostFluidHelper(idx - Y, +Y, phi, gfClamp); else vy[idx] = 0.f;
		if (d.is3d) {
			if (flags[idx - Z] & MF_FLUID) vz[idx] -= p[idx - Z] * ghostFluidHelper(idx - Z, +Z, phi, gfClamp); else vz[idx] = 0.f;
		}
	}
	if (curv) {
		if (fl) {
			if (flags[idx - X] & MF_EMPTY) vx[idx] += surfTensHelper(idx, -X, phi, curv, surfTens, gfClamp);
			if (flags[idx - Y] & MF_EMPTY) vy[idx] += surfTensHelper(idx, -Y, phi, curv, surfTens, gfClamp);
			if (d.is3d && (flags[idx - Z] & MF_EMPTY)) vz[idx] += surfTensHelper(idx, -Z, phi, curv, surfTens, gfClamp);
		} else if (emp) {
			vx[idx] -= (flags[idx - X] & MF_FLUID) ? surfTensHelper(idx - X, +X, phi, curv, surfTens, gfClamp) : 0.f;
			vy[idx] -= (flags[idx - Y] & MF_FLUID) ? surfTensHelper(idx - Y, +Y, phi, curv, surfTens, gfClamp) : 0.f;
			if (d.is3d) vz[idx] -= (flags[idx - Z] & MF_FLUID) ? surfTensHelper(idx - Z, +Z, phi, curv, surfTens, gfClamp) : 0.f;
		}
	}
}

// knReplaceClampedGhostFluidVels, plugin/pressure.cpp:198-214.  Writes touch empty cells' components, reads
// touch fluid cells' components of the same grid: disjoint, so in-place is race free.
__global__ void __launch_bounds__(BLOCK)
k_replace_clamped_gf(Dim d, float* __restrict__ vel, const int32_t* __restrict__ flags, const float* __restrict__ phi, float gfClamp) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	if (!(flags[idx] & MF_EMPTY)) return;
	const int64_t X = 1, Y = d.Y, Z = d.Z, n = d.n;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
	if ((flags[idx - X] & MF_FLUID) && ghostFluidWasClamped(idx - X, +X, phi, gfClamp)) vx[idx] = vx[idx - X];
	if ((flags[idx - Y] & MF_FLUID) && ghostFluidWasClamped(idx - Y, +Y, phi, gfClamp)) vy[idx] = vy[idx - Y];
	if (d.is3d && (flags[idx - Z] & MF_FLUID) && ghostFluidWasClamped(idx - Z, +Z, phi, gfClamp)) vz[idx] = vz[idx - Z];
	if ((flags[idx + X] & MF_FLUID) && ghostFluidWasClamped(idx + X, -X, phi, gfClamp)) vx[idx] = vx[idx + X];
	if ((flags[idx + Y] & MF_FLUID) && ghostFluidWasClamped(idx + Y, -Y, phi, gfClamp)) vy[idx] = vy[idx + Y];
	if (d.is3d && (flags[idx + Z] & MF_FLUID) && ghostFluidWasClamped(idx + Z, -Z, phi, gfClamp)) vz[idx] = vz[idx + Z];
}

// fixPressure, plugin/pressure.cpp:226-246 (a handful of scalar updates: one thread)
__global__ void k_fix_pressure(Dim d, int64_t p, float value, float* rhs, float* A0, float* Ai, float* Aj, float* Ak) {
	const int64_t X = 1, Y = d.Y, Z = d.Z;
	rhs[p + X] -= Ai[p] * value;
	rhs[p + Y] -= Aj[p] * value;
	rhs[p - X] -= Ai[p - X] * value;
	rhs[p - Y] -= Aj[p - Y] * value;
	if (d.is3d) {
		rhs[p + Z] -= Ak[p] * value;
		rhs[p - Z] -= Ak[p - Z] * value;
	}
	rhs[p] = value;
	A0[p] = 1.f;
	Ai[p] = Aj[p] = Ak[p] = 0.f;
	Ai[p - X] = 0.f;
	Aj[p - Y] = 0.f;
	if (d.is3d) Ak[p - Z] = 0.f;
}

// =========================================================================================================
// MIC(0) preconditioner, conjugategrad.cpp:66-97 (init) and :135-159 (apply).
//
// The reference runs these as single-threaded lexicographic sweeps with a (i-1, j-1, k-1) dependency
// (forward) / (i+1, j+1, k+1) (backward).  Here the grid is cut into 8x8x8 tiles; tiles on one hyperplane
// ti+tj+tk = L are independent and run as one launch (one 64-lane wave per tile); inside a tile the wave
// walks the 22 cell hyperplanes, lane = one x-row (lj,lk), neighbour values move by wave shuffles and the
// per-cell coefficients are staged in LDS.  Per-cell arithmetic is exactly the reference's expression, so the
// result is bit-identical to the serial sweep.
//   MODE 0: init   dst := Aprecond,  var1 := A0
//   MODE 1: forward substitution     dst := tmp, var1 := residual
//   MODE 2: backward substitution (tile and in-tile coordinates mirrored)
// =========================================================================================================
// load 8 consecutive floats of one x-row (logical order a = 0..7 <-> physical li); branch-free so that the
// compiler issues every load of a tile before the first wait.  `nv` = number of in-domain cells of the row (0..8).
template <bool VEC, bool REV>
__device__ __forceinline__ void load_row8(const float* __restrict__ base, int64_t rowidx, int nv, float out[8]) {
	float t[8];
	if (VEC) {
		// x0 % 8 == 0 and sx % 4 == 0: both halves are 16-byte aligned; a half is either fully inside or outside
		const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
		const int64_t i0 = nv > 0 ? rowidx : 0, i1 = nv > 4 ? rowidx + 4 : 0;
		float4 lo = *(const float4*)(base + i0), hi = *(const float4*)(base + i1);
		if (nv <= 0) lo = z;
		if (nv <= 4) hi = z;
		t[0] = lo.x; t[1] = lo.y; t[2] = lo.z; t[3] = lo.w;
		t[4] = hi.x; t[5] = hi.y; t[6] = hi.z; t[7] = hi.w;
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++) {
			const float v = base[e < nv ? rowidx + e : 0];
			t[e] = e < nv ? v : 0.f;
		}
	}
#pragma unroll
	for (int a = 0; a < 8; a++) out[a] = t[REV ? 7 - a : a];
}
template <bool VEC, bool REV>
__device__ __forceinline__ void load_row8i(const int32_t* __restrict__ base, int64_t rowidx, int nv, int out[8]) {
	float t[8];
	load_row8<VEC, REV>((const float*)base, rowidx, nv, t);
#pragma unroll
	for (int a = 0; a < 8; a++) out[a] = __float_as_int(t[a]);
}

template <int MODE, bool VEC>
__global__ void __launch_bounds__(64)
k_mic_tiles(Dim d, int level, int nti, int ntj, int ntk, const int32_t* __restrict__ flags, float* __restrict__ dst,
            const float* __restrict__ var1, const float* __restrict__ Ap, const float* __restrict__ Ai,
            const float* __restrict__ Aj, const float* __restrict__ Ak, const CgScalars* __restrict__ sc) {
	constexpr bool REV = (MODE == 2);
	constexpr int NC = (MODE == 0) ? 2 : 1;  // values handed to each neighbour
	const int tjl = blockIdx.x, tkl = blockIdx.y;
	const int til = level - tjl - tkl;
	if (til < 0 || til >= nti) return;
	const int ti = REV ? nti - 1 - til : til, tj = REV ? ntj - 1 - tjl : tjl, tk = REV ? ntk - 1 - tkl : tkl;
	const int lane = threadIdx.x, b = lane & 7, c = lane >> 3;
	const int lj = REV ? 7 - b : b, lk = REV ? 7 - c : c;
	const int x0 = ti * 8, j = tj * 8 + lj, k = tk * 8 + lk;
	const bool row_in = (j < d.sy) && (k < d.sz);
	const int64_t rowbase = (int64_t)x0 + d.Y * j + d.Z * k;
	const int back = REV ? 1 : -1;  // physical offset of the logical predecessor
	const int nvx = d.sx - x0 < 8 ? d.sx - x0 : 8;  // in-domain cells of a row of this tile
	const int nv = row_in ? nvx : 0;

	__shared__ float sV[512], sAi[512], sAj[512], sAk[512], sP[512], sD[512];
	__shared__ int sF[512];
	__shared__ float sHj[NC][64], sHk[NC][64];

	// ---- issue every global load of the tile (own row, i-halo cell, j-/k-halo rows), then consume ----
	int rF[8];
	float rV[8], rAi[8], rAj[8], rAk[8], rP[8], rD[8];
	load_row8i<VEC, REV>(flags, rowbase, nv, rF);
	load_row8<VEC, REV>(var1, rowbase, nv, rV);
	load_row8<VEC, REV>(Ai, rowbase, nv, rAi);
	load_row8<VEC, REV>(Aj, rowbase, nv, rAj);
	load_row8<VEC, REV>(Ak, rowbase, nv, rAk);
	if (MODE != 0) {
		load_row8<VEC, REV>(Ap, rowbase, nv, rP);
		load_row8<VEC, REV>(dst, rowbase, nv, rD);
	}
	// i-halo: the cell before my row (one per lane)
	const int gi = x0 + (REV ? 8 : -1);
	const bool hin = row_in && gi >= 0 && gi < d.sx;
	const int64_t hidx = hin ? rowbase + (REV ? 8 : -1) : 0;
	const float hAi = Ai[hidx], hAj = Aj[hidx], hAk = Ak[hidx], hD = dst[hidx];
	const float hP = (MODE == 1) ? Ap[hidx] : 0.f;
	// j-halo row (used by lanes with b == 0) and k-halo row (lanes with c == 0)
	const int jn = j + back, kn = k + back;
	const int nvj = ((b == 0) && (jn >= 0) && (jn < d.sy) && (k < d.sz)) ? nvx : 0;
	const int nvk = ((c == 0) && (kn >= 0) && (kn < d.sz) && (j < d.sy)) ? nvx : 0;
	const int64_t jrow = rowbase + (int64_t)back * d.Y, krow = rowbase + (int64_t)back * d.Z;
	float jD[8], jA[8], jB[8], jC[8], jP[8], kD[8], kA[8], kB[8], kC[8], kP[8];
	if (b == 0) {
		load_row8<VEC, REV>(dst, jrow, nvj, jD);
		if (MODE != 2) load_row8<VEC, REV>(Aj, jrow, nvj, jA);
		if (MODE == 1) load_row8<VEC, REV>(Ap, jrow, nvj, jP);
		if (MODE == 0) {
			load_row8<VEC, REV>(Ai, jrow, nvj, jB);
			load_row8<VEC, REV>(Ak, jrow, nvj, jC);
		}
	}
	if (c == 0) {
		load_row8<VEC, REV>(dst, krow, nvk, kD);
		if (MODE != 2) load_row8<VEC, REV>(Ak, krow, nvk, kA);
		if (MODE == 1) load_row8<VEC, REV>(Ap, krow, nvk, kP);
		if (MODE == 0) {
			load_row8<VEC, REV>(Ai, krow, nvk, kB);
			load_row8<VEC, REV>(Aj, krow, nvk, kC);
		}
	}

	// value(s) a finished neighbour cell hands to its logical successor in direction `dir`:
	//   MODE 1: (dst*A_dir)*Ap   MODE 2: dst   MODE 0: square(A_dir*Ap), A_dir*(A_o1+A_o2)*square(Ap)
	auto hand = [&](float dv, float adir, float osum, float ap, float& h0, float& h1) {
		h1 = 0.f;
		if (MODE == 1) {
			h0 = (dv * adir) * ap;
		} else if (MODE == 2) {
			h0 = dv;
		} else {
			const float t = adir * dv;  // dv = Aprecond of the neighbour (being built)
			h0 = t * t;
			h1 = adir * osum * (dv * dv);
		}
	};
#pragma unroll
	for (int a = 0; a < 8; a++) {
		const int s = lane * 8 + a;
		const bool in = a < 8 && ((REV ? 7 - a : a) < nv);
		const int fl = in ? ((rF[a] & MF_FLUID) ? 1 : 2) : 0;  // 1 fluid, 2 in-domain non-fluid, 0 outside
		sF[s] = fl;
		sV[s] = (fl == 1) ? rV[a] : 0.f;
		sAi[s] = rAi[a];
		sAj[s] = rAj[a];
		sAk[s] = rAk[a];
		if (MODE == 0) {
			sP[s] = 0.f;
			sD[s] = 0.f;  // Aprecond.clear(): non-fluid cells stay 0
		} else {
			sP[s] = rP[a];
			sD[s] = rD[a];
		}
	}
	float hi0, hi1;
	hand(hin ? hD : 0.f, hin ? hAi : 0.f, hAj + hAk, hP, hi0, hi1);
	if (!hin) hi0 = hi1 = 0.f;
	if (b == 0) {
#pragma unroll
		for (int a = 0; a < 8; a++) {
			float h0, h1;
			hand(jD[a], MODE != 2 ? jA[a] : 0.f, MODE == 0 ? (jB[a] + jC[a]) : 0.f, MODE == 1 ? jP[a] : 0.f, h0, h1);
			sHj[0][c * 8 + a] = h0;
			if (NC == 2) sHj[NC - 1][c * 8 + a] = h1;
		}
	}
	if (c == 0) {
#pragma unroll
		for (int a = 0; a < 8; a++) {
			float h0, h1;
			hand(kD[a], MODE != 2 ? kA[a] : 0.f, MODE == 0 ? (kB[a] + kC[a]) : 0.f, MODE == 1 ? kP[a] : 0.f, h0, h1);
			sHk[0][b * 8 + a] = h0;
			if (NC == 2) sHk[NC - 1][b * 8 + a] = h1;
		}
	}
	__syncthreads();
	if (sc && sc->done) return;   // checked after the loads were issued: one global round trip less per level

	// ---- 22 cell hyperplanes ----
	float oi0 = 0.f, oi1 = 0.f, oj0 = 0.f, oj1 = 0.f, ok0 = 0.f, ok1 = 0.f;
#pragma unroll 2
	for (int h = 0; h < 22; h++) {
		const int a = h - b - c;
		const bool valid = (a >= 0) && (a < 8);
		const int ac = a < 0 ? 0 : (a > 7 ? 7 : a);
		float ij0 = __shfl_up(oj0, 1, 64), ik0 = __shfl_up(ok0, 8, 64);
		float ij1 = 0.f, ik1 = 0.f;
		if (NC == 2) {
			ij1 = __shfl_up(oj1, 1, 64);
			ik1 = __shfl_up(ok1, 8, 64);
		}
		if (b == 0) {
			ij0 = sHj[0][c * 8 + ac];
			if (NC == 2) ij1 = sHj[NC - 1][c * 8 + ac];
		}
		if (c == 0) {
			ik0 = sHk[0][b * 8 + ac];
			if (NC == 2) ik1 = sHk[NC - 1][b * 8 + ac];
		}
		const float ii0 = (a == 0) ? hi0 : oi0;
		const float ii1 = (a == 0) ? hi1 : oi1;
		if (valid) {
			const int s = lane * 8 + ac;
			const int fl = sF[s];
			const float ai = sAi[s], aj = sAj[s], ak = sAk[s];
			if (MODE == 0) {
				float ap = 0.f;
				if (fl == 1) {
					const float a0 = sV[s];
					float e = a0 - ii0 - ij0 - ik0;
					const float s3 = ii1 + ij1 + ik1;
					// e -= tau * ( ... + 0. ): fp64 product and subtraction, conjugategrad.cpp:84-88
					const float tau = 0.97f;
					e = (float)((double)e - (double)tau * ((double)s3 + 0.));
					if (e < 0.25f * a0) e = a0;
					ap = (float)(1. / (double)sqrtf(e));
				}
				sD[s] = ap;
				const float ti_ = ai * ap, tj_ = aj * ap, tk_ = ak * ap;
				const float ap2 = ap * ap;
				oi0 = ti_ * ti_;
				oj0 = tj_ * tj_;
				ok0 = tk_ * tk_;
				oi1 = ai * (aj + ak) * ap2;
				oj1 = aj * (ai + ak) * ap2;
				ok1 = ak * (ai + aj) * ap2;
			} else if (MODE == 1) {
				const float p = sP[s];
				float val = sD[s];
				if (fl == 1) {
					val = p * (sV[s] - ii0 - ij0 - ik0);
					sD[s] = val;
				}
				oi0 = (val * ai) * p;
				oj0 = (val * aj) * p;
				ok0 = (val * ak) * p;
			} else {
				const float p = sP[s];
				float val = sD[s];
				if (fl == 1) {
					val = p * (val - ii0 * ai * p - ij0 * aj * p - ik0 * ak * p);
					sD[s] = val;
				}
				oi0 = oj0 = ok0 = val;
			}
		}
	}
	__syncthreads();
	// ---- write back my row (non-fluid cells carry their loaded value, so whole in-domain halves are stored) ----
	float w[8];
#pragma unroll
	for (int a = 0; a < 8; a++) w[REV ? 7 - a : a] = sD[lane * 8 + a];
	if (VEC) {
		if (nv > 0) *(float4*)(dst + rowbase) = make_float4(w[0], w[1], w[2], w[3]);
		if (nv > 4) *(float4*)(dst + rowbase + 4) = make_float4(w[4], w[5], w[6], w[7]);
	} else {
#pragma unroll
		for (int e = 0; e < 8; e++)
			if (e < nv) dst[rowbase + e] = w[e];
	}
}

// ---------------------------------------------------------------------------------------------------------
// MIC apply as ONE launch per sweep ("dataflow"): one wave per tile, tiles are handed out in hyperplane order by an
// atomic ticket (a ticketed tile's predecessors hold smaller tickets, i.e. they are already running or finished, so the
// wait below always ends whatever the dispatch order), and the three faces a tile hands to its +i/+j/+k successors
// travel as 8-byte {value, tag} granules written with ONE agent-scope (sc1, write-through) store each and polled with
// agent-scope (sc1, L1-bypassing) loads: no flag, no fence (MI355X_MICROARCH.md, "handoff-1to1").  tag = launch
// generation, so the exchange buffer never needs clearing.  Per-cell arithmetic is identical to k_mic_tiles.
// ---------------------------------------------------------------------------------------------------------
struct FlowCtl {
	int ticket, finished, err, pad;
};
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(unsigned long long* p, float v, unsigned tag) {
	const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
	__hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// same-XCD hand-off: a plain store reaches that XCD's L2, where an agent-scope (sc1) load of another CU of the same XCD
// finds it (0.26 us one way vs 0.72 us for sc1 -> sc1 across XCDs, tools/micro/pingpong.hip); NOT visible to other XCDs
__device__ __forceinline__ void granule_store_local(unsigned long long* p, float v, unsigned tag) {
	const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
	__hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
constexpr int FLOW_SPIN_LIMIT = 1 << 21;

template <int MODE, bool VEC>
__global__ void __launch_bounds__(64)
k_mic_flow(Dim d, int nti, int ntj, int ntk, int ntiles, const int* __restrict__ order, FlowCtl* ctl,
           unsigned long long* xch, unsigned gen, const int32_t* __restrict__ flags, float* __restrict__ dst,
           const float* __restrict__ var1, const float* __restrict__ Ap, const float* __restrict__ Ai,
           const float* __restrict__ Aj, const float* __restrict__ Ak, const CgScalars* __restrict__ sc) {
	static_assert(MODE == 1 || MODE == 2, "dataflow kernel implements the apply sweeps");
	constexpr bool REV = (MODE == 2);
	if (sc && sc->done) return;
	const int lane = threadIdx.x, b = lane & 7, c = lane >> 3;
	// double-buffered tile operands: while tile n runs its 22 steps out of one buffer, the operands of the tile this
	// wave will run next are already in flight (its ticket was drawn one tile earlier), so ticket, tile lookup and the
	// HBM round trip of the operands are off the dependency chain between tiles
	__shared__ float4 sA[2][512];   // {V, Ai, Aj, Ak}
	__shared__ float4 sB[2][512];   // {Aprecond, dst, fluid, -}
	__shared__ float sHj[64], sHk[64];
	const unsigned long long fresh0 = (unsigned long long)gen << 32;

	struct TileRegs {
		int F[8];
		float V[8], Ai[8], Aj[8], Ak[8], P[8], D[8];
	};
	auto tile_geom = [&](int packed, int& x0, int64_t& rowbase, int& nv) {
		const int til = packed & 1023, tjl = (packed >> 10) & 1023, tkl = (packed >> 20) & 1023;
		const int ti = REV ? nti - 1 - til : til, tj = REV ? ntj - 1 - tjl : tjl, tk = REV ? ntk - 1 - tkl : tkl;
		const int lj = REV ? 7 - b : b, lk = REV ? 7 - c : c;
		x0 = ti * 8;
		const int j = tj * 8 + lj, k = tk * 8 + lk;
		const bool row_in = (j < d.sy) && (k < d.sz);
		rowbase = (int64_t)x0 + d.Y * j + d.Z * k;
		const int nvx = d.sx - x0 < 8 ? d.sx - x0 : 8;
		nv = row_in ? nvx : 0;
	};
	auto issue = [&](TileRegs& r, int packed) {
		int x0, nv;
		int64_t rowbase;
		tile_geom(packed, x0, rowbase, nv);
		load_row8i<VEC, REV>(flags, rowbase, nv, r.F);
		load_row8<VEC, REV>(var1, rowbase, nv, r.V);
		load_row8<VEC, REV>(Ai, rowbase, nv, r.Ai);
		load_row8<VEC, REV>(Aj, rowbase, nv, r.Aj);
		load_row8<VEC, REV>(Ak, rowbase, nv, r.Ak);
		load_row8<VEC, REV>(Ap, rowbase, nv, r.P);
		load_row8<VEC, REV>(dst, rowbase, nv, r.D);
	};
	auto commit = [&](const TileRegs& r, int packed, int buf) {
		int x0, nv;
		int64_t rowbase;
		tile_geom(packed, x0, rowbase, nv);
#pragma unroll
		for (int a = 0; a < 8; a++) {
			const bool in = ((REV ? 7 - a : a) < nv);
			const bool fl = in && (r.F[a] & MF_FLUID);
			sA[buf][a * 64 + lane] = make_float4(fl ? r.V[a] : 0.f, r.Ai[a], r.Aj[a], r.Ak[a]);
			sB[buf][a * 64 + lane] = make_float4(r.P[a], r.D[a], fl ? 1.f : 0.f, 0.f);
		}
	};
	auto draw = [&]() {
		int t = 0;
		if (lane == 0) t = atomicAdd(&ctl->ticket, 1);
		return __builtin_amdgcn_readfirstlane(t);
	};

	int t_cur = draw();
	int t_nxt = draw();
	int pk_cur = (t_cur < ntiles) ? order[t_cur] : 0;
	int pk_nxt = (t_nxt < ntiles) ? order[t_nxt] : 0;
	TileRegs R;
	if (t_cur < ntiles) {
		issue(R, pk_cur);
		commit(R, pk_cur, 0);
	}
	int buf = 0, spins = 0;
	while (t_cur < ntiles) {
		int tnn_raw = 0;           // ticket of the tile after next: drawn now, looked at when this tile is done
		if (lane == 0) tnn_raw = atomicAdd(&ctl->ticket, 1);
		// ---- geometry of the current tile ----
		const int til = pk_cur & 1023, tjl = (pk_cur >> 10) & 1023, tkl = (pk_cur >> 20) & 1023;
		const int64_t tid = ((int64_t)tkl * ntj + tjl) * nti + til;
		const bool has_pi = til > 0, has_pj = (tjl > 0) && (b == 0), has_pk = (tkl > 0) && (c == 0);
		const bool has_si = til + 1 < nti, has_sj = (tjl + 1 < ntj) && (b == 7), has_sk = (tkl + 1 < ntk) && (c == 7);
		unsigned long long* out_i = xch + ((tid * 3 + 0) << 6) + lane;
		unsigned long long* out_j = xch + ((tid * 3 + 1) << 6) + c * 8;
		unsigned long long* out_k = xch + ((tid * 3 + 2) << 6) + b * 8;
		const unsigned long long* in_i = xch + (((tid - 1) * 3 + 0) << 6) + lane;
		const unsigned long long* in_j = xch + (((tid - nti) * 3 + 1) << 6) + c * 8;
		const unsigned long long* in_k = xch + (((tid - (int64_t)nti * ntj) * 3 + 2) << 6) + b * 8;
		// ---- first look at the predecessors' faces, then the next tile's operands (both stay in flight) ----
		unsigned long long gi = has_pi ? granule_load(in_i) : fresh0;
		unsigned long long gj[8], gk[8];
#pragma unroll
		for (int a = 0; a < 8; a++) gj[a] = has_pj ? granule_load(in_j + a) : fresh0;
#pragma unroll
		for (int a = 0; a < 8; a++) gk[a] = has_pk ? granule_load(in_k + a) : fresh0;
		if (t_nxt < ntiles) issue(R, pk_nxt);
		// ---- wait until every face value is there ----
		for (;;) {
			bool ok = ((unsigned)(gi >> 32) == gen);
#pragma unroll
			for (int a = 0; a < 8; a++) ok = ok && ((unsigned)(gj[a] >> 32) == gen) && ((unsigned)(gk[a] >> 32) == gen);
			if (ok || ++spins > FLOW_SPIN_LIMIT) break;
			__builtin_amdgcn_s_sleep(1);
			if ((unsigned)(gi >> 32) != gen) gi = granule_load(in_i);
#pragma unroll
			for (int a = 0; a < 8; a++) {
				if ((unsigned)(gj[a] >> 32) != gen) gj[a] = granule_load(in_j + a);
				if ((unsigned)(gk[a] >> 32) != gen) gk[a] = granule_load(in_k + a);
			}
		}
		if (b == 0) {
#pragma unroll
			for (int a = 0; a < 8; a++) sHj[c * 8 + a] = __uint_as_float((unsigned)gj[a]);
		}
		if (c == 0) {
#pragma unroll
			for (int a = 0; a < 8; a++) sHk[b * 8 + a] = __uint_as_float((unsigned)gk[a]);
		}
		const float hi0 = __uint_as_float((unsigned)gi);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// ---- 22 cell hyperplanes (no global loads inside) ----
		float oi0 = 0.f, oj0 = 0.f, ok0 = 0.f;
		int a = -b - c;
		int ac = 0;
		float4 nA = sA[buf][lane], nB = sB[buf][lane];
		float nHj = sHj[c * 8], nHk = sHk[b * 8];
#pragma unroll 2
		for (int h = 0; h < 22; h++) {
			const float4 cA = nA, cB = nB;
			const float hj = nHj, hk = nHk;
			const int cc = ac;
			{
				const int a1 = a + 1;
				ac = a1 < 0 ? 0 : (a1 > 7 ? 7 : a1);
				nA = sA[buf][ac * 64 + lane];
				nB = sB[buf][ac * 64 + lane];
				nHj = sHj[c * 8 + ac];
				nHk = sHk[b * 8 + ac];
			}
			const float dj = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(oj0), 0x111, 0xf, 0xf, false));
			const float sk = __shfl_up(ok0, 8, 64);
			const float ij0 = (b == 0) ? hj : dj;
			const float ik0 = (c == 0) ? hk : sk;
			const float ii0 = (a == 0) ? hi0 : oi0;
			const bool valid = (unsigned)a < 8u;
			const float ai = cA.y, aj = cA.z, ak = cA.w, p = cB.x;
			const bool fl = valid && (cB.z != 0.f);
			float val = cB.y;
			if (MODE == 1) {
				const float nv = p * (cA.x - ii0 - ij0 - ik0);
				val = fl ? nv : val;
				oi0 = (val * ai) * p;
				oj0 = (val * aj) * p;
				ok0 = (val * ak) * p;
			} else {
				const float nv = p * (val - ii0 * ai * p - ij0 * aj * p - ik0 * ak * p);
				val = fl ? nv : val;
				oi0 = oj0 = ok0 = val;
			}
			if (valid) {
				sB[buf][cc * 64 + lane].y = val;
				if (has_si && a == 7) granule_store(out_i, oi0, gen);
				if (has_sj) granule_store(out_j + a, oj0, gen);
				if (has_sk) granule_store(out_k + a, ok0, gen);
			}
			a++;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// ---- write back the finished tile ----
		{
			int x0, nv;
			int64_t rowbase;
			tile_geom(pk_cur, x0, rowbase, nv);
			float w[8];
#pragma unroll
			for (int e = 0; e < 8; e++) w[REV ? 7 - e : e] = sB[buf][e * 64 + lane].y;
			if (VEC) {
				if (nv > 0) *(float4*)(dst + rowbase) = make_float4(w[0], w[1], w[2], w[3]);
				if (nv > 4) *(float4*)(dst + rowbase + 4) = make_float4(w[4], w[5], w[6], w[7]);
			} else {
#pragma unroll
				for (int e = 0; e < 8; e++)
					if (e < nv) dst[rowbase + e] = w[e];
			}
		}
		// ---- land the next tile's operands in the other buffer, rotate ----
		if (t_nxt < ntiles) commit(R, pk_nxt, buf ^ 1);
		const int t_nn = __builtin_amdgcn_readfirstlane(tnn_raw);
		buf ^= 1;
		t_cur = t_nxt;
		pk_cur = pk_nxt;
		t_nxt = t_nn;
		pk_nxt = (t_nn < ntiles) ? order[t_nn] : 0;
	}
	if (spins > FLOW_SPIN_LIMIT) atomicExch(&ctl->err, 1);
	// the last workgroup to leave re-arms the ticket for the next sweep (visible at the kernel boundary)
	if (lane == 0) {
		const int f = atomicAdd(&ctl->finished, 1);
		if (f == (int)gridDim.x - 1) {
			ctl->ticket = 0;
			ctl->finished = 0;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// MIC apply, row-streaming form ("rows"): a workgroup of TWO waves owns an 8x8 bundle of x-rows (tj,tk) and streams
// along the whole x extent.
//   compute wave: lane (b,c) works on cell x' = h - b - c at step h.  The i-dependency never leaves the lane's
//     registers (no i-faces, no 14-step fill/drain per 8 cells), the j/k dependencies inside the bundle move by DPP /
//     ds_bpermute as in the tile kernel, and only the two outer faces of the bundle cross workgroups -- as tagged
//     8-byte sc1 granules, one per (x', face lane), polled 8 steps at a time in a lane-relative window (a consumer
//     bundle runs 15 steps behind its producer).  Its only vector-memory traffic is those granules: vmcnt is in-order,
//     so a polling load must never queue behind an HBM fetch.
//   memory wave: fetches the operands of the bundle's rows in 8-cell chunks, three chunks ahead, commits them to a
//     32-step LDS ring and writes finished chunks back to dst.  The ring is indexed by the STEP at which a lane
//     consumes the cell ((h+2) & 31), i.e. the skew b+c is applied when the memory wave stores, and every lane of the
//     compute wave reads the same ring row at a given step: immediate LDS offsets, no per-lane address arithmetic.
//   The two waves meet only through two LDS counters (chunks committed / blocks finished).
// Bundles are ticketed in anti-diagonal order (tj+tk): a workgroup only ever waits for bundles drawn before its own.
// 256^3: 1024 bundles, 62 bundle hops per sweep.  Per-cell arithmetic = k_mic_tiles = the reference's.
// ---------------------------------------------------------------------------------------------------------
constexpr int ROWS_PAD = 16;
struct RowsChunk {
	int F[8];
	float V[8], Ai[8], Aj[8], Ak[8], P[8], D[8];
};
constexpr int ROWS_THREADS = 384;   // wave 0 computes, 1-3 load (chunk n -> wave 1 + n % 3), 4 writes back, 5 polls the faces
template <int MODE, bool VEC>
__global__ void __launch_bounds__(ROWS_THREADS)
k_mic_rows(Dim d, int nbj, int nbk, int jb, int nstreams, int nchunks, const int* __restrict__ order, FlowCtl* ctl, int* xt,
           unsigned long long* xj, unsigned long long* xk, unsigned gen, const int32_t* __restrict__ flags,
           float* __restrict__ dst, const float* __restrict__ var1, const float* __restrict__ Ap,
           const float* __restrict__ Ai, const float* __restrict__ Aj, const float* __restrict__ Ak,
           const CgScalars* __restrict__ sc, long long* trace, int trace_ticket, int trace_ticket2) {
	static_assert(MODE == 1 || MODE == 2, "row-streaming kernel implements the apply sweeps");
	constexpr bool REV = (MODE == 2);
	if (sc && sc->done) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane & 7, c = lane >> 3;
	const int skew = b + c;
	// the compute wave is the critical path: everything else yields to it
	if (wave == 0) __builtin_amdgcn_s_setprio(3);
	else __builtin_amdgcn_s_setprio(0);
	__shared__ float4 sA[32 * 64];   // {fluid ? rhs : dst  (-> result), Ai, Aj, Ak}     index = ((h + 2) & 31) * 64 + lane
	__shared__ float2 sB[32 * 64];   // {Aprecond, fluid}
	__shared__ __attribute__((aligned(16))) float sFj[2][8][8];
	__shared__ __attribute__((aligned(16))) float sFk[2][8][8];   // face values of a block [block parity][face lane][step]
	__shared__ int s_ready[3], s_done, s_flushed, s_faces, s_ticket;
	const unsigned long long fresh0 = (unsigned long long)gen << 32;
	const int X8 = nchunks * 8;
	int spins = 0;

	for (;;) {
		if (threadIdx.x == 0) {
			// xt[0..7] tickets, xt[8..16] bounds of the per-XCD queues inside `order`, xt[17] = number of queues (1: one
			// global queue; 8: bundles are queued per XCD by k-slab, so that most faces are handed over inside one XCD's L2)
			const int nq = xt[17];
			const int q = nq > 1 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) : 0;   // HW_REG_XCC_ID
			const int lo = xt[8 + q], hi = xt[9 + q];
			int tk = nstreams;
			if (lo < hi) {
				const int tl = atomicAdd(&xt[q], 1);
				if (lo + tl < hi) tk = lo + tl;
			}
			s_ticket = tk;
			s_ready[0] = s_ready[1] = s_ready[2] = 0;
			s_done = 0;
			s_flushed = 0;
			s_faces = 0;
		}
		__syncthreads();
		const int t = s_ticket;
		if (t >= nstreams) break;
		const int pk = order[t];
		const int tjl = pk & 0xffff, tkl = pk >> 16;
		const int tj = REV ? nbj - 1 - tjl : tjl, tk = REV ? nbk - 1 - tkl : tkl;
		const int j = tj * 8 + (REV ? 7 - b : b), k = tk * 8 + (REV ? 7 - c : c);
		const bool row_in = (j < d.sy) && (k < d.sz);
		const int64_t rowbase = d.Y * j + d.Z * k;

		// face granules: per bundle and face an array [producer step h + 2][face lane]: the 8 face lanes of one step (one
		// store instruction) fill exactly one 64-byte line, and a consumer lane's 8-step window maps to 8 consecutive lines
		// (lane (7,c) publishes x' at step x'+7+c, lane (b,7) at x'+b+7: the same steps 8m+5 .. 8m+12 for every face lane)
		// jb = bundles per independent j-block (mf_set_mic_blocking: the caller has zeroed the Aj coupling across block
		// faces, so nothing crosses them); jb == nbj: one block = the reference algorithm
		const int tj_pred = REV ? tj + 1 : tj - 1, tj_succ = REV ? tj - 1 : tj + 1;
		const bool has_pj = (tjl > 0) && (tj / jb == tj_pred / jb) && (b == 0), has_pk = (tkl > 0) && (c == 0);
		const bool has_sj = (tjl + 1 < nbj) && (tj / jb == tj_succ / jb) && (b == 7), has_sk = (tkl + 1 < nbk) && (c == 7);
		const int64_t sid = (int64_t)tkl * nbj + tjl;
		const int64_t XP = X8 + 2 * ROWS_PAD;
		if (wave == 5) {
			// ================= face poller: the only wave that loads granules (and it never stores to global memory) =====
			const unsigned long long* in_j = xj + (sid - 1) * XP * 8 + c;          // + (h + 2) * 8
			const unsigned long long* in_k = xk + (sid - nbj) * XP * 8 + b;
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				const int xq = 8 * m - 2 - skew;
				unsigned long long gj[8], gk[8];
#pragma unroll
				for (int a = 0; a < 8; a++) gj[a] = gk[a] = fresh0;
				for (;;) {
					if (has_pj) {
#pragma unroll
						for (int a = 0; a < 8; a++) gj[a] = granule_load(in_j + (int64_t)(8 * m + 7 + a) * 8);
					}
					if (has_pk) {
#pragma unroll
						for (int a = 0; a < 8; a++) gk[a] = granule_load(in_k + (int64_t)(8 * m + 7 + a) * 8);
					}
					// tags only grow: the window is complete when its smallest tag is this sweep's generation
					unsigned tmin = gen;
#pragma unroll
					for (int a = 0; a < 8; a++) {
						const bool in = (unsigned)(xq + a) < (unsigned)X8;
						const unsigned tj_ = (unsigned)(gj[a] >> 32), tk_ = (unsigned)(gk[a] >> 32);
						tmin = min(tmin, in ? min(tj_, tk_) : gen);
					}
					if (__all(tmin == gen) || ++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				// the buffer of this parity was read by block m-2
				if (m >= 2) {
					while (__hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < m - 1) {
						if (++spins > FLOW_SPIN_LIMIT) break;
						__builtin_amdgcn_s_sleep(8);
					}
				}
				if (b == 0) {
#pragma unroll
					for (int a = 0; a < 8; a++) sFj[m & 1][c][a] = __uint_as_float((unsigned)gj[a]);
				}
				if (c == 0) {
#pragma unroll
					for (int a = 0; a < 8; a++) sFk[m & 1][b][a] = __uint_as_float((unsigned)gk[a]);
				}
				__hip_atomic_store(&s_faces, m + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else if (wave != 0) {
			// ================= memory waves: waves 1-3 load + commit, wave 4 writes finished chunks back =================
			// (separate waves because vmcnt is one in-order counter per wave: a wave that has loads of several chunks or
			// loads and stores in flight ends up waiting for all of them)
			auto chunk_geom = [&](int m, int64_t& rowidx, int& nv) {
				const int x0 = (REV ? nchunks - 1 - m : m) * 8;
				const int nvx = d.sx - x0 < 8 ? d.sx - x0 : 8;
				nv = row_in ? nvx : 0;
				rowidx = rowbase + x0;
			};
			auto wait_for = [&](int* flag, int need) {
				while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(8);
				}
			};
			if (wave <= 3) {
				auto issue = [&](RowsChunk& r, int m) {
					int64_t rowidx;
					int nv;
					chunk_geom(m, rowidx, nv);
					load_row8i<VEC, REV>(flags, rowidx, nv, r.F);
					if (MODE == 1) load_row8<VEC, REV>(var1, rowidx, nv, r.V);
					load_row8<VEC, REV>(Ai, rowidx, nv, r.Ai);
					load_row8<VEC, REV>(Aj, rowidx, nv, r.Aj);
					load_row8<VEC, REV>(Ak, rowidx, nv, r.Ak);
					load_row8<VEC, REV>(Ap, rowidx, nv, r.P);
					load_row8<VEC, REV>(dst, rowidx, nv, r.D);
				};
				auto commit = [&](const RowsChunk& r, int m) {
					int64_t rowidx;
					int nv;
					chunk_geom(m, rowidx, nv);
					const int p0 = 8 * m + skew + 2;
#pragma unroll
					for (int a = 0; a < 8; a++) {
						const bool in = ((REV ? 7 - a : a) < nv);
						const bool fl = in && (r.F[a] & MF_FLUID);
						const int idx = ((p0 + a) & 31) * 64 + lane;
						sA[idx] = make_float4((MODE == 1 && fl) ? r.V[a] : r.D[a], r.Ai[a], r.Aj[a], r.Ak[a]);
						sB[idx] = make_float2(r.P[a], fl ? 1.f : 0.f);
					}
				};
				// one chunk in flight per loader wave: its loads are issued as soon as the previous chunk of this wave is
				// committed (three blocks before the compute wave needs them), then the wave waits for the ring rows
				RowsChunk R;
				const int w = wave - 1;
#pragma unroll 1
				for (int n = w; n < nchunks; n += 3) {
					issue(R, n);
					// the ring rows of chunk n were last used by chunk n-4: it must have been written back
					if (n >= 4) wait_for(&s_flushed, n - 3);
					commit(R, n);
					__hip_atomic_store(&s_ready[w], n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			} else {
#pragma unroll 1
				for (int q = 0; q < nchunks; q++) {
					wait_for(&s_done, q + 3);   // chunk q is complete once block q+2 is finished
					int64_t rowidx;
					int nv;
					chunk_geom(q, rowidx, nv);
					const int p0 = 8 * q + skew + 2;
					float w[8];
#pragma unroll
					for (int e = 0; e < 8; e++) w[REV ? 7 - e : e] = sA[((p0 + e) & 31) * 64 + lane].x;
					__hip_atomic_store(&s_flushed, q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					if (VEC) {
						if (nv > 0) *(float4*)(dst + rowidx) = make_float4(w[0], w[1], w[2], w[3]);
						if (nv > 4) *(float4*)(dst + rowidx + 4) = make_float4(w[4], w[5], w[6], w[7]);
					} else {
#pragma unroll
						for (int e = 0; e < 8; e++)
							if (e < nv) dst[rowidx + e] = w[e];
					}
				}
			}
		} else {
			// ================= compute wave: LDS in, LDS + face granules out =================
			// Lane 0 stands in for lane 63's k face (same step, but its own x' runs 14 ahead of lane 63's)
			const bool corner_proxy = (lane == 0) && (tkl + 1 < nbk);
			const bool face_lane = has_sj || (has_sk && lane != 63) || corner_proxy;
			const int fskew = corner_proxy ? -14 : 0;
			unsigned long long* out_f = has_sj ? xj + sid * XP * 8 + c : (corner_proxy ? xk + sid * XP * 8 + 7 : xk + sid * XP * 8 + b);
			float oi0 = 0.f, oj0 = 0.f, ok0 = 0.f;
			float4 nA = sA[lane];                  // ring row of h = -2 (never valid)
			float2 nB = sB[lane];
			const bool tr = trace && (t == trace_ticket || t == trace_ticket2) && lane == 0;
			long long* trb = trace + (t == trace_ticket2 ? 8 * 4096 : 0);
			if (trace && lane == 0 && t < 4096) {
				trace[4 * 4096 + 2 * t] = wall_clock64();
				trace[6 * 4096 + t] = ((long long)blockIdx.x << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
			}
#define ROWS_TRACE(i) if (tr) trb[m * 4 + (i)] = wall_clock64();
			// plain (XCD-local) face stores only when every consumer of this bundle's faces runs on this XCD
			const int nq_ = xt[17];
			const bool local_faces = nq_ > 1 && ((tkl + 1 >= nbk) || ((tkl * nq_) / nbk == ((tkl + 1) * nq_) / nbk));
			auto block = [&](int m, auto edge_tag, auto local_tag) {
				constexpr bool EDGE = decltype(edge_tag)::value;
				constexpr bool LOCAL = decltype(local_tag)::value;
				const int xq = 8 * m - 2 - skew;                       // this lane's x' at the first step of the block
				ROWS_TRACE(0)
				if (m < nchunks) {
					int* flag = &s_ready[m % 3];
					while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < m + 1) {
						if (++spins > FLOW_SPIN_LIMIT) break;
						__builtin_amdgcn_s_sleep(1);
					}
				}
				ROWS_TRACE(1)
				while (__hip_atomic_load(&s_faces, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < m + 1) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				float gj[8], gk[8];
				{
					const float4 j0 = *(const float4*)&sFj[m & 1][c][0], j1 = *(const float4*)&sFj[m & 1][c][4];
					const float4 k0 = *(const float4*)&sFk[m & 1][b][0], k1 = *(const float4*)&sFk[m & 1][b][4];
					gj[0] = j0.x; gj[1] = j0.y; gj[2] = j0.z; gj[3] = j0.w; gj[4] = j1.x; gj[5] = j1.y; gj[6] = j1.z; gj[7] = j1.w;
					gk[0] = k0.x; gk[1] = k0.y; gk[2] = k0.z; gk[3] = k0.w; gk[4] = k1.x; gk[5] = k1.y; gk[6] = k1.z; gk[7] = k1.w;
				}
				ROWS_TRACE(2)
				const int base = (8 * m) & 31;
				unsigned long long* pf = out_f + (int64_t)(8 * m) * 8;   // row h + 2 = 8m + s
#pragma unroll
				for (int s = 0; s < 8; s++) {
					const float4 cA = nA;
					const float2 cB = nB;
					const int row = ((base + s) & 31) * 64 + lane;
					const int nrow = ((base + s + 1) & 31) * 64 + lane;
					nA = sA[nrow];
					nB = sB[nrow];
					const float dj = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(oj0), 0x111, 0xf, 0xf, false));
					const float sk = __shfl_up(ok0, 8, 64);
					const float ij0 = (b == 0) ? gj[s] : dj;
					const float ik0 = (c == 0) ? gk[s] : sk;
					const float ii0 = oi0;
					const bool valid = !EDGE || ((unsigned)(xq + s) < (unsigned)X8);
					const float ai = cA.y, aj = cA.z, ak = cA.w, p = cB.x;
					const bool fl = cB.y != 0.f;
					float val = cA.x;
					if (MODE == 1) {
						const float nv = p * (val - ii0 - ij0 - ik0);
						val = fl ? nv : val;
						oi0 = valid ? (val * ai) * p : 0.f;
						oj0 = valid ? (val * aj) * p : 0.f;
						ok0 = valid ? (val * ak) * p : 0.f;
					} else {
						const float nv = p * (val - ii0 * ai * p - ij0 * aj * p - ik0 * ak * p);
						val = fl ? nv : val;
						oi0 = oj0 = ok0 = valid ? val : 0.f;
					}
					// one face store per step: lanes b == 7 publish the j face, lanes c == 7 the k face, and lane 0 (never a
					// face lane) publishes the k value of the corner lane 63, which is busy with its j value
					const float corner = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ok0), 63));
					const float fv = (lane == 0) ? corner : ((b == 7) ? oj0 : ok0);
					if (valid) sA[row].x = val;
					if (face_lane) {
						const bool fvalid = !EDGE || ((unsigned)(xq + s + fskew) < (unsigned)X8);
						if (fvalid) {
							if (LOCAL) granule_store_local(pf + s * 8, fv, gen);
							else granule_store(pf + s * 8, fv, gen);
						}
					}
				}
				// LDS-only release: the block's granule stores need not have been acknowledged before the next block starts
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
				__hip_atomic_store(&s_done, m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				ROWS_TRACE(3)
			};
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				// interior: every lane's x' is inside [0, X8) for all 8 steps of the block
				const bool interior = (m >= 2 && m <= nchunks - 1);
				if (local_faces) {
					if (interior) block(m, std::false_type{}, std::true_type{});
					else block(m, std::true_type{}, std::true_type{});
				} else {
					if (interior) block(m, std::false_type{}, std::false_type{});
					else block(m, std::true_type{}, std::false_type{});
				}
			}
			if (trace && lane == 0 && t < 4096) trace[4 * 4096 + 2 * t + 1] = wall_clock64();
#undef ROWS_TRACE
		}
		__syncthreads();
	}
	if (spins > FLOW_SPIN_LIMIT) atomicExch(&ctl->err, 1);
	if (threadIdx.x == 0) {
		const int f = atomicAdd(&ctl->finished, 1);
		if (f == (int)gridDim.x - 1) {
			for (int q = 0; q < 8; q++) xt[q] = 0;
			ctl->finished = 0;
		}
	}
}

// host-side state of the dataflow sweeps (per device): tile order for the current grid, exchange buffer, generation
struct FlowState {
	int nti = 0, ntj = 0, ntk = 0, ntiles = 0;
	int nbj = 0, nbk = 0, nblocks = 0, nchunks = 0;   // streaming form
	int* border = nullptr;       // ticket order of the forward sweep, then of the backward sweep (nblocks entries each)
	int jb = 0;                  // bundles per j-block the order was built for
	int nq = 0;                  // ticket queues (1 or 8)
	int* rows_xt = nullptr;      // [2][18]: tickets, queue bounds, queue count -- forward sweep, backward sweep
	unsigned long long *sxj = nullptr, *sxk = nullptr;
	size_t sx_cap = 0;
	unsigned sgen = 0;
	int* order = nullptr;
	int level_count[3072];
	unsigned long long* xch = nullptr;
	size_t xch_cap = 0;
	FlowCtl* ctl = nullptr;
	unsigned gen = 0;
};
static FlowState g_flow[16];

static int flow_prepare(const Dim& d, FlowState** out, hipStream_t st, bool need_xch = false) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	const int nti = (d.sx + 7) / 8, ntj = (d.sy + 7) / 8, ntk = (d.sz + 7) / 8;
	if (nti > 1023 || ntj > 1023 || ntk > 1023) return fail("grid too large for the MIC tile order table");
	if (!f.ctl) {
		MF_HIP(hipMalloc((void**)&f.ctl, sizeof(FlowCtl)));
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
	}
	if (f.nti != nti || f.ntj != ntj || f.ntk != ntk) {
		MF_HIP(hipStreamSynchronize(st));
		const int nt = nti * ntj * ntk;
		int* h = (int*)malloc(sizeof(int) * nt);
		int q = 0;
		for (int L = 0; L <= nti + ntj + ntk - 3; L++) {
			const int q0 = q;
			for (int tk = 0; tk < ntk; tk++)
				for (int tj = 0; tj < ntj; tj++) {
					const int ti = L - tj - tk;
					if (ti >= 0 && ti < nti) h[q++] = ti | (tj << 10) | (tk << 20);
				}
			f.level_count[L] = q - q0;
		}
		if (f.order) MF_HIP(hipFree(f.order));
		MF_HIP(hipMalloc((void**)&f.order, sizeof(int) * nt));
		MF_HIP(hipMemcpy(f.order, h, sizeof(int) * nt, hipMemcpyHostToDevice));
		free(h);
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
		f.gen = 0;
		f.nti = nti;
		f.ntj = ntj;
		f.ntk = ntk;
		f.ntiles = nt;
		if (f.xch) MF_HIP(hipMemset(f.xch, 0, f.xch_cap));
	}
	if (need_xch) {
		const size_t need = (size_t)f.ntiles * 3 * 64 * sizeof(unsigned long long);
		if (need > f.xch_cap) {
			MF_HIP(hipStreamSynchronize(st));
			if (f.xch) MF_HIP(hipFree(f.xch));
			MF_HIP(hipMalloc((void**)&f.xch, need));
			f.xch_cap = need;
			MF_HIP(hipMemset(f.xch, 0, f.xch_cap));
			f.gen = 0;
		}
	}
	*out = &f;
	return 0;
}
// per-XCD ticket queues rely on workgroup b running on XCD b % 8 (round-robin dispatch, verified once per process)
__global__ void k_probe_xcc(int* out) {
	if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}
static bool xcd_round_robin_ok() {
	static int state = -1;
	if (state < 0) {
		state = 0;
		int* dbuf = nullptr;
		int hbuf[64];
		if (hipMalloc((void**)&dbuf, sizeof hbuf) == hipSuccess) {
			hipLaunchKernelGGL(k_probe_xcc, dim3(64), dim3(64), 0, 0, dbuf);
			if (hipMemcpy(hbuf, dbuf, sizeof hbuf, hipMemcpyDeviceToHost) == hipSuccess) {
				state = 1;
				for (int b = 0; b < 64; b++)
					if (hbuf[b] != (b & 7)) state = 0;
			}
			(void)hipFree(dbuf);
		}
	}
	return state == 1;
}
static thread_local int g_mic_jblock_rows = 0;   // mf_set_mic_blocking
static int rows_prepare(const Dim& d, FlowState** out, hipStream_t st) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	const int nbj = (d.sy + 7) / 8, nbk = (d.sz + 7) / 8, nchunks = (d.sx + 7) / 8;
	if (nbj > 65535 || nbk > 32767) return fail("grid too large for the MIC bundle order table");
	int jb = g_mic_jblock_rows > 0 ? g_mic_jblock_rows / 8 : nbj;
	if (jb < 1 || jb > nbj) jb = nbj;
	static const int use_xcd = getenv("MF_ROWS_XCD") ? atoi(getenv("MF_ROWS_XCD")) : 0;
	const int nq = (use_xcd && nbk >= 8 && xcd_round_robin_ok()) ? 8 : 1;
	if (!f.ctl) {
		MF_HIP(hipMalloc((void**)&f.ctl, sizeof(FlowCtl)));
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
	}
	if (f.nbj != nbj || f.nbk != nbk || f.nchunks != nchunks || f.jb != jb || f.nq != nq) {
		MF_HIP(hipStreamSynchronize(st));
		const int nb = nbj * nbk;
		int* h = (int*)malloc(sizeof(int) * 2 * nb);
		int xt[2][18];
		memset(xt, 0, sizeof xt);
		// tickets in topological order of each sweep: key = position of the bundle inside its j-block along the sweep
		// direction + tkl (anti-diagonals of the block-local dependency graph); one queue per XCD = k-slab of bundles
		for (int rev = 0; rev < 2; rev++) {
			int q = 0;
			for (int x = 0; x < 8; x++) {
				xt[rev][8 + x] = q;
				if (x < nq)
					for (int L = 0; L <= nbj + nbk - 2; L++)
						for (int bk = 0; bk < nbk; bk++) {
							if ((bk * nq) / nbk != x) continue;
							for (int bjl = 0; bjl < nbj; bjl++) {
								const int tj = rev ? nbj - 1 - bjl : bjl;   // physical bundle row
								const int b0 = (tj / jb) * jb, b1 = (b0 + jb < nbj ? b0 + jb : nbj);
								const int posj = rev ? (b1 - 1 - tj) : (tj - b0);
								if (posj + bk == L) h[rev * nb + q++] = bjl | (bk << 16);
							}
						}
			}
			xt[rev][16] = q;
			xt[rev][17] = nq;
		}
		if (!f.rows_xt) MF_HIP(hipMalloc((void**)&f.rows_xt, sizeof(xt)));
		MF_HIP(hipMemcpy(f.rows_xt, xt, sizeof(xt), hipMemcpyHostToDevice));
		if (f.border) MF_HIP(hipFree(f.border));
		MF_HIP(hipMalloc((void**)&f.border, sizeof(int) * 2 * nb));
		MF_HIP(hipMemcpy(f.border, h, sizeof(int) * 2 * nb, hipMemcpyHostToDevice));
		free(h);
		// one granule per (bundle, x', face lane) and face
		const size_t need = (size_t)nb * 8 * (8 * (size_t)nchunks + 2 * ROWS_PAD) * sizeof(unsigned long long);
		if (need > f.sx_cap) {
			if (f.sxj) MF_HIP(hipFree(f.sxj));
			if (f.sxk) MF_HIP(hipFree(f.sxk));
			MF_HIP(hipMalloc((void**)&f.sxj, need));
			MF_HIP(hipMalloc((void**)&f.sxk, need));
			f.sx_cap = need;
		}
		MF_HIP(hipMemset(f.sxj, 0, f.sx_cap));
		MF_HIP(hipMemset(f.sxk, 0, f.sx_cap));
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
		f.sgen = 0;
		f.nbj = nbj;
		f.nbk = nbk;
		f.nblocks = nb;
		f.nchunks = nchunks;
		f.jb = jb;
		f.nq = nq;
	}
	*out = &f;
	return 0;
}
// MF_MIC_MODE selects how the two apply sweeps are parallelised (all three give bit-identical results):
// 2 "rows"  : one launch per sweep, a 6-wave workgroup per 8x8 bundle of x-rows streaming along x (default for 3D grids:
//              256^3 apply 0.83 ms; 2D grids fall through to "tiles")
// 1 "tiles" : one launch per sweep, ticketed 8^3 tiles + tagged sc1 granules, operands of the next tile prefetched
//              (256^3 apply 1.32 ms)
// 0 "levels": one launch per tile hyperplane (no inter-workgroup waiting at all; the conservative fallback, 2.2 ms)
static int g_mic_mode = -1;
extern "C" int mf_set_mic_mode(const char* name) {
	if (!name || !*name) g_mic_mode = -1;
	else if (!strcmp(name, "levels")) g_mic_mode = 0;
	else if (!strcmp(name, "tiles")) g_mic_mode = 1;
	else if (!strcmp(name, "rows")) g_mic_mode = 2;
	else return fail("mf_set_mic_mode: unknown mode (rows | tiles | levels)");
	return 0;
}
extern "C" int mf_set_mic_blocking(int rows_j) {
	if (rows_j < 0 || (rows_j % 8) != 0) return fail("mf_set_mic_blocking: rows must be a non-negative multiple of 8");
	g_mic_jblock_rows = rows_j;
	return 0;
}
static int mic_mode() {
	if (g_mic_mode < 0) {
		const char* e = getenv("MF_MIC_MODE");
		g_mic_mode = (e && !strcmp(e, "levels")) ? 0 : ((e && !strcmp(e, "tiles")) ? 1 : 2);
	}
	return g_mic_mode;
}

template <int MODE>
static int launch_mic(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                      const float* Aj, const float* Ak, const CgScalars* sc, hipStream_t st) {
	const int nti = (d.sx + 7) / 8, ntj = (d.sy + 7) / 8, ntk = (d.sz + 7) / 8;
	const int levels = nti + ntj + ntk - 2;
	const bool vec = (d.sx % 4 == 0) && al16(flags) && al16(dst) && al16(var1) && al16(Ai) && al16(Aj) && al16(Ak) && (MODE == 0 || al16(Ap));
	if constexpr (MODE != 0) {
		if (mic_mode() == 2 && d.is3d) {
			FlowState* f;
			MF_TRY(rows_prepare(d, &f, st));
			f->sgen++;
			if (f->sgen == 0) {
				MF_HIP(hipMemsetAsync(f->sxj, 0, f->sx_cap, st));
				MF_HIP(hipMemsetAsync(f->sxk, 0, f->sx_cap, st));
				f->sgen = 1;
			}
			static int rwgs = -1;
			if (rwgs < 0) {
				const char* e = getenv("MF_ROWS_WGS");
				int dev = 0, ncu = 256;
				(void)hipGetDevice(&dev);
				(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
				// one bundle per CU measured best (256^3: 834 us per apply with 256 workgroups, 1040 us with 512): the
				// sweep is bound by the chain of face hand-offs, and a second bundle per CU slows both
				rwgs = e ? atoi(e) : ncu;
				if (rwgs < 1) rwgs = 1;
			}
			const int grid = f->nblocks < rwgs ? f->nblocks : rwgs;
			// MF_ROWS_TRACE=<ticket>: per-block wall-clock stamps (100 MHz) of that bundle's compute wave, printed per launch
			static long long* trace = nullptr;
			static int trace_ticket = -2;
			static const int trace_ticket2 = getenv("MF_ROWS_TRACE2") ? atoi(getenv("MF_ROWS_TRACE2")) : -1;
			if (trace_ticket == -2) {
				const char* e = getenv("MF_ROWS_TRACE");
				trace_ticket = e ? atoi(e) : -1;
				if (trace_ticket >= 0) {
					MF_HIP(hipMalloc((void**)&trace, sizeof(long long) * 12 * 4096));
					MF_HIP(hipMemset(trace, 0, sizeof(long long) * 12 * 4096));
				}
			}
			if (vec)
				hipLaunchKernelGGL((k_mic_rows<MODE, true>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, f->nchunks, f->border + (MODE == 2 ? f->nblocks : 0), f->ctl, f->rows_xt + (MODE == 2 ? 18 : 0), f->sxj, f->sxk, f->sgen, flags, dst, var1, Ap, Ai, Aj, Ak, sc, trace, trace_ticket, trace_ticket2);
			else
				hipLaunchKernelGGL((k_mic_rows<MODE, false>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, f->nchunks, f->border + (MODE == 2 ? f->nblocks : 0), f->ctl, f->rows_xt + (MODE == 2 ? 18 : 0), f->sxj, f->sxk, f->sgen, flags, dst, var1, Ap, Ai, Aj, Ak, sc, trace, trace_ticket, trace_ticket2);
			MF_LAUNCH_CHECK();
			if (trace) {
				static int printed = 0;
				MF_HIP(hipStreamSynchronize(st));
				if (printed++ < 4) {
					const int nb = f->nchunks + 2;
					long long* h = (long long*)malloc(sizeof(long long) * 4 * nb);
					MF_HIP(hipMemcpy(h, trace, sizeof(long long) * 4 * nb, hipMemcpyDeviceToHost));
					fprintf(stderr, "[rows trace] mode %d ticket %d: block: gap ready poll steps (us)\n", MODE, trace_ticket);
					for (int m = 0; m < nb; m++)
						fprintf(stderr, "  m=%2d  %6.2f %6.2f %6.2f %6.2f   t=%.2f\n", m, m ? (h[m * 4] - h[m * 4 - 1]) * 0.01 : 0.0, (h[m * 4 + 1] - h[m * 4]) * 0.01,
						        (h[m * 4 + 2] - h[m * 4 + 1]) * 0.01, (h[m * 4 + 3] - h[m * 4 + 2]) * 0.01, (h[m * 4 + 3] - h[0]) * 0.01);
					if (trace_ticket2 >= 0) {
						// producer (ticket) vs consumer (ticket2): consumer block m needs the producer's step 8m+12 (its block m+1)
						long long* h2 = (long long*)malloc(sizeof(long long) * 4 * nb);
						MF_HIP(hipMemcpy(h2, trace + 8 * 4096, sizeof(long long) * 4 * nb, hipMemcpyDeviceToHost));
						fprintf(stderr, "[rows trace] hand-off %d -> %d: m  producer_end(m+1)  consumer_faces_ready(m)  delta   consumer_block_start(m)\n", trace_ticket, trace_ticket2);
						for (int m = 0; m + 1 < nb; m++)
							fprintf(stderr, "  m=%2d  %8.2f  %8.2f  %6.2f   %8.2f\n", m, (h[(m + 1) * 4 + 3] - h[0]) * 0.01, (h2[m * 4 + 2] - h[0]) * 0.01,
							        (h2[m * 4 + 2] - h[(m + 1) * 4 + 3]) * 0.01, (h2[m * 4] - h[0]) * 0.01);
						free(h2);
					}
					free(h);
					const int ns = f->nblocks < 4096 ? f->nblocks : 4096;
					long long* g = (long long*)malloc(sizeof(long long) * 2 * ns);
					MF_HIP(hipMemcpy(g, trace + 4 * 4096, sizeof(long long) * 2 * ns, hipMemcpyDeviceToHost));
					long long t0 = g[0];
					for (int i = 0; i < ns; i++) if (g[2 * i] < t0) t0 = g[2 * i];
					fprintf(stderr, "[rows trace] bundles: ticket start end (us since first start)\n");
					for (int i = 0, L = 0; i < ns; L++, i += (L < f->nbj ? L : 1) + 0) {
						fprintf(stderr, "  t=%4d  %8.2f %8.2f\n", i, (g[2 * i] - t0) * 0.01, (g[2 * i + 1] - t0) * 0.01);
						if (L > 200) break;
					}
					free(g);
					long long* hw = (long long*)malloc(sizeof(long long) * ns);
					MF_HIP(hipMemcpy(hw, trace + 6 * 4096, sizeof(long long) * ns, hipMemcpyDeviceToHost));
					fprintf(stderr, "[rows trace] compute wave placement: ticket block simd cu sh se\n");
					for (int i = 0; i < ns && i < 1024; i += 37) {
						const unsigned v = (unsigned)hw[i];
						fprintf(stderr, "  t=%4d blk=%4d simd=%u cu=%u sh=%u se=%u wave=%u\n", i, (int)(hw[i] >> 32), (v >> 4) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7, v & 15);
					}
					free(hw);
				}
			}
			return 0;
		}
		if (mic_mode() >= 1) {
			FlowState* f;
			MF_TRY(flow_prepare(d, &f, st, true));
			f->gen++;
			if (f->gen == 0) {  // 32-bit wrap: stale tags could alias -> clear once
				MF_HIP(hipMemsetAsync(f->xch, 0, f->xch_cap, st));
				f->gen = 1;
			}
			static int wgs = -1;
			if (wgs < 0) {
				const char* e = getenv("MF_FLOW_WGS");
				wgs = e ? atoi(e) : 512;   // two single-wave workgroups per CU measured best (256: too few, 1024: contention)
				if (wgs < 1) wgs = 1;
			}
			const int grid = f->ntiles < wgs ? f->ntiles : wgs;
			if (vec)
				hipLaunchKernelGGL((k_mic_flow<MODE, true>), dim3(grid), dim3(64), 0, st, d, nti, ntj, ntk, f->ntiles, f->order, f->ctl, f->xch, f->gen, flags, dst, var1, Ap, Ai, Aj, Ak, sc);
			else
				hipLaunchKernelGGL((k_mic_flow<MODE, false>), dim3(grid), dim3(64), 0, st, d, nti, ntj, ntk, f->ntiles, f->order, f->ctl, f->xch, f->gen, flags, dst, var1, Ap, Ai, Aj, Ak, sc);
			MF_LAUNCH_CHECK();
			return 0;
		}
	}
	for (int L = 0; L < levels; L++) {
		if (vec)
			hipLaunchKernelGGL((k_mic_tiles<MODE, true>), dim3(ntj, ntk), dim3(64), 0, st, d, L, nti, ntj, ntk, flags, dst, var1, Ap, Ai, Aj, Ak, sc);
		else
			hipLaunchKernelGGL((k_mic_tiles<MODE, false>), dim3(ntj, ntk), dim3(64), 0, st, d, L, nti, ntj, ntk, flags, dst, var1, Ap, Ai, Aj, Ak, sc);
	}
	MF_LAUNCH_CHECK();
	return 0;
}

// =========================================================================================================
// PCG scalar kernels (one block) -- scalars never leave the device inside an iteration
// =========================================================================================================
__global__ void __launch_bounds__(BLOCK) k_cg_begin(CgScalars* sc, int nb, const double* __restrict__ partials, float accuracy, int useL2) {
	double acc = strided_sum(partials, nb);
	acc = block_sum(acc);
	if (threadIdx.x == 0) {
		sc->sigma = (float)acc;  // mSigma = GridDotProduct(mTmp, mResidual), conjugategrad.cpp:234
		sc->alpha = sc->nalpha = sc->beta = 0.f;
		sc->resNorm = 1e20f;
		sc->accuracy = accuracy;
		sc->iterations = 0;
		sc->done = 0;
		sc->diverged = 0;
		sc->useL2 = useL2;
	}
}
// alpha = sigma / dp, conjugategrad.cpp:250-252
__global__ void __launch_bounds__(BLOCK) k_cg_alpha(CgScalars* sc, int nb, const double* __restrict__ partials) {
	if (sc->done) return;
	double acc = strided_sum(partials, nb);
	acc = block_sum(acc);
	if (threadIdx.x == 0) {
		const float dp = (float)acc;
		float alpha = 0.f;
		if (fabs((double)dp) > 0.) alpha = sc->sigma / dp;
		sc->dp = dp;
		sc->alpha = alpha;
		sc->nalpha = -alpha;
		sc->iterations++;
	}
}
// dst += alpha*search ; residual += (-alpha)*tmp ; [PC_NONE: tmp = residual] ; partial min/max (or sum of
// squares) of the new residual.  conjugategrad.cpp:254-255, 265, 268-272
template <bool COPY_TMP>
__global__ void __launch_bounds__(BLOCK)
k_cg_axpy2(int64_t n, const CgScalars* __restrict__ sc, float* __restrict__ dst, const float* __restrict__ search,
           float* __restrict__ residual, float* __restrict__ tmp, float* __restrict__ fpart, double* __restrict__ dpart) {
	if (sc->done) return;
	const float alpha = sc->alpha, nalpha = sc->nalpha;
	const bool l2 = sc->useL2 != 0;
	float lo = FLT_MAX, hi = -FLT_MAX;
	double ss = 0.0;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		float4 x = ((float4*)dst)[q];
		const float4 s = ((const float4*)search)[q];
		float4 r = ((float4*)residual)[q];
		const float4 t = ((const float4*)tmp)[q];
		x.x = x.x + alpha * s.x; x.y = x.y + alpha * s.y; x.z = x.z + alpha * s.z; x.w = x.w + alpha * s.w;
		r.x = r.x + nalpha * t.x; r.y = r.y + nalpha * t.y; r.z = r.z + nalpha * t.z; r.w = r.w + nalpha * t.w;
		((float4*)dst)[q] = x;
		((float4*)residual)[q] = r;
		if (COPY_TMP) ((float4*)tmp)[q] = r;
		if (l2) {
			ss += (double)r.x * (double)r.x;
			ss += (double)r.y * (double)r.y;
			ss += (double)r.z * (double)r.z;
			ss += (double)r.w * (double)r.w;
		} else {
			lo = fminf(fminf(lo, r.x), fminf(r.y, fminf(r.z, r.w)));
			hi = fmaxf(fmaxf(hi, r.x), fmaxf(r.y, fmaxf(r.z, r.w)));
		}
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		dst[i] = dst[i] + alpha * search[i];
		const float r = residual[i] + nalpha * tmp[i];
		residual[i] = r;
		if (COPY_TMP) tmp[i] = r;
		if (l2) ss += (double)r * (double)r;
		lo = fminf(lo, r);
		hi = fmaxf(hi, r);
	}
	if (l2) {
		ss = block_sum(ss);
		if (threadIdx.x == 0) dpart[blockIdx.x] = ss;
	} else {
		block_minmax(lo, hi);
		if (threadIdx.x == 0) {
			fpart[2 * blockIdx.x] = lo;
			fpart[2 * blockIdx.x + 1] = hi;
		}
	}
}
// sigmaNew partials: GridDotProduct(tmp, residual), conjugategrad.cpp:279
__global__ void __launch_bounds__(BLOCK)
k_cg_dot(int64_t n, const CgScalars* __restrict__ sc, const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partials) {
	if (sc->done) return;
	double acc = 0.0;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		const float4 x = ((const float4*)a)[q], y = ((const float4*)b)[q];
		const float p0 = x.x * y.x, p1 = x.y * y.y, p2 = x.z * y.z, p3 = x.w * y.w;
		acc += (double)p0;
		acc += (double)p1;
		acc += (double)p2;
		acc += (double)p3;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		const float p = a[i] * b[i];
		acc += (double)p;
	}
	acc = block_sum(acc);
	if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// residual norm, convergence test, beta.  conjugategrad.cpp:268-295
__global__ void __launch_bounds__(BLOCK)
k_cg_beta(CgScalars* sc, int nbr, const float* __restrict__ fpart, const double* __restrict__ dpart_res, int nbd,
          const double* __restrict__ dpart_dot) {
	if (sc->done) return;
	float lo = FLT_MAX, hi = -FLT_MAX;
	double ss = 0.0, dd = 0.0;
	const bool l2 = sc->useL2 != 0;
	for (int i = threadIdx.x; i < nbr; i += blockDim.x) {
		if (l2)
			ss += dpart_res[i];
		else {
			lo = fminf(lo, fpart[2 * i]);
			hi = fmaxf(hi, fpart[2 * i + 1]);
		}
	}
	dd = strided_sum(dpart_dot, nbd);
	block_minmax(lo, hi);
	__syncthreads();
	ss = block_sum(ss);
	__syncthreads();
	dd = block_sum(dd);
	if (threadIdx.x == 0) {
		float resNorm;
		if (l2)
			resNorm = (float)ss;
		else {
			const float alo = fabsf(lo), ahi = fabsf(hi);
			resNorm = alo > ahi ? alo : ahi;
		}
		sc->resNorm = resNorm;
		if (resNorm < sc->accuracy) {
			sc->sigma = resNorm;
			sc->done = 1;
		} else {
			const float sigmaNew = (float)dd;
			sc->beta = sigmaNew / sc->sigma;
			sc->sigmaNew = sigmaNew;
			sc->sigma = sigmaNew;
			if (!((double)resNorm < 1e35)) {
				sc->diverged = 1;  // the host finishes this iteration's search update, then reports
			}
		}
	}
}
// search = tmp + beta*search, conjugategrad.cpp:193-196, 283
__global__ void __launch_bounds__(BLOCK)
k_cg_update_search(int64_t n, CgScalars* __restrict__ sc, float* __restrict__ search, const float* __restrict__ tmp) {
	if (sc->done) return;
	const float beta = sc->beta;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		float4 s = ((float4*)search)[q];
		const float4 t = ((const float4*)tmp)[q];
		s.x = t.x + beta * s.x; s.y = t.y + beta * s.y; s.z = t.z + beta * s.z; s.w = t.w + beta * s.w;
		((float4*)search)[q] = s;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		search[i] = tmp[i] + beta * search[i];
	}
}
__global__ void k_cg_latch_diverged(CgScalars* sc) {
	if (sc->diverged) sc->done = 1;
}

extern "C" {

int mf_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                    const float* Ai, const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	return launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, (hipStream_t)stream, nullptr);
}

int mf_time_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                         const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	hipEvent_t e0, e1;
	MF_HIP(hipEventCreate(&e0));
	MF_HIP(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++) MF_TRY(launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, st, nullptr));
	MF_HIP(hipEventRecord(e0, st));
	for (int i = 0; i < reps; i++) MF_TRY(launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, st, nullptr));
	MF_HIP(hipEventRecord(e1, st));
	MF_HIP(hipEventSynchronize(e1));
	float ms = 0.f;
	MF_HIP(hipEventElapsedTime(&ms, e0, e1));
	*avg_us = (double)ms * 1000.0 / (reps > 0 ? reps : 1);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return 0;
}

int mf_make_laplace_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak,
                           const float* fractions, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_make_laplace, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, A0, Ai, Aj, Ak, fractions);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_make_rhs(int sx, int sy, int sz, const int32_t* flags, float* rhs, const float* vel, const float* perCellCorr,
                const float* fractions, const float* obvel, const float* phi, const float* curv, float surfTens,
                float gfClamp, int32_t* cnt_host, double* sum_host, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	hipStream_t st = (hipStream_t)stream;
	const int nb = blocks_for(d.n, BLOCK, 2048);
	hipLaunchKernelGGL(k_make_rhs, dim3(nb), dim3(BLOCK), 0, st, d, flags, rhs, vel, perCellCorr, fractions, obvel, phi, curv, surfTens, gfClamp, ws->partials);
	hipLaunchKernelGGL(k_sum2_finish, dim3(1), dim3(BLOCK), 0, st, nb, ws->partials, ws->partials + MAX_BLOCKS, (double*)ws->scalars);
	MF_LAUNCH_CHECK();
	if (cnt_host || sum_host) {
		MF_HIP(hipMemcpyAsync(ws->host, ws->scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
		MF_HIP(hipStreamSynchronize(st));
		const double* h = (const double*)ws->host;
		if (sum_host) *sum_host = h[0];
		if (cnt_host) *cnt_host = (int32_t)h[1];
	}
	return 0;
}

int mf_apply_ghost_fluid_diagonal(int sx, int sy, int sz, float* A0, const int32_t* flags, const float* phi, float gfClamp, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_ghost_fluid_diag, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, A0, flags, phi, gfClamp);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_correct_velocity(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* pressure, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_correct_velocity, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, pressure);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_correct_velocity_ghost_fluid(int sx, int sy, int sz, float* vel, const int32_t* flags, const float* pressure,
                                    const float* phi, float gfClamp, const float* curv, float surfTens, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_correct_velocity_gf, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, flags, pressure, phi, gfClamp, curv, surfTens);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_replace_clamped_ghost_fluid_vels(int sx, int sy, int sz, float* vel, const int32_t* flags, const float* pressure,
                                        const float* phi, float gfClamp, void* stream) {
	(void)pressure;
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_replace_clamped_gf, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, flags, phi, gfClamp);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_fix_pressure(int sx, int sy, int sz, int64_t fixPidx, float value, float* rhs, float* A0, float* Ai, float* Aj, float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (fixPidx - d.Y < 0 || fixPidx + d.Y >= d.n || fixPidx - d.Z < 0 || fixPidx + d.Z >= d.n) return fail("fixPressure: cell %lld on the domain border", (long long)fixPidx);
	hipLaunchKernelGGL(k_fix_pressure, dim3(1), dim3(1), 0, (hipStream_t)stream, d, fixPidx, value, rhs, A0, Ai, Aj, Ak);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_mic_init(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0, const float* Ai,
                const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	MF_HIP(hipMemsetAsync(Aprecond, 0, sizeof(float) * d.n, (hipStream_t)stream));
	return launch_mic<0>(d, flags, Aprecond, A0, nullptr, Ai, Aj, Ak, nullptr, (hipStream_t)stream);
}
int mf_mic_apply(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1, const float* Aprecond,
                 const float* Ai, const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	MF_TRY(launch_mic<1>(d, flags, dst, var1, Aprecond, Ai, Aj, Ak, nullptr, (hipStream_t)stream));
	return launch_mic<2>(d, flags, dst, var1, Aprecond, Ai, Aj, Ak, nullptr, (hipStream_t)stream);
}

// a dataflow sweep that gives up waiting for a face (FLOW_SPIN_LIMIT) latches an error flag on the device; mf_cg_solve
// looks at it itself, callers that drive mf_mic_apply directly (the z-slab solver) ask here once per solve
int mf_mic_check(void* stream) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	if (!g_flow[dev].ctl) return 0;
	MF_HIP(hipStreamSynchronize((hipStream_t)stream));
	FlowCtl fc;
	MF_HIP(hipMemcpy(&fc, g_flow[dev].ctl, sizeof fc, hipMemcpyDeviceToHost));
	if (fc.err) {
		MF_HIP(hipMemset(g_flow[dev].ctl, 0, sizeof(FlowCtl)));
		return fail("MIC dataflow sweep: a workgroup timed out waiting for its predecessor faces");
	}
	return 0;
}

int mf_cg_solve(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* rhs, float* residual, float* search,
                float* tmp, const float* A0, const float* Ai, const float* Aj, const float* Ak, float* Aprecond, int pc,
                float accuracy, int maxIter, int useL2Norm, float* out_host, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	hipStream_t st = (hipStream_t)stream;
	if (pc != MF_PC_NONE && pc != MF_PC_MICP) return fail("GridCg<APPLYMAT>::setICPreconditioner: Invalid method specified.");
	if (pc == MF_PC_MICP && !d.is3d) pc = MF_PC_NONE;  // conjugategrad.cpp:315-321
	if (!(al16(dst) && al16(rhs) && al16(residual) && al16(search) && al16(tmp))) return fail("mf_cg_solve: work grids must be 16-byte aligned");
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	CgScalars* sc = (CgScalars*)ws->scalars;
	double* p_dot = ws->partials;                   // apply-matrix / sigma dot partials
	double* p_res = ws->partials + MAX_BLOCKS;      // sum-of-squares partials of the residual
	double* p_sig = ws->partials + 2 * MAX_BLOCKS;  // dot(tmp, residual) partials
	float* p_mm = ws->fpartials;
	const int nbs = blocks_for(n >> 2, BLOCK, 2048);

	// ---- doInit, conjugategrad.cpp:210-235 ----
	MF_HIP(hipMemsetAsync(dst, 0, sizeof(float) * n, st));
	MF_HIP(hipMemcpyAsync(residual, rhs, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
	if (pc == MF_PC_MICP) {
		MF_TRY(mf_mic_init(sx, sy, sz, flags, Aprecond, A0, Ai, Aj, Ak, stream));
		MF_TRY(mf_mic_apply(sx, sy, sz, flags, tmp, residual, Aprecond, Ai, Aj, Ak, stream));
	} else {
		MF_HIP(hipMemcpyAsync(tmp, residual, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
	}
	MF_HIP(hipMemcpyAsync(search, tmp, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
	MF_HIP(hipMemsetAsync(sc, 0, sizeof(CgScalars), st));
	hipLaunchKernelGGL(k_cg_dot, dim3(nbs), dim3(BLOCK), 0, st, n, sc, tmp, residual, p_sig);
	hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(BLOCK), 0, st, sc, nbs, p_sig, accuracy, useL2Norm);
	MF_LAUNCH_CHECK();

	// ---- iterate, conjugategrad.cpp:238-299; the host only polls `done`, one batch behind the batch it has just queued
	// (every kernel of an iteration returns at once when `done` is already set, so running ahead costs a few empty
	// launches after convergence and keeps the GPU from idling between iterations) ----
	const int batch = (pc == MF_PC_MICP && mic_mode() == 0) ? 1 : 4;
	CgScalars h;
	memset(&h, 0, sizeof h);
	h.resNorm = 1e20f;
	static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
	if (!ev[0]) {
		MF_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
		MF_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
	}
	CgScalars* hslot = (CgScalars*)ws->host;   // two pinned slots
	int issued = 0, slot = 0, pending = -1;
	while (issued < maxIter) {
		const int todo = (maxIter - issued < batch) ? (maxIter - issued) : batch;
		for (int it = 0; it < todo; it++) {
			int nba = 0;
			MF_TRY(launch_apply_matrix<true>(d, flags, tmp, search, A0, Ai, Aj, Ak, p_dot, sc, st, &nba));
			hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(BLOCK), 0, st, sc, nba, p_dot);
			if (pc == MF_PC_MICP) {
				hipLaunchKernelGGL((k_cg_axpy2<false>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, dst, search, residual, tmp, p_mm, p_res);
				MF_TRY(launch_mic<1>(d, flags, tmp, residual, Aprecond, Ai, Aj, Ak, sc, st));
				MF_TRY(launch_mic<2>(d, flags, tmp, residual, Aprecond, Ai, Aj, Ak, sc, st));
			} else {
				hipLaunchKernelGGL((k_cg_axpy2<true>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, dst, search, residual, tmp, p_mm, p_res);
			}
			hipLaunchKernelGGL(k_cg_dot, dim3(nbs), dim3(BLOCK), 0, st, n, sc, tmp, residual, p_sig);
			hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(BLOCK), 0, st, sc, nbs, p_mm, p_res, nbs, p_sig);
			hipLaunchKernelGGL(k_cg_update_search, dim3(nbs), dim3(BLOCK), 0, st, n, sc, search, tmp);
			hipLaunchKernelGGL(k_cg_latch_diverged, dim3(1), dim3(1), 0, st, sc);
		}
		MF_LAUNCH_CHECK();
		issued += todo;
		MF_HIP(hipMemcpyAsync(&hslot[slot], sc, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		MF_HIP(hipEventRecord(ev[slot], st));
		if (pending >= 0) {
			MF_HIP(hipEventSynchronize(ev[pending]));
			memcpy(&h, &hslot[pending], sizeof h);
			if (h.done) break;
		}
		pending = slot;
		slot ^= 1;
	}
	// the final state, after everything that was queued
	MF_HIP(hipMemcpyAsync(&hslot[0], sc, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	memcpy(&h, &hslot[0], sizeof h);
	if (maxIter <= 0) {
		MF_HIP(hipMemcpyAsync(ws->host, sc, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		MF_HIP(hipStreamSynchronize(st));
		memcpy(&h, ws->host, sizeof h);
	}
	out_host[0] = (float)h.iterations;
	out_host[1] = h.resNorm;
	out_host[2] = h.sigma;
	if (pc == MF_PC_MICP && mic_mode() >= 1) {
		int dev = 0;
		MF_HIP(hipGetDevice(&dev));
		if (g_flow[dev].ctl) {
			FlowCtl fc;
			MF_HIP(hipMemcpy(&fc, g_flow[dev].ctl, sizeof fc, hipMemcpyDeviceToHost));
			if (fc.err) {
				MF_HIP(hipMemset(g_flow[dev].ctl, 0, sizeof(FlowCtl)));
				return fail("MIC dataflow sweep: a tile timed out waiting for its predecessor faces");
			}
		}
	}
	if (h.diverged) return fail("GridCg::iterate: The CG solver diverged, residual norm > 1e30, stopping.");
	return 0;
}

}  // extern "C"
