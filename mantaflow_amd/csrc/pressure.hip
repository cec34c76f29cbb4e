// pressure.hip -- pressure projection on gfx950: 7-point ApplyMatrix stencil, Laplace matrix / rhs assembly,
// velocity correction (+ ghost fluid) and the device-resident PCG loop (the MIC(0) preconditioner lives in mic.hip).
// Reference: source/conjugategrad.{h,cpp}, source/plugin/pressure.cpp (cited per kernel).
#include "common.h"
#include "pressure.h"
#include <float.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

using namespace mf;

static thread_local int g_last_shortcut[3] = {0, 0, 0};      // what mf_cg_last_shortcut reports

// =========================================================================================================
// ApplyMatrix, conjugategrad.h:118-151
//   non-fluid: dst = src ; fluid: left-to-right fp32 sum of the 7 products.  28 B/cell of compulsory traffic
//   (flags, src, A0, Ai, Aj, Ak read once + dst written); the +-X/+-Y/+-Z neighbours of src and the -X/-Y/-Z
//   entries of Ai/Aj/Ak are re-reads served by L1/L2 (each XCD owns a contiguous z-range of the grid).
//   Optional fusion: per-block fp64 partial sums of dst*src (GridDotProduct(tmp, search)).
// =========================================================================================================

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
// PACKED: flags + Ai + Aj + Ak come as one byte per cell (k_mic_pack: fluid bit, "coefficient is -1" bits; only offered by the
// host when every coefficient is exactly +0 or -1): 5 four-byte loads per thread replace 11 sixteen-byte loads, the values
// are rebuilt exactly and everything after the load section is shared
__device__ __forceinline__ int4 unpack_flags(unsigned w) {
	return make_int4((w & 0x1u) ? MF_FLUID : 0, (w & 0x100u) ? MF_FLUID : 0, (w & 0x10000u) ? MF_FLUID : 0, (w & 0x1000000u) ? MF_FLUID : 0);
}
template <unsigned BIT>
__device__ __forceinline__ float4 unpack_coef(unsigned w) {
	return make_float4((w & BIT) ? -1.f : 0.f, (w & (BIT << 8)) ? -1.f : 0.f, (w & (BIT << 16)) ? -1.f : 0.f, (w & (BIT << 24)) ? -1.f : 0.f);
}
// AM_NT: non-temporal accesses (common.h: ld_nt4 / st_nt4) for what ApplyMatrix touches once -- 1: A0 / Ai / flags loads, 2: the dst
// store as well -- 91 -> 69 us at 256^3 (3: Aj / Ak too: 78 us, they are re-read as the j-1 / k-1 coefficients)
#ifndef AM_NT
#define AM_NT 2
#endif
// Inside the PCG every vector stream EXCEPT the residual is non-temporal -- search and tmp in ApplyMatrix (own rows, result), tmp in the
// residual update, search / tmp / pressure in the search update; the same in the z-slab kernels -- so that what the MIC sweeps read
// (residual, Aprecond, packed bytes, tmp between the two sweeps) stays in L2 / the memory-side cache across the iteration: 72.3 -> 70.8 ms
// per 256^3 step, --slab 74.6 -> 72.3 ms.  Only the complete set pays (any one of these alone: no difference or slower), and only where
// the vectors do not fit the caches anyway: a run-time flag of the kernels (`nt`), set for systems of more than PCG_NT_CELLS cells (10 Mi:
// the 129-plane window of one rank of two, 8.45 M cells, is 1.2 % faster cached) that are swept whole (a liquid scene whose kernels skip
// most bundles touches a fraction: 23.5 ms per dam-break step cached, 23.9 non-temporal).
#ifndef PCG_NT_MI
#define PCG_NT_MI 10
#endif
constexpr int64_t PCG_NT_CELLS = (int64_t)PCG_NT_MI << 20;
// AMP_NT: 1 = the thread's own src rows non-temporal as well (68.2 -> 66.2 us; the z neighbours too: 76.8 us)
#ifndef AMP_NT
#define AMP_NT 1
#endif
template <bool DOT, bool IS3D, int R, bool PACKED>
__global__ void __launch_bounds__(BLOCK)
k_apply_matrix_v5(Dim d, const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ src,
                  const float* __restrict__ A0, const float* __restrict__ Ai, const float* __restrict__ Aj,
                  const float* __restrict__ Ak, double* __restrict__ partials, const CgScalars* __restrict__ sc, int jgroups, int tpb,
                  const unsigned char* __restrict__ pack, int dk0, int dk1, int a0p, const int* __restrict__ bempty = nullptr, int nbj = 0,
                  const int* __restrict__ outside_bad = nullptr, const int* __restrict__ xr = nullptr) {   // DOT covers the planes [dk0, dk1) (a z-slab's own)
	// bempty (mf_cg_solve, liquid scenes): 8 x 8 bundles of rows without a fluid cell in which src is known to be zero (k_cg_outside_zero found
	// rhs and the work grids zero there, and every kernel of the iteration keeps it so): dst = src = 0 is there already, nothing to do
	// a0p bit 0 (PACKED only): bits 4-7 of the packed bytes hold the diagonal (k_mic_pack) -- A0 is not read at all, 9 instead of 13 B per cell;
	// bit 1 (DOT, the PCG): src rows and dst non-temporal (PCG_NT_CELLS)
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
	// (R divides 8: the rows of a thread lie in one bundle; the lanes this thread exchanges +-X neighbours with work on the same rows)
	// (xr: the x-range of the system's non-zero packed bytes, k_pack_xrange -- the quads outside it are as empty as the empty bundles;
	// the first / last quad inside fetch their outer x-neighbour themselves, the lane next to them has left)
	int q_lo = 0, q_hi = qx;
	if (IS3D && bempty && outside_bad[0] == 0) {
		if (xr) {
			q_lo = (xr[0] & ~7) >> 2;
			q_hi = ((xr[1] + 7) & ~7) >> 2;
		}
		if (bempty[(k >> 3) * nbj + (j0 >> 3)] || qi < q_lo || qi >= q_hi) continue;
	}
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
		if (PACKED) ajv[0] = unpack_coef<4u>(*(const unsigned*)(pack + row0 + jm * Y));
		else ajv[0] = *(const float4*)(Aj + row0 + jm * Y);
	}
#pragma unroll
	for (int r = 0; r < R; r++) {
		const int jr = (j0 + r < d.sy) ? r : (d.sy - 1 - j0);
		const int64_t idx = row0 + jr * Y;
		sv[r + 1] = ((AMP_NT & 1) && (!DOT || (a0p & 2))) ? ld_nt4(src + idx) : *(const float4*)(src + idx);      // (inside the PCG -- DOT -- only as part of the complete set)
		unsigned pw = 0;
		if (PACKED) pw = *(const unsigned*)(pack + idx);
		if (PACKED && (a0p & 1)) a0[r] = make_float4((float)((pw >> 4) & 15u), (float)((pw >> 12) & 15u), (float)((pw >> 20) & 15u), (float)((pw >> 28) & 15u));
		else a0[r] = (AM_NT >= 1) ? ld_nt4(A0 + idx) : *(const float4*)(A0 + idx);
		if (PACKED) {
			f[r] = unpack_flags(pw);
			ajv[r + 1] = unpack_coef<4u>(pw);
			ai[r] = unpack_coef<2u>(pw);
		} else {
			f[r] = (AM_NT >= 1) ? ld_nt4i(flags + idx) : *(const int4*)(flags + idx);
			ajv[r + 1] = (AM_NT >= 3) ? ld_nt4(Aj + idx) : *(const float4*)(Aj + idx);
			ai[r] = (AM_NT >= 1) ? ld_nt4(Ai + idx) : *(const float4*)(Ai + idx);
		}
		if (IS3D) {
			const int64_t im = (k > 0) ? idx - Z : idx, ip = (k < d.sz - 1) ? idx + Z : idx;
			if (PACKED) {
				ak[r] = unpack_coef<8u>(pw);
				akm[r] = unpack_coef<8u>(*(const unsigned*)(pack + im));
			} else {
				ak[r] = (AM_NT >= 3) ? ld_nt4(Ak + idx) : *(const float4*)(Ak + idx);
				akm[r] = *(const float4*)(Ak + im);
			}
			szm[r] = (AMP_NT & 2) ? ld_nt4(src + im) : *(const float4*)(src + im);
			szp[r] = (AMP_NT & 2) ? ld_nt4(src + ip) : *(const float4*)(src + ip);
		}
	}
	{
		const int jr = (j0 + R < d.sy) ? R : (d.sy - 1 - j0);
		sv[R + 1] = *(const float4*)(src + row0 + jr * Y);
	}
	// ---- +-X neighbours: adjacent lanes hold the adjacent quads of the same row, except at row / wave edges ----
	const bool edge_l = (lane == 0) || (qi == 0) || (qi == q_lo), edge_r = (lane == 63) || (qi == qx - 1) || (qi == q_hi - 1);
#pragma unroll
	for (int r = 0; r < R; r++) {
		const int jr = (j0 + r < d.sy) ? r : (d.sy - 1 - j0);
		const int64_t idx = row0 + jr * Y;
		sl[r] = wave_shr1(sv[r + 1].w);
		al[r] = wave_shr1(ai[r].w);
		sr[r] = wave_shl1(sv[r + 1].x);
		if (edge_l) {
			sl[r] = (idx > 0) ? src[idx - 1] : 0.f;
			if (PACKED) al[r] = (idx > 0 && (pack[idx - 1] & 2u)) ? -1.f : 0.f;
			else al[r] = (idx > 0) ? Ai[idx - 1] : 0.f;
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
			if (AM_NT >= 2 && (!DOT || (a0p & 2))) st_nt4(dst + row0 + r * Y, res);
			else *(float4*)(dst + row0 + r * Y) = res;
			if (DOT && k >= dk0 && k < dk1) {
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
                               const CgScalars* sc, hipStream_t st, int* nblocks, const unsigned char* pack = nullptr,
                               int dk0 = 0, int dk1 = 0x7fffffff, bool* ranged = nullptr, bool a0p = false, const int* bempty = nullptr,
                               int nbj = 0, const int* outside_bad = nullptr, const int* xr = nullptr, int nt = -1) {
	if (ranged) *ranged = false;
	const int a0pn = (a0p ? 1 : 0) | ((DOT && (nt < 0 ? d.n > PCG_NT_CELLS : nt != 0)) ? 2 : 0);
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
	if (d.is3d && pack)                                                                                                                           \
		hipLaunchKernelGGL((k_apply_matrix_v5<DOT, true, RR, true>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, jgroups, tpb, pack, dk0, dk1, a0pn, bempty, nbj, outside_bad, xr); \
	else if (d.is3d)                                                                                                                              \
		hipLaunchKernelGGL((k_apply_matrix_v5<DOT, true, RR, false>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, jgroups, tpb, pack, dk0, dk1, a0pn, bempty, nbj, outside_bad, xr); \
	else                                                                                                                                          \
		hipLaunchKernelGGL((k_apply_matrix_v5<DOT, false, RR, false>), dim3(nb), dim3(BLOCK), 0, st, d, flags, dst, src, A0, Ai, Aj, Ak, partials, sc, jgroups, tpb, pack, dk0, dk1, a0pn, bempty, nbj, outside_bad, xr);
		if (R == 4) { AM5(4) } else if (R == 2) { AM5(2) } else { AM5(1) }
#undef AM5
		if (ranged) *ranged = true;
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
	if (sc->done) {
		if (threadIdx.x == 0) sc->xpending = 0;
		return;
	}
	if (sc->diverged) {
		// the iteration that diverged has finished its search update (the reference throws after it): stop everything queued
		// behind it.  (This iteration's ApplyMatrix has already run; it only wrote tmp.)
		if (threadIdx.x == 0) {
			sc->done = 1;
			sc->xpending = 0;
		}
		return;
	}
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
		sc->xpending = 1;
	}
}
// residual += (-alpha)*tmp ; [PC_NONE: tmp = residual] ; partial min/max (or sum of squares) of the new residual: k_cg_axpy2 without
// the dst update, which mf_cg_solve leaves to k_cg_update_search_x (one pass over `search` less per iteration)
// (Folding this pass into the loader waves of the forward MIC sweep -- they read residual and tmp anyway -- was built and measured:
// bit-exact, the 30 us of this kernel go away and the sweep gets 21 us slower (its middle third already streams ~4 TB/s): 75.5 ms per
// 256^3 step either way.  Not kept.)
// EDOT (liquid scenes): the MIC sweeps leave bundles of rows without fluid out, so tmp is in those cells what it is here and their share
// of dot(M^-1 r, r) = sum tmp * r_new can be formed in this pass, which streams both anyway (bempty / nbj: mic_empty_map; sx % 4 == 0,
// so a quad lies in one row); the shares go to epart[block] and are appended to the sweep's partials (whose entries for those bundles
// are 0).  A kernel of its own for them cost 33 us per iteration in the 379 x 356 x 124 dam break.
template <bool COPY_TMP, bool EDOT = false>
__global__ void __launch_bounds__(BLOCK)
k_cg_axpy_r(int64_t n, const CgScalars* __restrict__ sc, float* __restrict__ residual, float* __restrict__ tmp, float* __restrict__ fpart,
            double* __restrict__ dpart, const int* __restrict__ bempty = nullptr, int nbj = 0, int sx = 0, int sy = 0,
            double* __restrict__ epart = nullptr, const int* __restrict__ outside_bad = nullptr, const int* __restrict__ xr = nullptr, int nt = 0) {
	if (sc->done) return;
	// outside_bad[0] == 0 (k_cg_outside_zero): residual and tmp are zero in the bundles the sweeps leave out and stay so -- their quads
	// are neither read nor written, they enter the min / max as the zeros they are and add nothing to the sums
	const bool zero_outside = EDOT && outside_bad[0] == 0;
	const int xlo = (EDOT && xr) ? (xr[0] & ~7) : 0, xhi = (EDOT && xr) ? ((xr[1] + 7) & ~7) : 0x7fffffff;
	const float nalpha = sc->nalpha;
	const bool l2 = sc->useL2 != 0;
	float lo = FLT_MAX, hi = -FLT_MAX;
	double ss = 0.0, es = 0.0;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		bool in_empty = false;
		if (EDOT) {
			const int64_t row = (4 * q) / sx;
			const int j = (int)(row % sy), k = (int)(row / sy);
			in_empty = bempty[(k >> 3) * nbj + (j >> 3)] != 0;
			const int x = (int)(4 * q - row * sx);
			if (zero_outside && (in_empty || x < xlo || x >= xhi)) {
				lo = fminf(lo, 0.f);
				hi = fmaxf(hi, 0.f);
				continue;
			}
		}
		float4 r = ((float4*)residual)[q];
		const float4 t = nt ? ld_nt4(tmp + 4 * q) : ((const float4*)tmp)[q];      // (non-temporal here alone: 64.4 vs 63.7 ms per step)
		r.x = r.x + nalpha * t.x; r.y = r.y + nalpha * t.y; r.z = r.z + nalpha * t.z; r.w = r.w + nalpha * t.w;
		((float4*)residual)[q] = r;
		if (COPY_TMP) ((float4*)tmp)[q] = r;
		if (EDOT) {
			if (in_empty) {
				es += (double)(t.x * r.x);
				es += (double)(t.y * r.y);
				es += (double)(t.z * r.z);
				es += (double)(t.w * r.w);
			}
		}
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
	if (EDOT) {
		__syncthreads();
		es = block_sum(es);
		if (threadIdx.x == 0) epart[blockIdx.x] = es;
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

// dst += alpha*search (conjugategrad.cpp:254, left over from this iteration: xpending) and, unless the iteration has converged,
// search = tmp + beta*search (:283) -- `search` is read once for both
// SKIP (liquid scenes, see k_cg_outside_zero): dst, search and tmp are zero in the bundles without fluid and stay zero
template <bool SKIP>
__global__ void __launch_bounds__(BLOCK)
k_cg_update_search_x(int64_t n, const CgScalars* __restrict__ sc, float* __restrict__ dst, float* __restrict__ search, const float* __restrict__ tmp,
                     const int* __restrict__ bempty = nullptr, int nbj = 0, int sx = 0, int sy = 0, const int* __restrict__ outside_bad = nullptr,
                     const int* __restrict__ xr = nullptr, int nt = 0) {
	if (!sc->xpending) return;
	const bool upd = !sc->done;
	const int xlo = (SKIP && xr) ? (xr[0] & ~7) : 0, xhi = (SKIP && xr) ? ((xr[1] + 7) & ~7) : 0x7fffffff;
	const float alpha = sc->alpha, beta = sc->beta;
	const int64_t n4 = n >> 2;
	const bool zero_outside = SKIP && outside_bad[0] == 0;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		if (SKIP && zero_outside) {
			const int64_t row = (4 * q) / sx;
			const int j = (int)(row % sy), k = (int)(row / sy);
			const int xc = (int)(4 * q - row * sx);
			if (bempty[(k >> 3) * nbj + (j >> 3)] || xc < xlo || xc >= xhi) continue;
		}
		float4 s = nt ? ld_nt4(search + 4 * q) : ((float4*)search)[q];
		// the pressure is touched here and nowhere else in an iteration, tmp for the last time before ApplyMatrix overwrites it: non-temporal,
		// so that 128 MB per 256^3 iteration do not displace what the sweeps and the stencil re-read (63.1 vs 63.7 ms per step of tools/micro/ntv_bench.py)
		float4 x = nt ? ld_nt4(dst + 4 * q) : ((float4*)dst)[q];
		x.x = x.x + alpha * s.x; x.y = x.y + alpha * s.y; x.z = x.z + alpha * s.z; x.w = x.w + alpha * s.w;
		if (nt) st_nt4(dst + 4 * q, x);
		else ((float4*)dst)[q] = x;
		if (upd) {
			const float4 t = nt ? ld_nt4(tmp + 4 * q) : ((const float4*)tmp)[q];
			s.x = t.x + beta * s.x; s.y = t.y + beta * s.y; s.z = t.z + beta * s.z; s.w = t.w + beta * s.w;
			if (nt) st_nt4(search + 4 * q, s);
			else ((float4*)search)[q] = s;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		const float s = search[i];
		dst[i] = dst[i] + alpha * s;
		if (upd) search[i] = tmp[i] + beta * s;
	}
}

// the same for views that do not start on a 16-byte boundary
__global__ void __launch_bounds__(BLOCK)
k_cg_update_search_x_scalar(int64_t n, const CgScalars* __restrict__ sc, float* __restrict__ dst, float* __restrict__ search, const float* __restrict__ tmp) {
	if (!sc->xpending) return;
	const bool upd = !sc->done;
	const float alpha = sc->alpha, beta = sc->beta;
	for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
		const float s = search[i];
		dst[i] = dst[i] + alpha * s;
		if (upd) search[i] = tmp[i] + beta * s;
	}
}
// z-slab PCG scalar steps on the gathered per-rank reductions g[world][2] = {max|residual|, dot} (rows combined in rank order: the
// same bits on every rank; conjugategrad.cpp:250-291), on a CgScalars block: `done` mirrors the stop state, `xpending` says that this
// iteration's x += alpha * search is still to be done by the search update
__global__ void k_slab_alpha_x(const double* __restrict__ g, int world, CgScalars* __restrict__ sc, const int32_t* __restrict__ state) {
	if (state && state[0]) {
		sc->alpha = 0.f;
		sc->nalpha = -0.f;
		sc->xpending = 0;
		sc->done = 1;
		return;
	}
	double acc = 0.0;
	for (int r = 0; r < world; r++) acc += g[2 * r + 1];
	const float dp = (float)acc;
	const float a = (fabs((double)dp) > 0.) ? sc->sigma / dp : 0.f;
	sc->alpha = a;
	sc->nalpha = -a;
	sc->xpending = 1;
	sc->done = 0;
	sc->sigmaPrev = sc->sigma;
}
// k_slab_alpha_x + the residual update (k_cg_axpy_r, min / max partials) in one launch: every block forms alpha from the gathered rows
// itself (the same bits everywhere), block 0 publishes the scalars for the kernels behind it.  Nothing this kernel writes is read by
// another block of it (sigma and the stop state are written by the search update's kernel only).
__global__ void __launch_bounds__(BLOCK)
k_slab_axpy_r(int64_t n, const double* __restrict__ g, int world, CgScalars* __restrict__ sc, const int32_t* __restrict__ state,
              float* __restrict__ residual, const float* __restrict__ tmp, float* __restrict__ fpart, int nt) {
	const bool stopped = state[0] != 0;
	float a = 0.f;
	if (!stopped) {
		double acc = 0.0;
		for (int r = 0; r < world; r++) acc += g[2 * r + 1];
		const float dp = (float)acc;
		a = (fabs((double)dp) > 0.) ? sc->sigma / dp : 0.f;
	}
	const float nalpha = -a;
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		sc->alpha = a;
		sc->nalpha = nalpha;
		sc->xpending = stopped ? 0 : 1;
		sc->done = stopped ? 1 : 0;
		if (!stopped) sc->sigmaPrev = sc->sigma;
	}
	if (stopped) return;
	float lo = FLT_MAX, hi = -FLT_MAX;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		float4 r = ((float4*)residual)[q];
		const float4 t = nt ? ld_nt4(tmp + 4 * q) : ((const float4*)tmp)[q];
		r.x = r.x + nalpha * t.x; r.y = r.y + nalpha * t.y; r.z = r.z + nalpha * t.z; r.w = r.w + nalpha * t.w;
		((float4*)residual)[q] = r;
		lo = fminf(fminf(lo, r.x), fminf(r.y, fminf(r.z, r.w)));
		hi = fmaxf(fmaxf(hi, r.x), fmaxf(r.y, fmaxf(r.z, r.w)));
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		const float r = residual[i] + nalpha * tmp[i];
		residual[i] = r;
		lo = fminf(lo, r);
		hi = fmaxf(hi, r);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		fpart[2 * blockIdx.x] = lo;
		fpart[2 * blockIdx.x + 1] = hi;
	}
}
__global__ void k_slab_beta_x(const double* __restrict__ g, int world, CgScalars* __restrict__ sc, float accuracy, int iter, int32_t* __restrict__ state) {
	if (state && state[0]) return;
	double acc = 0.0, mx = 0.0;
	for (int r = 0; r < world; r++) {
		acc += g[2 * r + 1];
		mx = g[2 * r] > mx ? g[2 * r] : mx;
	}
	const float sigmaNew = (float)acc;
	const float rn = (float)mx;
	sc->resNorm = rn;
	sc->beta = sigmaNew / sc->sigma;
	sc->sigma = sigmaNew;
	if (state) {
		if (rn < accuracy) {
			state[0] = 1;
			state[1] = iter;
			sc->done = 1;
		} else if (!(rn < 1e35f)) {
			state[0] = 2;
			state[1] = iter;
			sc->done = 1;
		}
	}
}
// k_slab_beta_x + k_cg_update_search_x in one launch, the same way: every block forms beta and the stopping test from the gathered rows,
// block 0 publishes them (sigma, the stop state) -- the other blocks read sigmaPrev / xpending / alpha, which the residual update's
// kernel wrote
__global__ void __launch_bounds__(BLOCK)
k_slab_update_search_x(int64_t n, const double* __restrict__ g, int world, CgScalars* __restrict__ sc, float accuracy, int iter,
                       int32_t* __restrict__ state, float* __restrict__ dst, float* __restrict__ search, const float* __restrict__ tmp, int nt) {
	if (!sc->xpending) return;          // stopped before this iteration
	double acc = 0.0, mx = 0.0;
	for (int r = 0; r < world; r++) {
		acc += g[2 * r + 1];
		mx = g[2 * r] > mx ? g[2 * r] : mx;
	}
	const float sigmaNew = (float)acc;
	const float rn = (float)mx;
	const float beta = sigmaNew / sc->sigmaPrev;
	const bool converged = rn < accuracy, diverged = !converged && !(rn < 1e35f);
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		sc->resNorm = rn;
		sc->beta = beta;
		sc->sigma = sigmaNew;
		if (converged || diverged) {
			state[0] = converged ? 1 : 2;
			state[1] = iter;
			sc->done = 1;
		}
	}
	const bool upd = !(converged || diverged);
	const float alpha = sc->alpha;
	const int64_t n4 = n >> 2;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		float4 s = nt ? ld_nt4(search + 4 * q) : ((float4*)search)[q];
		float4 x = nt ? ld_nt4(dst + 4 * q) : ((float4*)dst)[q];
		x.x = x.x + alpha * s.x; x.y = x.y + alpha * s.y; x.z = x.z + alpha * s.z; x.w = x.w + alpha * s.w;
		if (nt) st_nt4(dst + 4 * q, x);
		else ((float4*)dst)[q] = x;
		if (upd) {
			const float4 t = nt ? ld_nt4(tmp + 4 * q) : ((const float4*)tmp)[q];
			s.x = t.x + beta * s.x; s.y = t.y + beta * s.y; s.z = t.z + beta * s.z; s.w = t.w + beta * s.w;
			if (nt) st_nt4(search + 4 * q, s);
			else ((float4*)search)[q] = s;
		}
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		const float s = search[i];
		dst[i] = dst[i] + alpha * s;
		if (upd) search[i] = tmp[i] + beta * s;
	}
}

// one-block finishers of the slab entry points
__global__ void __launch_bounds__(BLOCK) k_fin_sum(int nb, const double* __restrict__ partials, double* __restrict__ out) {
	double acc = strided_sum(partials, nb);
	acc = block_sum(acc);
	if (threadIdx.x == 0) out[0] = acc;
}
__global__ void __launch_bounds__(BLOCK) k_fin_maxabs(int nb, const float* __restrict__ fpart, double* __restrict__ out) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int i = threadIdx.x; i < nb; i += blockDim.x) {
		lo = fminf(lo, fpart[2 * i]);
		hi = fmaxf(hi, fpart[2 * i + 1]);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		const float alo = fabsf(lo), ahi = fabsf(hi);
		out[0] = (double)(alo > ahi ? alo : ahi);
	}
}

// cgSolveDiffusion matrix set-up, conjugategrad.cpp:364-375
__global__ void __launch_bounds__(BLOCK)
k_diffusion_matrix(int64_t n, const int32_t* __restrict__ flags, float* __restrict__ A0, float* __restrict__ Ai, float* __restrict__ Aj,
                   float* __restrict__ Ak, float alpha) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= n) return;
	if (flags[idx] & MF_OBSTACLE) {
		Ai[idx] = Aj[idx] = Ak[idx] = 0.f;
		A0[idx] = 1.f;
	} else {
		Ai[idx] *= alpha;
		Aj[idx] *= alpha;
		Ak[idx] *= alpha;
		A0[idx] = A0[idx] * alpha + 1.f;      // two roundings (-ffp-contract=off), as "A0 *= alpha; A0 += 1." in the reference
	}
}

static int cg_solve_core(const Dim& d, const int32_t* flags, float* dst, const float* rhs, float* residual, float* search, float* tmp,
                         const float* A0, const float* Ai, const float* Aj, const float* Ak, float* Aprecond, int pc, float accuracy,
                         int maxIter, int useL2Norm, float* out_host, void* stream, const unsigned char* free_pack);
// the system of mf_cg_solve with its rows padded from sx to px cells (pad cells: obstacle, zero coefficients, zero rhs), and back
__global__ void __launch_bounds__(BLOCK)
k_pad_system(int sx, int px, int64_t np_, const int32_t* __restrict__ flags, const float* __restrict__ rhs, const float* __restrict__ A0,
             const float* __restrict__ Ai, const float* __restrict__ Aj, const float* __restrict__ Ak, int32_t* __restrict__ pf, float* __restrict__ pr,
             float* __restrict__ p0, float* __restrict__ pi, float* __restrict__ pj, float* __restrict__ pk) {
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < np_; q += (int64_t)gridDim.x * BLOCK) {
		const int64_t row = q / px;
		const int i = (int)(q - row * px);
		const bool in = i < sx;
		const int64_t s = row * sx + i;
		pf[q] = in ? flags[s] : MF_OBSTACLE;
		pr[q] = in ? rhs[s] : 0.f;
		p0[q] = in ? A0[s] : 0.f;
		pi[q] = in ? Ai[s] : 0.f;
		pj[q] = in ? Aj[s] : 0.f;
		pk[q] = in ? Ak[s] : 0.f;
	}
}
__global__ void __launch_bounds__(BLOCK)
k_unpad(int sx, int px, int64_t n, const float* __restrict__ padded, float* __restrict__ out) {
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n; q += (int64_t)gridDim.x * BLOCK) {
		const int64_t row = q / sx;
		out[q] = padded[row * px + (q - row * sx)];
	}
}

extern "C" {

int mf_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                    const float* Ai, const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	bool a0p = false;
	const unsigned char* pk = mic_pack_user(flags, A0, Ai, Aj, Ak, &a0p);
	return launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, (hipStream_t)stream, nullptr, pk, 0, 0x7fffffff, nullptr, a0p);
}

static int time_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                             const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* stream, bool packed) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	const unsigned char* pack = nullptr;
	bool a0p = false;
	if (packed) {
		MF_TRY(mic_pack_query(d, flags, A0, Ai, Aj, Ak, &pack, &a0p, st));
		if (!pack) return fail("mf_time_apply_matrix_packed: no packed coefficients for these grids (call mf_mic_init on them; the off-diagonals must all be +0 or -1)");
	}
	hipEvent_t e0, e1;
	MF_HIP(hipEventCreate(&e0));
	MF_HIP(hipEventCreate(&e1));
	for (int i = 0; i < 3; i++) MF_TRY(launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, st, nullptr, pack, 0, 0x7fffffff, nullptr, a0p));
	MF_HIP(hipEventRecord(e0, st));
	for (int i = 0; i < reps; i++) MF_TRY(launch_apply_matrix<false>(d, flags, dst, src, A0, Ai, Aj, Ak, nullptr, nullptr, st, nullptr, pack, 0, 0x7fffffff, nullptr, a0p));
	MF_HIP(hipEventRecord(e1, st));
	MF_HIP(hipEventSynchronize(e1));
	float ms = 0.f;
	MF_HIP(hipEventElapsedTime(&ms, e0, e1));
	*avg_us = (double)ms * 1000.0 / (reps > 0 ? reps : 1);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	return 0;
}
int mf_time_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                         const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* stream) {
	return time_apply_matrix(sx, sy, sz, flags, dst, src, A0, Ai, Aj, Ak, reps, avg_us, stream, false);
}
int mf_time_apply_matrix_packed(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                                const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* stream) {
	return time_apply_matrix(sx, sy, sz, flags, dst, src, A0, Ai, Aj, Ak, reps, avg_us, stream, true);
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

// ---- fused pieces of the z-slab PCG (mantaflow_amd/slab.py): scalars live in a CgScalars-layout device block ----
int mf_apply_matrix_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                            const float* Ai, const float* Aj, const float* Ak, int k0, int k1, const void* scalars, double* dot_dev,
                            void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	if (k0 < 0 || k1 > sz || k0 > k1) return fail("mf_apply_matrix_dot_dev: invalid plane range");
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const CgScalars* sc = (const CgScalars*)scalars;
	int nb = 0;
	bool ranged = false;
	bool a0p = false;
	const unsigned char* pk = mic_pack_user(flags, A0, Ai, Aj, Ak, &a0p);
	MF_TRY(launch_apply_matrix<true>(d, flags, dst, src, A0, Ai, Aj, Ak, ws->partials, sc, st, &nb, pk, k0, k1, &ranged, a0p));
	if (!ranged) {
		// the fallback kernels sum over the whole grid: redo the dot over the requested planes
		const int64_t XY = (int64_t)sx * sy, n = (int64_t)(k1 - k0) * XY;
		nb = blocks_for(n >> 2, BLOCK, 2048);
		hipLaunchKernelGGL(k_cg_dot, dim3(nb), dim3(BLOCK), 0, st, n, sc, dst + k0 * XY, src + k0 * XY, ws->partials);
	}
	hipLaunchKernelGGL(k_fin_sum, dim3(1), dim3(BLOCK), 0, st, nb, ws->partials, dot_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_cg_slab_axpy2(int64_t n, const void* scalars, float* x, const float* search, float* residual, const float* tmp,
                     double* maxabs_dev, void* stream) {
	if (n <= 0) return 0;
	hipStream_t st = (hipStream_t)stream;
	if (!(al16(x) && al16(search) && al16(residual) && al16(tmp))) {
		// views that do not start on a 16-byte boundary (odd plane sizes): the unfused sequence
		const float* alpha_dev = &((const CgScalars*)scalars)->alpha;
		MF_TRY(mf_grid_scaled_add_dev(n, x, search, alpha_dev, 1.f, stream));
		MF_TRY(mf_grid_scaled_add_dev(n, residual, tmp, alpha_dev, -1.f, stream));
		return mf_grid_max_abs_dev_f64(n, residual, maxabs_dev, stream);
	}
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n >> 2, BLOCK, 2048);
	hipLaunchKernelGGL((k_cg_axpy2<false>), dim3(nb), dim3(BLOCK), 0, st, n, (const CgScalars*)scalars, x, search, residual, (float*)tmp, ws->fpartials, ws->partials + MAX_BLOCKS);
	hipLaunchKernelGGL(k_fin_maxabs, dim3(1), dim3(BLOCK), 0, st, nb, ws->fpartials, maxabs_dev);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_diffusion_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak, float alpha,
                        void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const int64_t n = (int64_t)sx * sy * sz;
	hipLaunchKernelGGL(k_diffusion_matrix, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, n, flags, A0, Ai, Aj, Ak, alpha);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_cg_last_shortcut(int32_t* out) {
	out[0] = g_last_shortcut[0];
	out[1] = g_last_shortcut[1];
	out[2] = g_last_shortcut[2];
	return 0;
}
int mf_cg_solve(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* rhs, float* residual, float* search,
                float* tmp, const float* A0, const float* Ai, const float* Aj, const float* Ak, float* Aprecond, int pc,
                float accuracy, int maxIter, int useL2Norm, float* out_host, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	if (pc != MF_PC_NONE && pc != MF_PC_MICP) return fail("GridCg<APPLYMAT>::setICPreconditioner: Invalid method specified.");
	if (pc == MF_PC_MICP && !d.is3d) pc = MF_PC_NONE;  // conjugategrad.cpp:315-321
	if (!(al16(dst) && al16(rhs) && al16(residual) && al16(search) && al16(tmp))) return fail("mf_cg_solve: work grids must be 16-byte aligned");
	if (maxIter <= 0) {
		// GridCg::solve (conjugategrad.cpp:302-307) never reaches iterate(), and doInit is only called from there: pressure
		// and the work grids stay untouched; mIterations = 0, mResNorm = 1e20 (constructor, :203)
		out_host[0] = 0.f;
		out_host[1] = 1e20f;
		out_host[2] = 0.f;
		return 0;
	}
	// Row lengths that are not a multiple of 8 (benchmark_dam.py: 3.2 res + 8) would take the sweeps and ApplyMatrix off their
	// 16-byte paths and off the packed coefficient bytes (379 x 356 x 124: 0.83 instead of ~0.5 ms per iteration).  The PCG then
	// runs on its own copy of the system with the rows padded to the next multiple of 8: pad cells are obstacle cells with zero
	// coefficients and zero rhs, so they pass zeros through ApplyMatrix and the sweeps and add zeros to every reduction -- the
	// iterates of the fluid cells are those of the unpadded system.  (A fluid cell in the last column, which MakeLaplaceMatrix
	// never writes, would read the pad cell instead of the next row's first cell, both times a zero coefficient: the same value.)
	static const bool nopad = getenv("MF_CG_NOPAD") != nullptr;
	if (pc == MF_PC_MICP && (sx % 8) != 0 && sx >= 16 && !nopad) {
		const int px = (sx + 7) & ~7;
		const int64_t rows = (int64_t)sy * sz, np_ = (int64_t)px * rows;
		if (np_ < ((int64_t)1 << 31)) {
			int dev = 0;
			MF_HIP(hipGetDevice(&dev));
			static float* pad_buf[16] = {};
			static int64_t pad_cap[16] = {};
			if (np_ > pad_cap[dev & 15]) {
				MF_HIP(hipStreamSynchronize(st));
				if (pad_buf[dev & 15]) MF_HIP(hipFree(pad_buf[dev & 15]));
				pad_cap[dev & 15] = ((np_ + np_ / 8 + 63) / 64) * 64;       // every array of the block 256-byte aligned
				MF_HIP(hipMalloc((void**)&pad_buf[dev & 15], sizeof(float) * 11 * pad_cap[dev & 15]));
			}
			float* b = pad_buf[dev & 15];
			const int64_t cap = pad_cap[dev & 15];
			float *p_flags = b, *p_dst = b + cap, *p_rhs = b + 2 * cap, *p_res = b + 3 * cap, *p_search = b + 4 * cap, *p_tmp = b + 5 * cap,
			      *p_A0 = b + 6 * cap, *p_Ai = b + 7 * cap, *p_Aj = b + 8 * cap, *p_Ak = b + 9 * cap, *p_Ap = b + 10 * cap;
			// the work grids are fresh (zeroed) temp grids in solvePressureSystem, and the algorithm relies on it: the sweeps never
			// write a non-fluid cell of tmp, ApplyMatrix copies search there, and the dots run over all cells
			MF_HIP(hipMemsetAsync(p_dst, 0, sizeof(float) * np_, st));
			MF_HIP(hipMemsetAsync(p_res, 0, sizeof(float) * (2 * cap + np_), st));      // residual, search, tmp
			MF_HIP(hipMemsetAsync(p_Ap, 0, sizeof(float) * np_, st));
			hipLaunchKernelGGL(k_pad_system, dim3(blocks_for(np_, BLOCK, 4096)), dim3(BLOCK), 0, st, sx, px, np_, flags, rhs, A0, Ai, Aj, Ak, (int32_t*)p_flags, p_rhs,
			                   p_A0, p_Ai, p_Aj, p_Ak);
			MF_LAUNCH_CHECK();
			MF_TRY(mf_cg_solve(px, sy, sz, (const int32_t*)p_flags, p_dst, p_rhs, p_res, p_search, p_tmp, p_A0, p_Ai, p_Aj, p_Ak, p_Ap, pc, accuracy,
			                   maxIter, useL2Norm, out_host, stream));
			hipLaunchKernelGGL(k_unpad, dim3(blocks_for(d.n, BLOCK, 4096)), dim3(BLOCK), 0, st, sx, px, d.n, p_dst, dst);
			MF_LAUNCH_CHECK();
			return 0;
		}
	}
	return cg_solve_core(d, flags, dst, rhs, residual, search, tmp, A0, Ai, Aj, Ak, Aprecond, pc, accuracy, maxIter, useL2Norm, out_host, stream, nullptr);
}

}  // extern "C"

// Liquid scenes: most 8 x 8 bundles of rows hold no fluid cell (benchmark_dam.py: 6 % of the cells are fluid).  MakeRhs leaves rhs zero
// outside the fluid and the work grids are fresh zero grids, so every PCG vector is zero in those bundles and stays zero (ApplyMatrix copies
// src, the sweeps leave them out, the vector updates combine zeros) -- the kernels of an iteration then skip them altogether.  This kernel
// is what establishes the premise, on the device, once per solve: bad[0] counts the quads of empty bundles in which rhs or tmp is not +0.
// x-range of the packed system (bytes of k_mic_pack / k_setup_fused; sx % 8 == 0): xr[0] = first cell with a non-zero byte, xr[1] = one
// past the last one, plus one (a set "Ai is -1" bit couples the cell behind it).  Cells outside are non-fluid cells without couplings.
__global__ void __launch_bounds__(BLOCK)
k_pack_xrange(int64_t n, int sx, const unsigned char* __restrict__ pack, int* __restrict__ xr) {
	int lo = 0x7fffffff, hi = 0;
	const int64_t n8 = n >> 3;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n8; q += (int64_t)gridDim.x * BLOCK) {
		const unsigned long long w = ((const unsigned long long*)pack)[q];
		if (w == 0ull) continue;
		const int x0 = (int)((8 * q) % sx);
		const int a = x0 + (__ffsll((long long)w) - 1) / 8, b = x0 + (63 - __clzll((long long)w)) / 8 + 2;
		lo = a < lo ? a : lo;
		hi = b > hi ? b : hi;
	}
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) {
		const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
		lo = l2 < lo ? l2 : lo;
		hi = h2 > hi ? h2 : hi;
	}
	if ((threadIdx.x & 63) == 0 && hi > 0) {
		atomicMin(&xr[0], lo);
		atomicMax(&xr[1], hi);
	}
}
__global__ void __launch_bounds__(BLOCK)
k_cg_outside_zero(int64_t n, int sx, int sy, const int* __restrict__ bempty, int nbj, const float* __restrict__ rhs, const float* __restrict__ tmp,
                  const float* __restrict__ search, int* __restrict__ bad, const int* __restrict__ xr) {
	const int64_t n4 = n >> 2;
	int found = 0;
	// (xr: also the cells outside the x-range of the packed system, rounded outwards to chunks of 8 -- what the trimmed sweeps leave out)
	const int xlo = xr ? (xr[0] & ~7) : 0, xhi = xr ? ((xr[1] + 7) & ~7) : sx;
	for (int64_t q = blockIdx.x * (int64_t)BLOCK + threadIdx.x; q < n4; q += (int64_t)gridDim.x * BLOCK) {
		const int64_t row = (4 * q) / sx;
		const int j = (int)(row % sy), k = (int)(row / sy);
		const int x = (int)(4 * q - row * sx);
		if (!bempty[(k >> 3) * nbj + (j >> 3)] && x >= xlo && x < xhi) continue;
		const uint4 a = ((const uint4*)rhs)[q], b = ((const uint4*)tmp)[q], c = ((const uint4*)search)[q];
		if ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w | c.x | c.y | c.z | c.w) != 0u) found = 1;
	}
	if (__any(found) && (threadIdx.x & 63) == 0) atomicAdd(bad, 1);
}
// doInit + iterate loop of GridCg (conjugategrad.cpp:210-307).  free_pack (mf_solve_pressure_fused): the system exists as packed bytes
// only -- A0 / Ai / Aj / Ak are null, the MIC factor is already in Aprecond and the system is registered with the sweeps
static int cg_solve_core(const Dim& d, const int32_t* flags, float* dst, const float* rhs, float* residual, float* search, float* tmp,
                         const float* A0, const float* Ai, const float* Aj, const float* Ak, float* Aprecond, int pc, float accuracy,
                         int maxIter, int useL2Norm, float* out_host, void* stream, const unsigned char* free_pack) {
	const int sx = d.sx, sy = d.sy, sz = d.sz;
	const int64_t n = d.n;
	hipStream_t st = (hipStream_t)stream;
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
		if (!free_pack) MF_TRY(mf_mic_init(sx, sy, sz, flags, Aprecond, A0, Ai, Aj, Ak, stream));
		MF_TRY(mf_mic_apply(sx, sy, sz, flags, tmp, residual, Aprecond, Ai, Aj, Ak, stream));
	} else {
		MF_HIP(hipMemcpyAsync(tmp, residual, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
	}
	MF_HIP(hipMemcpyAsync(search, tmp, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
	MF_HIP(hipMemsetAsync(sc, 0, sizeof(CgScalars), st));
	hipLaunchKernelGGL(k_cg_dot, dim3(nbs), dim3(BLOCK), 0, st, n, sc, tmp, residual, p_sig);
	hipLaunchKernelGGL(k_cg_begin, dim3(1), dim3(BLOCK), 0, st, sc, nbs, p_sig, accuracy, useL2Norm);
	MF_LAUNCH_CHECK();
	// ApplyMatrix reads the same packed coefficient bytes as the MIC sweeps when mf_mic_init found the matrix packable
	const unsigned char* am_pack = nullptr;
	bool am_a0p = false;
	if (free_pack) {
		am_pack = free_pack;
		am_a0p = true;
	} else if (pc == MF_PC_MICP) {
		MF_TRY(mic_pack_query(d, flags, A0, Ai, Aj, Ak, &am_pack, &am_a0p, st));
	}

	// ---- iterate, conjugategrad.cpp:238-299; the host only polls `done`, one batch behind the batch it has just queued
	// (every kernel of an iteration returns at once when `done` is already set, so running ahead costs a few empty
	// launches after convergence and keeps the GPU from idling between iterations) ----
	// liquid scenes whose sweeps leave bundles out: their dot shares ride on the residual update (k_cg_axpy_r<.., EDOT>)
	const int* be_map = nullptr;
	int be_nbj = 0;
	if (pc == MF_PC_MICP && (sx % 4) == 0) MF_TRY(mic_empty_map(d, flags, Aprecond, Aj, Ak, &be_map, &be_nbj, st));
	const int be_nb = ((sy + 7) / 8) * ((sz + 7) / 8);      // the sweep's partials (one per bundle) come first, the nbs of the residual update behind them
	int* p_bad = (int*)((char*)ws->scalars + WS_PCG_FLAGS);
	// the map the vector kernels skip by: the same one, also where the sweep sums the shares of the empty bundles itself (<= one bundle per CU)
	const int* sk_map = be_map;
	int sk_nbj = be_nbj;
	if (!sk_map && pc == MF_PC_MICP && (sx % 4) == 0) MF_TRY(mic_empty_map(d, flags, Aprecond, Aj, Ak, &sk_map, &sk_nbj, st, true));
	// ... and the sweeps keep to the x-range of the fluid (the packed bytes tell it), if the host may know that everything outside is +0
	struct TrimGuard {
		~TrimGuard() { mic_set_trim(0, 0); }
	} trim_guard;
	g_last_shortcut[0] = g_last_shortcut[1] = g_last_shortcut[2] = 0;
	const int* xr_dev = nullptr;      // device x-range of the packed system, for the kernels that skip by it
	const int pcg_nt = (!sk_map && n > PCG_NT_CELLS) ? 1 : 0;      // the vector streams around the sweeps non-temporal (see PCG_NT_CELLS)
	if (sk_map) {
		// (n % 4 == 0 here: sx % 4 == 0.)  residual = rhs and dst = 0 were set above; tmp and search are the caller's
		int* p_xr = p_bad + 1;
		const bool ranged = am_pack != nullptr && (sx % 8) == 0;
		if (ranged) xr_dev = p_xr;
		const int init3[3] = {0, 0x7fffffff, 0};
		MF_HIP(hipMemcpyAsync(p_bad, init3, sizeof init3, hipMemcpyHostToDevice, st));
		if (ranged) hipLaunchKernelGGL(k_pack_xrange, dim3(blocks_for(n >> 3, BLOCK, 2048)), dim3(BLOCK), 0, st, n, sx, am_pack, p_xr);
		hipLaunchKernelGGL(k_cg_outside_zero, dim3(nbs), dim3(BLOCK), 0, st, n, sx, sy, sk_map, sk_nbj, rhs, tmp, search, p_bad, ranged ? p_xr : nullptr);
		MF_LAUNCH_CHECK();
		static const bool notrim = getenv("MF_MIC_NOTRIM") != nullptr;
		if (ranged && !notrim) {
			int h3[3] = {1, 0, 0};
			MF_HIP(hipMemcpyAsync(h3, p_bad, sizeof h3, hipMemcpyDeviceToHost, st));
			MF_HIP(hipStreamSynchronize(st));
			int c0 = h3[1] >> 3, c1 = (h3[2] + 7) >> 3;
			if (c1 > sx / 8) c1 = sx / 8;
			if (h3[0] == 0 && h3[2] > 0 && c1 > c0) {
				mic_set_trim(8 * c0, c1 - c0);
				g_last_shortcut[0] = 1;
				if (c1 - c0 < sx / 8) {
					g_last_shortcut[1] = 8 * c0;
					g_last_shortcut[2] = 8 * (c1 - c0);
				}
			}
		}
	}
	const int batch = (pc == MF_PC_MICP && mic_mode() == 0) ? 1 : 4;
	CgScalars h;
	memset(&h, 0, sizeof h);
	h.resNorm = 1e20f;
	// one event pair per device (an event belongs to the device that was current when it was created)
	static thread_local hipEvent_t ev_dev[16][2] = {};
	int dev_ = 0;
	MF_HIP(hipGetDevice(&dev_));
	hipEvent_t* ev = ev_dev[dev_ & 15];
	if (!ev[0]) {
		MF_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
		MF_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
	}
	CgScalars* hslot = (CgScalars*)ws->host;   // two pinned slots
	int issued = 0, slot = 0, pending = -1;
	while (issued < maxIter) {
		const int todo = (maxIter - issued < batch) ? (maxIter - issued) : batch;
		for (int it = 0; it < todo; it++) {
			int nba = 0, nsig = 0;
			bool beta_done = false;
			MF_TRY(launch_apply_matrix<true>(d, flags, tmp, search, A0, Ai, Aj, Ak, p_dot, sc, st, &nba, am_pack, 0, 0x7fffffff, nullptr, am_a0p, sk_map,
			                                 sk_nbj, sk_map ? p_bad : nullptr, xr_dev, pcg_nt));
			hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(BLOCK), 0, st, sc, nba, p_dot);
			if (pc == MF_PC_MICP) {
				if (sk_map)      // (without be_map the shares it writes behind the sweep's partials are not summed: the sweep has them)
					hipLaunchKernelGGL((k_cg_axpy_r<false, true>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, residual, tmp, p_mm, p_res, sk_map, sk_nbj, sx, sy, p_sig + be_nb, p_bad, xr_dev, pcg_nt);
				else
					hipLaunchKernelGGL((k_cg_axpy_r<false>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, residual, tmp, p_mm, p_res, (const int*)nullptr, 0, 0, 0,
					                   (double*)nullptr, (const int*)nullptr, (const int*)nullptr, pcg_nt);
				MF_TRY(mic_launch(1, d, flags, tmp, residual, Aprecond, Ai, Aj, Ak, sc, st));
				// sigma_new = dot(tmp, residual) comes out of the backward sweep's write-back (one partial per row bundle); the shares of
				// the bundles the sweeps leave out: from the residual update above, behind the sweep's partials
				// ... and the beta step is the tail of the sweep's last workgroup (one launch less per iteration)
				MF_TRY(mic_launch_dot(d, flags, tmp, residual, Aprecond, Ai, Aj, Ak, sc, p_sig, &nsig, st, be_map != nullptr,
				                      BetaTail{sc, nbs, p_mm, p_res, be_map ? nbs : 0}, &beta_done));
				if (be_map) {
					if (nsig != be_nb) return fail("mf_cg_solve: the backward sweep wrote %d dot partials, %d expected", nsig, be_nb);
					nsig += nbs;
				}
			} else {
				hipLaunchKernelGGL((k_cg_axpy_r<true>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, residual, tmp, p_mm, p_res);
			}
			if (nsig == 0) {
				hipLaunchKernelGGL(k_cg_dot, dim3(nbs), dim3(BLOCK), 0, st, n, sc, tmp, residual, p_sig);
				nsig = nbs;
			}
			if (!beta_done) hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(BLOCK), 0, st, sc, nbs, p_mm, p_res, nsig, p_sig);
			if (sk_map)
				hipLaunchKernelGGL((k_cg_update_search_x<true>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, dst, search, tmp, sk_map, sk_nbj, sx, sy, p_bad, xr_dev, pcg_nt);
			else
				hipLaunchKernelGGL((k_cg_update_search_x<false>), dim3(nbs), dim3(BLOCK), 0, st, n, sc, dst, search, tmp, (const int*)nullptr, 0, 0, 0, (const int*)nullptr,
				                   (const int*)nullptr, pcg_nt);
		}
		MF_LAUNCH_CHECK();
		issued += todo;
		MF_HIP(hipMemcpyAsync(&hslot[slot], sc, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		MF_HIP(hipEventRecord(ev[slot], st));
		if (pending >= 0) {
			MF_HIP(hipEventSynchronize(ev[pending]));
			memcpy(&h, &hslot[pending], sizeof h);
			if (h.done || h.diverged) break;
		}
		pending = slot;
		slot ^= 1;
	}
	// the final state, after everything that was queued
	MF_HIP(hipMemcpyAsync(&hslot[0], sc, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	memcpy(&h, &hslot[0], sizeof h);
	out_host[0] = (float)h.iterations;
	out_host[1] = h.resNorm;
	out_host[2] = h.sigma;
	if (pc == MF_PC_MICP) MF_TRY(mic_flow_error());
	if (h.diverged) return fail("GridCg::iterate: The CG solver diverged, residual norm > 1e30, stopping.");
	return 0;
}

// ---- matrix-free set-up of a plain MakeLaplaceMatrix system (no fractions, no ghost fluid, no optional rhs terms): ONE pass over
// flags and vel writes rhs (MakeRhs, pressure.cpp:32-84), the packed byte of every cell -- fluid bit, "Ai / Aj / Ak is -1" bits and the
// integer diagonal in bits 4-7 (MakeLaplaceMatrix, conjugategrad.h:154-187; k_mic_pack's layout) -- and clears the empty-bundle entry
// of every 8 x 8 bundle of rows that holds a fluid cell.  The float arrays A0 / Ai / Aj / Ak never exist.  A thread owns 4 cells.
__global__ void __launch_bounds__(BLOCK)
k_setup_fused(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ rhs, unsigned char* __restrict__ pack,
              int* __restrict__ bempty, int nbj) {
	const int qx = d.sx >> 2;
	const int64_t T = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (T >= (int64_t)qx * d.sy * d.sz) return;
	const int qi = (int)(T % qx);
	const int64_t rg = T / qx;
	const int j = (int)(rg % d.sy), k = (int)(rg / d.sy), i0 = 4 * qi;
	const int64_t Y = d.Y, Z = d.Z, n = d.n;
	const int64_t idx = i0 + Y * j + Z * k;
	const int4 f4 = *(const int4*)(flags + idx);
	const int f[4] = {f4.x, f4.y, f4.z, f4.w};
	float r[4] = {0.f, 0.f, 0.f, 0.f};
	unsigned bytes = 0;
	const bool row_in = j >= 1 && j <= d.sy - 2 && k >= 1 && k <= d.sz - 2;
	if (row_in) {
		const int4 ym4 = *(const int4*)(flags + idx - Y), yp4 = *(const int4*)(flags + idx + Y);
		const int4 zm4 = *(const int4*)(flags + idx - Z), zp4 = *(const int4*)(flags + idx + Z);
		const int ym[4] = {ym4.x, ym4.y, ym4.z, ym4.w}, yp[4] = {yp4.x, yp4.y, yp4.z, yp4.w};
		const int zm[4] = {zm4.x, zm4.y, zm4.z, zm4.w}, zp[4] = {zp4.x, zp4.y, zp4.z, zp4.w};
		const int fxm = i0 > 0 ? flags[idx - 1] : MF_OBSTACLE, fxp = i0 + 4 < d.sx ? flags[idx + 4] : MF_OBSTACLE;
		const float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
		const float4 ux4 = *(const float4*)(vx + idx), uy4 = *(const float4*)(vy + idx), uyp4 = *(const float4*)(vy + idx + Y);
		const float4 uz4 = *(const float4*)(vz + idx), uzp4 = *(const float4*)(vz + idx + Z);
		const float uxp = i0 + 4 < d.sx ? vx[idx + 4] : 0.f;
		const float ux[5] = {ux4.x, ux4.y, ux4.z, ux4.w, uxp};
		const float uy[4] = {uy4.x, uy4.y, uy4.z, uy4.w}, uyp[4] = {uyp4.x, uyp4.y, uyp4.z, uyp4.w};
		const float uz[4] = {uz4.x, uz4.y, uz4.z, uz4.w}, uzp[4] = {uzp4.x, uzp4.y, uzp4.z, uzp4.w};
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const int i = i0 + c;
			const bool fl = (f[c] & MF_FLUID) != 0;
			unsigned b = fl ? 1u : 0u;
			if (fl && i >= 1 && i <= d.sx - 2) {
				const int xm = c > 0 ? f[c - 1] : fxm, xp = c < 3 ? f[c + 1] : fxp;
				const unsigned a0 = (unsigned)!(xm & MF_OBSTACLE) + (unsigned)!(xp & MF_OBSTACLE) + (unsigned)!(ym[c] & MF_OBSTACLE) +
				                    (unsigned)!(yp[c] & MF_OBSTACLE) + (unsigned)!(zm[c] & MF_OBSTACLE) + (unsigned)!(zp[c] & MF_OBSTACLE);
				b |= ((xp & MF_FLUID) ? 2u : 0u) | ((yp[c] & MF_FLUID) ? 4u : 0u) | ((zp[c] & MF_FLUID) ? 8u : 0u) | (a0 << 4);
				float set = ux[c] - ux[c + 1] + uy[c] - uyp[c];
				set += uz[c] - uzp[c];
				r[c] = set;
			}
			bytes |= b << (8 * c);
		}
	} else {
#pragma unroll
		for (int c = 0; c < 4; c++) bytes |= ((f[c] & MF_FLUID) ? 1u : 0u) << (8 * c);
	}
	*(float4*)(rhs + idx) = make_float4(r[0], r[1], r[2], r[3]);
	*(unsigned*)(pack + idx) = bytes;
	if (bytes & 0x01010101u) bempty[(k >> 3) * nbj + (j >> 3)] = 0;
}

extern "C" {

int mf_solve_pressure_fused(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* pressure, float* rhs, float* residual,
                            float* search, float* tmp, float* Aprecond, float accuracy, int maxIter, int useL2Norm, float* out_host,
                            void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	if (!d.is3d || (sx % 8) != 0 || mic_mode() != 2) return fail("mf_solve_pressure_fused: needs a 3D grid with sx % 8 == 0 and the \"rows\" sweeps");
	if (!(al16(flags) && al16(vel) && al16(pressure) && al16(rhs) && al16(residual) && al16(search) && al16(tmp) && al16(Aprecond)))
		return fail("mf_solve_pressure_fused: grids must be 16-byte aligned");
	unsigned char* pack;
	int* bempty;
	int nbj;
	MF_TRY(mic_fused_begin(d, st, &pack, &bempty, &nbj));
	const int64_t nthr = (int64_t)(sx >> 2) * sy * sz;
	hipLaunchKernelGGL(k_setup_fused, dim3((unsigned)((nthr + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d, flags, vel, rhs, pack, bempty, nbj);
	MF_LAUNCH_CHECK();
	MF_TRY(mic_fused_finish(d, flags, Aprecond, st));
	if (maxIter <= 0) {
		out_host[0] = 0.f;
		out_host[1] = 1e20f;
		out_host[2] = 0.f;
		return 0;
	}
	// the sweeps never write a non-fluid cell of tmp: it has to start from zeros (a fresh temp grid in the reference)
	MF_HIP(hipMemsetAsync(tmp, 0, sizeof(float) * d.n, st));
	return cg_solve_core(d, flags, pressure, rhs, residual, search, tmp, nullptr, nullptr, nullptr, nullptr, Aprecond, MF_PC_MICP, accuracy, maxIter,
	                     useL2Norm, out_host, stream, pack);
}


// the stretches of a z-slab PCG iteration between two communication points, queued with one call each (CgScalars layout: sigma,
// alpha, nalpha, beta, resNorm at floats 0, 1, 2, 3, 4); x += alpha * search rides on the search update as in mf_cg_solve
int mf_cg_slab_after_dp(const double* gathered, int world, void* scalars, const int32_t* state_dev, int64_t own_off, int64_t n_own,
                        float* residual, float* tmp, double* maxabs_dev, int sx, int sy, int sz,
                        const int32_t* flags, const float* Aprecond, const float* Ai, const float* Aj, const float* Ak, double* dot_dev,
                        void* stream) {
	hipStream_t st = (hipStream_t)stream;
	CgScalars* sc = (CgScalars*)scalars;
	float* r = residual + own_off;
	float* t = tmp + own_off;
	const bool one_launch = n_own > 0 && state_dev && al16(r) && al16(t);
	if (!one_launch) hipLaunchKernelGGL(k_slab_alpha_x, dim3(1), dim3(1), 0, st, gathered, world, sc, state_dev);
	if (n_own > 0) {
		if (one_launch) {
			Workspace* ws;
			MF_TRY(get_workspace(&ws));
			const int nb = blocks_for(n_own >> 2, BLOCK, 2048);
			// the alpha step rides in the residual update (every block forms it from the gathered rows)
			hipLaunchKernelGGL(k_slab_axpy_r, dim3(nb), dim3(BLOCK), 0, st, n_own, gathered, world, sc, state_dev, r, t, ws->fpartials, n_own > PCG_NT_CELLS ? 1 : 0);
			MF_LAUNCH_CHECK();
			// max |r| of these partials and dot(tmp, r): folded by the last workgroup of the backward sweep (no one-block launches)
			MF_TRY(check_dim(sx, sy, sz));
			const Dim d = mkdim(sx, sy, sz);
			if (!d.is3d) return fail("mICP only supports 3D grids so far");
			return mic_apply_dot_fold(d, flags, tmp, residual, Aprecond, Ai, Aj, Ak, dot_dev, nb, ws->fpartials, maxabs_dev, sc, st);
		} else {
			// views that do not start on a 16-byte boundary (odd plane sizes): the unfused sequence (alpha is 0 once stopped)
			MF_TRY(mf_grid_scaled_add_dev(n_own, r, t, &sc->alpha, -1.f, stream));
			MF_TRY(mf_grid_max_abs_dev_f64(n_own, r, maxabs_dev, stream));
		}
	}
	return mf_mic_apply_dot_dev(sx, sy, sz, flags, tmp, residual, Aprecond, Ai, Aj, Ak, dot_dev, stream);
}
int mf_cg_slab_after_zr(const double* gathered, int world, void* scalars, float accuracy, int iter, int32_t* state_dev, int64_t own_off,
                        int64_t n_own, float* x, float* search, const float* tmp, void* stream) {
	hipStream_t st = (hipStream_t)stream;
	CgScalars* sc = (CgScalars*)scalars;
	float* xs = x + own_off;
	float* s = search + own_off;
	const float* t = tmp + own_off;
	if (n_own > 0 && state_dev && al16(xs) && al16(s) && al16(t)) {
		// the beta step and the stopping test ride in the search update
		hipLaunchKernelGGL(k_slab_update_search_x, dim3(blocks_for(n_own >> 2, BLOCK, 2048)), dim3(BLOCK), 0, st, n_own, gathered, world, sc,
		                   accuracy, iter, state_dev, xs, s, t, n_own > PCG_NT_CELLS ? 1 : 0);
		MF_LAUNCH_CHECK();
		return 0;
	}
	hipLaunchKernelGGL(k_slab_beta_x, dim3(1), dim3(1), 0, st, gathered, world, sc, accuracy, iter, state_dev);
	if (n_own <= 0) return 0;
	hipLaunchKernelGGL(k_cg_update_search_x_scalar, dim3(blocks_for(n_own, BLOCK, 2048)), dim3(BLOCK), 0, st, n_own, sc, xs, s, t);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
