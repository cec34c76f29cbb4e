// mic.hip -- the MIC(0) preconditioner of the pressure solve on gfx950: InitPreconditionModifiedIncompCholesky2 and the two
// substitution sweeps of ApplyPreconditionModifiedIncompCholesky2 (source/conjugategrad.cpp:66-97, 135-159), which the
// reference runs as single-threaded lexicographic sweeps.  Two parallelisations with identical results: "rows" (k_mic_rows /
// k_mic_rows_init, one dataflow launch per sweep, the default) and "levels" (k_mic_tiles, one launch per tile hyperplane).
#include "common.h"
#include "pressure.h"
#include <float.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>
#include <vector>

using namespace mf;

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

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
// Single-launch ("dataflow") sweeps: work items are handed out in dependency order by an atomic ticket (a ticketed item's
// predecessors hold smaller tickets, i.e. they are already running or finished, so a wait always ends whatever the dispatch
// order -- ONE queue: that argument needs nothing about which workgroups are resident), and the faces an item hands to its
// successors travel as 8-byte {value, tag} granules written with ONE agent-scope (sc1, write-through) store each and polled
// with agent-scope (sc1, L1-bypassing) loads: no flag, no fence (MI355X_MICROARCH.md, "handoff-1to1").  tag = launch
// generation, so the exchange buffers never need clearing.
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
constexpr int FLOW_SPIN_LIMIT = 1 << 21;

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
	unsigned long long W;     // packed path: 8 cells x {fluid, Ai == -1, Aj == -1, Ak == -1} bits
};
// wave 0 computes, 1-3 load (chunk n -> wave 1 + n % 3), 4 writes back, 5 polls the incoming faces, 6 publishes the outgoing ones
constexpr int ROWS_THREADS = 448;
// steps of operands held in LDS (8 per chunk): the loader waves may run ROWS_RING / 8 - 1 chunks ahead of the write-back
#ifndef ROWS_RING
#define ROWS_RING 64
#endif
#ifndef ROWS_NAP
#define ROWS_NAP 1
#endif
template <int MODE, bool VEC>
__global__ void __launch_bounds__(ROWS_THREADS)
k_mic_rows(Dim d, int nbj, int nbk, int jb, int nstreams, int nchunks, int xoff0, const int* __restrict__ order, FlowCtl* ctl, int* xt,
           unsigned long long* xj, unsigned long long* xk, unsigned gen, const int32_t* __restrict__ flags,
           float* __restrict__ dst, const float* __restrict__ var1, const float* __restrict__ Ap,
           const float* __restrict__ Ai, const float* __restrict__ Aj, const float* __restrict__ Ak,
           const CgScalars* __restrict__ sc, double* __restrict__ dotpart, const int* __restrict__ bempty,
           const unsigned char* __restrict__ pack, const int* __restrict__ pack_ok, int empty_ext, BetaTail tail) {
	// empty_ext: the dot shares of the bundles left out are summed by the caller's residual update (mf_cg_solve) -- their entries are 0
	static_assert(MODE == 1 || MODE == 2, "row-streaming kernel implements the apply sweeps");
	// dotpart (backward sweep only): GridDotProduct(dst, var1) (conjugategrad.cpp:175-178: fp32 product, fp64 sum) fused into the
	// write-back wave, one partial per bundle (and x-block) at dotpart[sid] -- the sum the PCG needs right after this sweep
	const bool with_dot = (MODE == 2) && dotpart != nullptr;
	if (bempty && bempty[nbj * nbk] == 0) bempty = nullptr;     // no empty bundle (smoke scenes): no per-bundle look-ups
	// packed operands (k_mic_pack, built by mf_mic_init when every Ai / Aj / Ak is exactly 0 or -1, as MakeLaplaceMatrix writes
	// them): one byte per cell replaces the flags + Ai + Aj + Ak streams (16 B per cell) of both sweeps; the values are rebuilt
	// exactly, so the arithmetic is unchanged.  What the sweeps wait for is the hand-off between bundles, and that waits behind
	// the operand stream of its own CU -- fewer outstanding HBM requests make every hand-off shorter.
	const bool packed = pack != nullptr && pack_ok[0] != 0;
	constexpr bool REV = (MODE == 2);
	if (sc && sc->done) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane & 7;
	// (A Gray-coded lane -> k mapping with DPP row_ror:8 / v_permlane16_swap / v_permlane32_swap instead of the ds_bpermute for the
	// k-neighbour was tried twice -- tools/micro/permlane_swap.hip: bit-exact, 591 instead of 572 us per apply mid-round and 431 instead
	// of 412 us on the final kernel: the swaps and the selects behind them are no shorter than the one LDS permute.)
	const int c = lane >> 3;
	const int skew = b + c;
	// the compute wave is the critical path: everything else yields to it
	if (wave == 0) __builtin_amdgcn_s_setprio(3);
	else __builtin_amdgcn_s_setprio(0);
	__shared__ float4 sA[ROWS_RING * 64];   // {fluid ? rhs : dst  (-> result), Ai, Aj, Ak}     index = ((h + 2) & (ROWS_RING - 1)) * 64 + lane
	__shared__ float2 sB[ROWS_RING * 64];   // {Aprecond, fluid}
	__shared__ float sR[MODE == 2 ? ROWS_RING * 64 : 1];   // with_dot: var1 of the cell
	__shared__ __attribute__((aligned(16))) float sFj[2][8][8];
	__shared__ __attribute__((aligned(16))) float sFk[2][8][8];   // face values of a block [block parity][face lane][step]
	// s_flags = {chunks committed by loader wave 1, 2, 3, face blocks published}: one 16-byte LDS word, so that the compute wave
	// reads all of them -- and, speculatively, the block's face values -- in ONE LDS round trip per block
	__shared__ __attribute__((aligned(16))) int s_flags[4];
	__shared__ int s_done, s_flushed, s_ticket;
	__shared__ int s_half, s_pub;      // half blocks finished by the compute wave / published by wave 6
	int* const s_ready = s_flags;
	const unsigned long long fresh0 = (unsigned long long)gen << 32;
	const int X8 = nchunks * 8;
	int spins = 0;

	for (;;) {
		if (threadIdx.x == 0) {
			// ONE ticket queue in dependency order (xt[0]): whoever holds the lowest unfinished ticket finds all its predecessors
			// drawn by workgroups that are already running, so the sweep ends whichever workgroups are resident (a second rank on
			// the device, a CU mask).  Per-XCD queues (round 2: 1 % at 256^3) gave that guarantee up and are gone.
			const int tl = atomicAdd(&xt[0], 1);
			s_ticket = tl < nstreams ? tl : nstreams;
			s_ready[0] = s_ready[1] = s_ready[2] = 0;
			s_done = 0;
			s_flushed = 0;
			s_flags[3] = 0;
			s_half = 0;
			s_pub = 0;
		}
		__syncthreads();
		const int t = s_ticket;
		if (t >= nstreams) break;
		const int pk = order[t];
		const int tjl = pk & 0xfff, tkl = (pk >> 12) & 0xfff, xb = pk >> 24;
		const int tj = REV ? nbj - 1 - tjl : tjl, tk = REV ? nbk - 1 - tkl : tkl;
		const int j = tj * 8 + (REV ? 7 - b : b), k = tk * 8 + (REV ? 7 - c : c);
		const bool row_in = (j < d.sy) && (k < d.sz);
		// x-block xb (mf_set_mic_blocking_x: the caller has zeroed the Ai coupling across the block faces): an independent
		// system of X8 cells starting at xoff; one x-block = the whole row = the reference algorithm.  xoff0 (mf_cg_solve, liquid
		// scenes): the rows are swept over the cells [xoff0, xoff0 + X8) only -- every cell outside is a non-fluid cell without couplings
		// (zero packed byte) whose value is +0 and stays +0, which is what the first / last cell inside would read from it
		const int xoff = xb * X8 + xoff0;
		const int xlim = d.sx - xoff < X8 ? d.sx - xoff : X8;          // cells of this x-block that exist
		const int64_t rowbase = d.Y * j + d.Z * k + xoff;

		// face granules: per bundle and face an array [producer step h + 2][face lane]: the 8 face lanes of one step (one
		// store instruction) fill exactly one 64-byte line, and a consumer lane's 8-step window maps to 8 consecutive lines
		// (lane (7,c) publishes x' at step x'+7+c, lane (b,7) at x'+b+7: the same steps 8m+5 .. 8m+12 for every face lane)
		// jb = bundles per independent j-block (mf_set_mic_blocking: the caller has zeroed the Aj coupling across block
		// faces, so nothing crosses them); jb == nbj: one block = the reference algorithm
		const int tj_pred = REV ? tj + 1 : tj - 1, tj_succ = REV ? tj - 1 : tj + 1;
		const int tk_pred = REV ? tk + 1 : tk - 1, tk_succ = REV ? tk - 1 : tk + 1;
		const int64_t sid = ((int64_t)xb * nbk + tkl) * nbj + tjl;
		// A bundle without fluid cells (bempty, k_bundle_empty) is not swept: its cells pass through unchanged and everything it
		// would hand to its neighbours multiplies a zero coefficient there, so neighbours neither wait for it nor publish to it.
		// (Its share of the fused dot(dst, var1) comes from k_mic_empty_dot, launched next to the backward sweep: summed here -- a lane
		// per row, 384 strided loads each -- it kept a workgroup off the ticket queue for tens of microseconds per empty bundle, and a
		// liquid scene has hundreds of them: the backward sweep of the 379 x 356 x 124 dam break took 166 us against 103 us forward.)
		// Where every workgroup draws one ticket only (nstreams <= gridDim.x) nobody waits for this workgroup, and the sum stays here (one
		// launch less per iteration: 128^3).
		if (bempty && bempty[tk * nbj + tj]) {
			if (MODE == 2 && with_dot && wave == 0 && empty_ext) {
				if (lane == 0) part_store(&dotpart[sid], 0.0);
			} else if (MODE == 2 && with_dot && wave == 0 && nstreams <= (int)gridDim.x) {
				double dacc = 0.0;
				if (row_in)
					for (int x = xlim - 1; x >= 0; x--) dacc += (double)(dst[rowbase + x] * var1[rowbase + x]);
#pragma unroll
				for (int o = 32; o >= 1; o >>= 1) dacc += __shfl_xor(dacc, o, 64);
				if (lane == 0) part_store(&dotpart[sid], dacc);
			}
			__syncthreads();
			continue;
		}
		const bool pj_live = (tjl > 0) && (tj / jb == tj_pred / jb) && !(bempty && bempty[tk * nbj + tj_pred]);
		const bool pk_live = (tkl > 0) && !(bempty && bempty[tk_pred * nbj + tj]);
		const bool sj_live = (tjl + 1 < nbj) && (tj / jb == tj_succ / jb) && !(bempty && bempty[tk * nbj + tj_succ]);
		const bool sk_live = (tkl + 1 < nbk) && !(bempty && bempty[tk_succ * nbj + tj]);
		const int64_t XP = X8 + 2 * ROWS_PAD;
		if (wave == 6) {
			// ================= face publisher: the compute wave only leaves its results in the ring; this wave turns the outer rows /
			// columns of every finished half block into face granules -- (val * Aj) * Aprecond resp. (val * Ak) * Aprecond, the same
			// two fp32 products the compute wave feeds to its inner neighbours (forward sweep), or val itself (backward sweep).
			// One store instruction per half block: lane = {face, face lane, step}.  (The compute wave is bound by the number of
			// instructions it issues; the nine per step that published the faces are gone from it.)
			const int fsel = lane >> 5, idx = (lane >> 2) & 7, st = lane & 3;
			const int L = fsel == 0 ? idx * 8 + 7 : 56 + idx;            // source lane (b, c) = (7, idx) resp. (idx, 7)
			const int skewL = (L & 7) + (L >> 3);
			const bool live = fsel == 0 ? sj_live : sk_live;
			unsigned long long* outp = (fsel == 0 ? xj : xk) + sid * XP * 8 + idx;
			const int nhalf = 2 * (nchunks + 2);
#pragma unroll 1
			for (int n = 0; n < nhalf; n++) {
				while (__hip_atomic_load(&s_half, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < n + 1) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				const int h2 = 4 * n + st;                                  // ring row (= step + 2) of this lane's granule
				const int xg = h2 - 2 - skewL;                              // its cell
				if (live && (unsigned)xg < (unsigned)X8) {
					const int slot = (h2 & (ROWS_RING - 1)) * 64 + L;
					const float4 cA = sA[slot];
					float fv = cA.x;
					if (MODE == 1) {
						const float p = sB[slot].x;
						const float t = cA.x * (fsel == 0 ? cA.z : cA.w);
						fv = packed ? t : t * p;
					}
					granule_store(outp + (int64_t)h2 * 8, fv, gen);
				}
				// the ring rows may be overwritten now (LDS-only release: the granule stores need not have been acknowledged)
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
				__hip_atomic_store(&s_pub, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else if (wave == 5) {
			// ================= face poller: the only wave that loads granules (and it never stores to global memory) =====
			// (Tried on top of this: a second poller wave out of phase with the first (497 instead of 468 us per apply at 256^3), two
			// polls in flight in this wave with hand-written loads and vmcnt waits (the compiler copies the load targets before the
			// wait: wrong values), four scalar-path poller waves (tools/micro/rows_scalar_poller.diff: bit-exact, 700+ us).)
			// One load instruction fetches a whole half window of BOTH faces: lane = {face, step within the half window, face lane}
			// (4 steps x 8 face lanes = 32 granules per face).  Two instructions per poll instead of sixteen 8-lane ones: the
			// round trip of a poll is that of its slowest load, and every load queues behind the CU's operand stream.
			const int pf_ = lane >> 5, pr = (lane >> 3) & 3, pidx = lane & 7;
			const bool plive = pf_ == 0 ? pj_live : pk_live;
			const unsigned long long* in_f = (pf_ == 0 ? xj + (sid - 1) * XP * 8 : xk + (sid - nbj) * XP * 8) + pidx;     // + (h + 2) * 8
			float* const sF = pf_ == 0 ? &sFj[0][0][0] : &sFk[0][0][0];       // [block parity][face lane][step]
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				// the consumer lane of this granule (b, c) = (0, pidx) resp. (pidx, 0) works on x' = 8m - 2 - pidx + step
				const int x0 = 8 * m - 2 - pidx + pr;
				const bool in0 = plive && (unsigned)x0 < (unsigned)X8, in1 = plive && (unsigned)(x0 + 4) < (unsigned)X8;
				const unsigned long long* p0 = in_f + (int64_t)(8 * m + 7 + pr) * 8;
				unsigned long long g0 = fresh0, g1 = fresh0;
				int pub = 0;
				for (;;) {
					// tags only grow: a half window (4 steps) is complete when every granule in range carries this sweep's generation.
					// The halves are published separately: the compute wave starts a block on the first one, i.e. a consumer bundle
					// runs 11 instead of 15 steps behind its producer (7 steps of skew + the granularity of the hand-off).
					// (Quarter windows -- 9 steps of lag, four flag checks and four publisher rounds per block -- were built as well: bit-exact
					// and 530 instead of 396 us per 256^3 apply.)
					if (pub == 0 && in0) g0 = granule_load(p0);
					if (in1) g1 = granule_load(p0 + 4 * 8);
					const bool giveup = ++spins > FLOW_SPIN_LIMIT;
					if (pub == 0 && (__all((unsigned)(g0 >> 32) == gen) || giveup)) {
						// the buffer of this parity was read by block m-2
						if (m >= 2) {
							while (__hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < m - 1) {
								if (++spins > FLOW_SPIN_LIMIT) break;
								__builtin_amdgcn_s_sleep(ROWS_NAP);
							}
						}
						sF[((m & 1) * 8 + pidx) * 8 + pr] = __uint_as_float((unsigned)g0);
						__hip_atomic_store(&s_flags[3], 2 * m + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						pub = 1;
					}
					if (pub == 1 && (__all((unsigned)(g1 >> 32) == gen) || giveup)) {
						sF[((m & 1) * 8 + pidx) * 8 + 4 + pr] = __uint_as_float((unsigned)g1);
						__hip_atomic_store(&s_flags[3], 2 * m + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						break;
					}
				}
			}
		} else if (wave != 0) {
			// ================= memory waves: waves 1-3 load + commit, wave 4 writes finished chunks back =================
			// (separate waves because vmcnt is one in-order counter per wave: a wave that has loads of several chunks or
			// loads and stores in flight ends up waiting for all of them)
			auto chunk_geom = [&](int m, int64_t& rowidx, int& nv) {
				const int x0 = (REV ? nchunks - 1 - m : m) * 8;
				const int nvx = xlim - x0 < 8 ? xlim - x0 : 8;             // <= 0 past the end of the row
				nv = row_in ? nvx : 0;
				rowidx = rowbase + x0;
			};
			auto wait_for = [&](int* flag, int need) {
				while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(ROWS_NAP);
				}
			};
			if (wave <= 3) {
				// forward sweep, packed operands: a chunk whose 8 cells are all fluid does not need the old dst (it only passes
				// through non-fluid cells).  Its packed word is fetched one round ahead (wnext, with the previous chunk of this
				// wave), so the decision costs no extra round trip; the first chunk of a wave loads dst unconditionally.
				unsigned long long wnext = 0ull;
				bool have_wnext = false;
				auto issue = [&](RowsChunk& r, int m) {
					int64_t rowidx;
					int nv;
					chunk_geom(m, rowidx, nv);
					bool need_dst = true;
					if (packed) {
						if (MODE == 1 && have_wnext) {
							r.W = wnext;
							need_dst = (r.W & 0x0101010101010101ull) != 0x0101010101010101ull;
						} else {
							r.W = nv > 0 ? *(const unsigned long long*)(pack + rowidx) : 0ull;     // sx % 8 == 0: nv is 8 or <= 0
						}
						if (MODE == 1 && m + 3 < nchunks) {
							int64_t ridx3;
							int nv3;
							chunk_geom(m + 3, ridx3, nv3);
							wnext = nv3 > 0 ? *(const unsigned long long*)(pack + ridx3) : 0ull;
							have_wnext = true;
						}
					} else {
						load_row8i<VEC, REV>(flags, rowidx, nv, r.F);
						load_row8<VEC, REV>(Ai, rowidx, nv, r.Ai);
						load_row8<VEC, REV>(Aj, rowidx, nv, r.Aj);
						load_row8<VEC, REV>(Ak, rowidx, nv, r.Ak);
					}
					if (MODE == 1 || with_dot) load_row8<VEC, REV>(var1, rowidx, nv, r.V);
					load_row8<VEC, REV>(Ap, rowidx, nv, r.P);
					if (need_dst) load_row8<VEC, REV>(dst, rowidx, nv, r.D);
				};
				auto commit = [&](const RowsChunk& r, int m) {
					int64_t rowidx;
					int nv;
					chunk_geom(m, rowidx, nv);
					const int p0 = 8 * m + skew + 2;
#pragma unroll
					for (int a = 0; a < 8; a++) {
						const bool in = ((REV ? 7 - a : a) < nv);
						bool fl;
						float cai, caj, cak;
						if (packed) {
							const unsigned bits = (unsigned)(r.W >> (8 * (REV ? 7 - a : a))) & 0xffu;
							fl = (bits & 1u) != 0;
							cai = (bits & 2u) ? -1.f : 0.f;
							caj = (bits & 4u) ? -1.f : 0.f;
							cak = (bits & 8u) ? -1.f : 0.f;
						} else {
							fl = in && (r.F[a] & MF_FLUID);
							cai = r.Ai[a];
							caj = r.Aj[a];
							cak = r.Ak[a];
						}
						const int idx = ((p0 + a) & (ROWS_RING - 1)) * 64 + lane;
						// packed operands: the ring holds A * Aprecond instead of A.  For A in {+0, -1} (what the packed bytes certify)
						// (x * A) * p == x * (A * p) bit for bit, for every x and p (signed zeros, infinities and NaNs included): the
						// compute wave saves a multiplication per neighbour and per step, on its dependency chain
						if (packed) {
							cai = cai * r.P[a];
							caj = caj * r.P[a];
							cak = cak * r.P[a];
						}
						sA[idx] = make_float4((MODE == 1 && fl) ? r.V[a] : r.D[a], cai, caj, cak);
						sB[idx] = make_float2(r.P[a], fl ? 1.f : 0.f);
						if (MODE == 2 && with_dot) sR[idx] = r.V[a];     // 0 outside the grid (load_row8)
					}
				};
				// one chunk in flight per loader wave: its loads are issued as soon as the previous chunk of this wave is
				// committed (three blocks before the compute wave needs them), then the wave waits for the ring rows
				RowsChunk R;
				const int w = wave - 1;
#pragma unroll 1
				for (int n = w; n < nchunks; n += 3) {
					issue(R, n);
					// the ring rows of chunk n were last used by chunk n - ROWS_RING / 8: it must have been written back
					if (n >= ROWS_RING / 8) wait_for(&s_flushed, n - ROWS_RING / 8 + 1);
					// ... and its outer rows / columns (finished with block n-2) must have been published
					if (n >= ROWS_RING / 8) wait_for(&s_pub, 2 * (n - ROWS_RING / 8 + 2) + 2);
					commit(R, n);
					__hip_atomic_store(&s_ready[w], n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			} else {
				double dacc = 0.0;
#pragma unroll 1
				for (int q = 0; q < nchunks; q++) {
					wait_for(&s_done, q + 3);   // chunk q is complete once block q+2 is finished
					int64_t rowidx;
					int nv;
					chunk_geom(q, rowidx, nv);
					const int p0 = 8 * q + skew + 2;
					float w[8];
#pragma unroll
					for (int e = 0; e < 8; e++) {
						const int slot = ((p0 + e) & (ROWS_RING - 1)) * 64 + lane;
						const float res = sA[slot].x;
						w[REV ? 7 - e : e] = res;
						if (MODE == 2 && with_dot) dacc += (double)(res * sR[slot]);
					}
					__hip_atomic_store(&s_flushed, q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
					// (non-temporal loads of Aprecond / var1 / dst or stores of dst make the sweeps slower -- 66.9-68.4 vs 63.7 ms per step: their
					// hand-offs want the operands in L2 / MALL)
					if (VEC) {
						if (nv > 0) *(float4*)(dst + rowidx) = make_float4(w[0], w[1], w[2], w[3]);
						if (nv > 4) *(float4*)(dst + rowidx + 4) = make_float4(w[4], w[5], w[6], w[7]);
					} else {
#pragma unroll
						for (int e = 0; e < 8; e++)
							if (e < nv) dst[rowidx + e] = w[e];
					}
				}
				if (MODE == 2 && with_dot) {
					// fixed butterfly over the 64 rows of the bundle: the same bits on every run
#pragma unroll
					for (int o = 32; o >= 1; o >>= 1) dacc += __shfl_xor(dacc, o, 64);
					if (lane == 0) part_store(&dotpart[sid], dacc);      // sc1 + completion: the last workgroup may fold it (tail)
				}
			}
		} else {
			// ================= compute wave: LDS in, LDS out (wave 6 publishes the faces) =================
			float oi0 = 0.f, oj0 = 0.f, ok0 = 0.f;
			float4 nA = sA[lane];                  // ring row of h = -2 (never valid)
			float2 nB = sB[lane];
			auto block = [&](int m, auto edge_tag, auto pre_tag) {
				constexpr bool EDGE = decltype(edge_tag)::value;
				constexpr bool PRE = decltype(pre_tag)::value;      // the ring holds A * Aprecond (packed operands)
				const int xq = 8 * m - 2 - skew;                       // this lane's x' at the first step of the block
				// flags and face values in one batch of LDS reads: the LDS serves a wave's requests in order, and the poller
				// publishes the values before the flag -- a flag read that shows block m is followed by reads that see its values
				typedef int rows_i4 __attribute__((ext_vector_type(4)));
				typedef float rows_f4 __attribute__((ext_vector_type(4)));
				typedef const volatile __attribute__((address_space(3))) rows_i4* lflags;
				typedef const volatile __attribute__((address_space(3))) rows_f4* lface;
				rows_f4 j0, j1, k0, k1;
				const int r3 = m % 3;
				bool have2;
				for (;;) {
					const rows_i4 fg = *(lflags)s_flags;
					j0 = *(lface)&sFj[m & 1][c][0];
					j1 = *(lface)&sFj[m & 1][c][4];
					k0 = *(lface)&sFk[m & 1][b][0];
					k1 = *(lface)&sFk[m & 1][b][4];
					const int rdy = r3 == 0 ? fg.x : (r3 == 1 ? fg.y : fg.z);
					have2 = fg.w >= 2 * m + 2;         // second half there as well: j1 / k1 are its values
					if (((m >= nchunks) || rdy >= m + 1) && fg.w >= 2 * m + 1) break;
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
				float gj[8], gk[8];
				gj[0] = j0.x; gj[1] = j0.y; gj[2] = j0.z; gj[3] = j0.w; gj[4] = j1.x; gj[5] = j1.y; gj[6] = j1.z; gj[7] = j1.w;
				gk[0] = k0.x; gk[1] = k0.y; gk[2] = k0.z; gk[3] = k0.w; gk[4] = k1.x; gk[5] = k1.y; gk[6] = k1.z; gk[7] = k1.w;
				const int base = (8 * m) & (ROWS_RING - 1);
#pragma unroll
				for (int s = 0; s < 8; s++) {
					if (s == 4 && !have2) {
						// second half of the block's face values
						for (;;) {
							const int fw = *(const volatile __attribute__((address_space(3))) int*)&s_flags[3];
							j1 = *(lface)&sFj[m & 1][c][4];
							k1 = *(lface)&sFk[m & 1][b][4];
							if (fw >= 2 * m + 2 || ++spins > FLOW_SPIN_LIMIT) break;
							__builtin_amdgcn_s_sleep(1);
						}
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
						gj[4] = j1.x; gj[5] = j1.y; gj[6] = j1.z; gj[7] = j1.w;
						gk[4] = k1.x; gk[5] = k1.y; gk[6] = k1.z; gk[7] = k1.w;
					}
					const float4 cA = nA;
					const float2 cB = nB;
					// base is a multiple of 8: the rows of a block never wrap inside it (immediate LDS offsets from one address)
					const int row = (base + s) * 64 + lane;
					const int nrow = (s < 7) ? (base + s + 1) * 64 + lane : ((base + 8) & (ROWS_RING - 1)) * 64 + lane;
					nA = sA[nrow];
					nB = sB[nrow];
					const float dj = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oj0), 0x111, 0xf, 0xf, true));   // row_shr:1; lanes b == 0 take the face value
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
						oi0 = valid ? (PRE ? val * ai : (val * ai) * p) : 0.f;
						oj0 = valid ? (PRE ? val * aj : (val * aj) * p) : 0.f;
						ok0 = valid ? (PRE ? val * ak : (val * ak) * p) : 0.f;
					} else {
						const float nv = PRE ? p * (val - ii0 * ai - ij0 * aj - ik0 * ak) : p * (val - ii0 * ai * p - ij0 * aj * p - ik0 * ak * p);
						val = fl ? nv : val;
						oi0 = oj0 = ok0 = valid ? val : 0.f;
					}
					if (valid) sA[row].x = val;
					if (s == 3) {
						// first half of the block is in the ring (the LDS serves this wave's writes in order: no wait needed)
						asm volatile("" ::: "memory");
						*(volatile __attribute__((address_space(3))) int*)&s_half = 2 * m + 1;
					}
				}
				// the block is in the ring (in-order LDS again: the counters are written after the results, without a wait)
				asm volatile("" ::: "memory");
				*(volatile __attribute__((address_space(3))) int*)&s_done = m + 1;
				*(volatile __attribute__((address_space(3))) int*)&s_half = 2 * m + 2;
			};
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				// interior: every lane's x' is inside [0, X8) for all 8 steps of the block
				const bool interior = (m >= 2 && m <= nchunks - 1);
				if (packed) {
					if (interior) block(m, std::false_type{}, std::true_type{});
					else block(m, std::true_type{}, std::true_type{});
				} else {
					if (interior) block(m, std::false_type{}, std::false_type{});
					else block(m, std::true_type{}, std::false_type{});
				}
			}
		}
		__syncthreads();
	}
	if (spins > FLOW_SPIN_LIMIT) atomicExch(&ctl->err, 1);
	__shared__ int s_lastwg;
	if (threadIdx.x == 0) {
		const int f = atomicAdd(&ctl->finished, 1);
		s_lastwg = (f == (int)gridDim.x - 1) ? 1 : 0;
		if (s_lastwg) {
			xt[0] = 0;
			ctl->finished = 0;
		}
	}
	if (MODE == 2 && !tail.sc && tail.sum_out) {
		// z-slab solver: the one-block folds behind the sweep (k_fin_maxabs_live, k_mic_fin_sum) in the workgroup that finished last
		__syncthreads();
		if (!s_lastwg) return;
		float lo = FLT_MAX, hi = -FLT_MAX;
		if (tail.nbr > 0) tail_minmax256(tail.fpart, tail.nbr, lo, hi);
		__syncthreads();
		const double dd = tail_sum256<true>(dotpart, tail.nsig);
		if (threadIdx.x == 0) {
			tail.sum_out[0] = dd;
			if (tail.nbr > 0 && !(tail.live && tail.live->done)) {
				const float alo = fabsf(lo), ahi = fabsf(hi);
				tail.maxabs_out[0] = (double)(alo > ahi ? alo : ahi);
			}
		}
		return;
	}
	if (MODE == 2 && tail.sc) {
		// the beta step (k_cg_beta) in the workgroup that finished last: every other workgroup's dot partials were complete before
		// it reported in.  Same folds in the same order as the one-block kernel (256 threads take part).
		__syncthreads();
		if (!s_lastwg) return;
		CgScalars* w = tail.sc;
		const bool l2 = w->useL2 != 0;
		float lo = FLT_MAX, hi = -FLT_MAX;
		double ss = 0.0;
		if (l2) ss = tail_sum256<false>(tail.dpart_res, tail.nbr);
		else tail_minmax256(tail.fpart, tail.nbr, lo, hi);
		__syncthreads();
		const double dd = tail_sum256<true>(dotpart, tail.nsig);
		if (threadIdx.x == 0) {
			float resNorm;
			if (l2)
				resNorm = (float)ss;
			else {
				const float alo = fabsf(lo), ahi = fabsf(hi);
				resNorm = alo > ahi ? alo : ahi;
			}
			w->resNorm = resNorm;
			if (resNorm < w->accuracy) {
				w->sigma = resNorm;
				w->done = 1;
			} else {
				const float sigmaNew = (float)dd;
				w->beta = sigmaNew / w->sigma;
				w->sigmaNew = sigmaNew;
				w->sigma = sigmaNew;
				if (!((double)resNorm < 1e35)) w->diverged = 1;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// InitPreconditionModifiedIncompCholesky2 (conjugategrad.cpp:66-97) as a row-streaming dataflow sweep: the forward sweep's roles and
// hand-off machinery (k_mic_rows), one launch instead of one per tile hyperplane (94 at 256^3).  Differences: the operands are A0,
// Ai, Aj, Ak and the flags (nothing is packed yet -- mf_mic_init builds the packed bytes from its own output), and TWO values cross
// every neighbour relation: (A_dir * Aprecond)^2 and A_dir * (A_o1 + A_o2) * Aprecond^2 of the predecessor cell -- two DPP moves,
// two LDS permutes, two granules per face lane and step (arrays xj / xj1, xk / xk1).  Per-cell arithmetic = k_mic_tiles<0>.
// ---------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ void __launch_bounds__(ROWS_THREADS)
k_mic_rows_init(Dim d, int nbj, int nbk, int jb, int nstreams, int nchunks, const int* __restrict__ order, FlowCtl* ctl, int* xt,
                unsigned long long* xj, unsigned long long* xk, unsigned long long* xj1, unsigned long long* xk1, unsigned gen,
                const int32_t* __restrict__ flags, float* __restrict__ dst, const float* __restrict__ A0,
                const float* __restrict__ Ai, const float* __restrict__ Aj, const float* __restrict__ Ak,
                const unsigned char* __restrict__ pack) {
	// pack (nullable): the matrix-free set-up of mf_solve_pressure_fused -- flags, the 0 / -1 off-diagonals and the integer diagonal of a
	// MakeLaplaceMatrix system as one byte per cell (k_mic_pack's layout); flags / A0 / Ai / Aj / Ak are then not read
	constexpr bool REV = false;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = lane & 7, c = lane >> 3;
	const int skew = b + c;
	if (wave == 0) __builtin_amdgcn_s_setprio(3);
	else __builtin_amdgcn_s_setprio(0);
	__shared__ float4 sA[ROWS_RING * 64];   // {fluid ? A0 : 0  (-> Aprecond), Ai, Aj, Ak}     index = ((h + 2) & (ROWS_RING - 1)) * 64 + lane
	__shared__ float sB[ROWS_RING * 64];    // fluid
	__shared__ __attribute__((aligned(16))) float sFj[2][2][8][8];     // [value][block parity][face lane][step]
	__shared__ __attribute__((aligned(16))) float sFk[2][2][8][8];
	__shared__ __attribute__((aligned(16))) int s_flags[4];
	__shared__ int s_done, s_flushed, s_ticket, s_half, s_pub;
	int* const s_ready = s_flags;
	const unsigned long long fresh0 = (unsigned long long)gen << 32;
	const int X8 = nchunks * 8;
	int spins = 0;

	for (;;) {
		if (threadIdx.x == 0) {
			// one ticket queue in dependency order, as in k_mic_rows
			const int tl = atomicAdd(&xt[0], 1);
			s_ticket = tl < nstreams ? tl : nstreams;
			s_ready[0] = s_ready[1] = s_ready[2] = 0;
			s_done = 0;
			s_flushed = 0;
			s_flags[3] = 0;
			s_half = 0;
			s_pub = 0;
		}
		__syncthreads();
		const int t = s_ticket;
		if (t >= nstreams) break;
		const int pk = order[t];
		const int tjl = pk & 0xfff, tkl = (pk >> 12) & 0xfff, xb = pk >> 24;
		const int tj = tjl, tk = tkl;
		const int j = tj * 8 + b, k = tk * 8 + c;
		const bool row_in = (j < d.sy) && (k < d.sz);
		const int xoff = xb * X8;
		const int xlim = d.sx - xoff < X8 ? d.sx - xoff : X8;
		const int64_t rowbase = d.Y * j + d.Z * k + xoff;
		const int64_t sid = ((int64_t)xb * nbk + tkl) * nbj + tjl;
		const bool pj_live = (tjl > 0) && (tj / jb == (tj - 1) / jb);
		const bool pk_live = (tkl > 0);
		const bool sj_live = (tjl + 1 < nbj) && (tj / jb == (tj + 1) / jb);
		const bool sk_live = (tkl + 1 < nbk);
		const int64_t XP = X8 + 2 * ROWS_PAD;
		if (wave == 6) {
			// ================= face publisher: both hand-off values of the outer rows / columns, recomputed from the ring =========
			const int fsel = lane >> 5, idx = (lane >> 2) & 7, st = lane & 3;
			const int L = fsel == 0 ? idx * 8 + 7 : 56 + idx;
			const int skewL = (L & 7) + (L >> 3);
			const bool live = fsel == 0 ? sj_live : sk_live;
			unsigned long long* out0 = (fsel == 0 ? xj : xk) + sid * XP * 8 + idx;
			unsigned long long* out1 = (fsel == 0 ? xj1 : xk1) + sid * XP * 8 + idx;
			const int nhalf = 2 * (nchunks + 2);
#pragma unroll 1
			for (int n = 0; n < nhalf; n++) {
				while (__hip_atomic_load(&s_half, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < n + 1) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				const int h2 = 4 * n + st;
				const int xg = h2 - 2 - skewL;
				if (live && (unsigned)xg < (unsigned)X8) {
					const int slot = (h2 & (ROWS_RING - 1)) * 64 + L;
					const float4 cA = sA[slot];
					const float ap = cA.x;
					const float adir = fsel == 0 ? cA.z : cA.w;
					const float osum = fsel == 0 ? (cA.y + cA.w) : (cA.y + cA.z);
					const float t_ = adir * ap;
					const float ap2 = ap * ap;
					const float h0 = t_ * t_;
					const float h1 = adir * osum * ap2;
					granule_store(out0 + (int64_t)h2 * 8, h0, gen);
					granule_store(out1 + (int64_t)h2 * 8, h1, gen);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
				__hip_atomic_store(&s_pub, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
		} else if (wave == 5) {
			// ================= face poller: one load per half window, face and value array =================
			const int pf_ = lane >> 5, pr = (lane >> 3) & 3, pidx = lane & 7;
			const bool plive = pf_ == 0 ? pj_live : pk_live;
			const int64_t foff = (pf_ == 0 ? (sid - 1) : (sid - nbj)) * XP * 8 + pidx;
			const unsigned long long* in_f0 = (pf_ == 0 ? xj : xk) + foff;
			const unsigned long long* in_f1 = (pf_ == 0 ? xj1 : xk1) + foff;
			float* const sF0 = pf_ == 0 ? &sFj[0][0][0][0] : &sFk[0][0][0][0];
			float* const sF1 = pf_ == 0 ? &sFj[1][0][0][0] : &sFk[1][0][0][0];
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				const int x0 = 8 * m - 2 - pidx + pr;
				const bool in0 = plive && (unsigned)x0 < (unsigned)X8, in1 = plive && (unsigned)(x0 + 4) < (unsigned)X8;
				const int64_t r0 = (int64_t)(8 * m + 7 + pr) * 8;
				unsigned long long g0a = fresh0, g0b = fresh0, g1a = fresh0, g1b = fresh0;
				int pub = 0;
				for (;;) {
					if (pub == 0 && in0) {
						g0a = granule_load(in_f0 + r0);
						g0b = granule_load(in_f1 + r0);
					}
					if (in1) {
						g1a = granule_load(in_f0 + r0 + 4 * 8);
						g1b = granule_load(in_f1 + r0 + 4 * 8);
					}
					const bool giveup = ++spins > FLOW_SPIN_LIMIT;
					if (pub == 0 && (__all((unsigned)(g0a >> 32) == gen && (unsigned)(g0b >> 32) == gen) || giveup)) {
						if (m >= 2) {
							while (__hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < m - 1) {
								if (++spins > FLOW_SPIN_LIMIT) break;
								__builtin_amdgcn_s_sleep(ROWS_NAP);
							}
						}
						sF0[((m & 1) * 8 + pidx) * 8 + pr] = __uint_as_float((unsigned)g0a);
						sF1[((m & 1) * 8 + pidx) * 8 + pr] = __uint_as_float((unsigned)g0b);
						__hip_atomic_store(&s_flags[3], 2 * m + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						pub = 1;
					}
					if (pub == 1 && (__all((unsigned)(g1a >> 32) == gen && (unsigned)(g1b >> 32) == gen) || giveup)) {
						sF0[((m & 1) * 8 + pidx) * 8 + 4 + pr] = __uint_as_float((unsigned)g1a);
						sF1[((m & 1) * 8 + pidx) * 8 + 4 + pr] = __uint_as_float((unsigned)g1b);
						__hip_atomic_store(&s_flags[3], 2 * m + 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
						break;
					}
				}
			}
		} else if (wave != 0) {
			auto chunk_geom = [&](int m, int64_t& rowidx, int& nv) {
				const int x0 = m * 8;
				const int nvx = xlim - x0 < 8 ? xlim - x0 : 8;
				nv = row_in ? nvx : 0;
				rowidx = rowbase + x0;
			};
			auto wait_for = [&](int* flag, int need) {
				while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(ROWS_NAP);
				}
			};
			if (wave <= 3) {
				// ================= loaders =================
				const int w = wave - 1;
#pragma unroll 1
				for (int n = w; n < nchunks; n += 3) {
					int64_t rowidx;
					int nv;
					chunk_geom(n, rowidx, nv);
					int rF[8];
					float r0[8], ri[8], rj[8], rk[8];
					if (pack) {
						const unsigned long long W = nv > 0 ? *(const unsigned long long*)(pack + rowidx) : 0ull;     // sx % 8 == 0
#pragma unroll
						for (int a = 0; a < 8; a++) {
							const unsigned bits = (unsigned)(W >> (8 * a)) & 0xffu;
							rF[a] = (bits & 1u) ? MF_FLUID : 0;
							r0[a] = (float)(bits >> 4);
							ri[a] = (bits & 2u) ? -1.f : 0.f;
							rj[a] = (bits & 4u) ? -1.f : 0.f;
							rk[a] = (bits & 8u) ? -1.f : 0.f;
						}
					} else {
						load_row8i<VEC, REV>(flags, rowidx, nv, rF);
						load_row8<VEC, REV>(A0, rowidx, nv, r0);
						load_row8<VEC, REV>(Ai, rowidx, nv, ri);
						load_row8<VEC, REV>(Aj, rowidx, nv, rj);
						load_row8<VEC, REV>(Ak, rowidx, nv, rk);
					}
					if (n >= ROWS_RING / 8) wait_for(&s_flushed, n - ROWS_RING / 8 + 1);
					if (n >= ROWS_RING / 8) wait_for(&s_pub, 2 * (n - ROWS_RING / 8 + 2) + 2);
					const int p0 = 8 * n + skew + 2;
#pragma unroll
					for (int a = 0; a < 8; a++) {
						const bool fl = (a < nv) && (rF[a] & MF_FLUID);
						const int idx = ((p0 + a) & (ROWS_RING - 1)) * 64 + lane;
						sA[idx] = make_float4(fl ? r0[a] : 0.f, ri[a], rj[a], rk[a]);
						sB[idx] = fl ? 1.f : 0.f;
					}
					__hip_atomic_store(&s_ready[w], n + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			} else {
				// ================= write-back: Aprecond of finished chunks =================
#pragma unroll 1
				for (int q = 0; q < nchunks; q++) {
					wait_for(&s_done, q + 3);
					int64_t rowidx;
					int nv;
					chunk_geom(q, rowidx, nv);
					const int p0 = 8 * q + skew + 2;
					float w[8];
#pragma unroll
					for (int e = 0; e < 8; e++) w[e] = sA[((p0 + e) & (ROWS_RING - 1)) * 64 + lane].x;
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
			// ================= compute wave =================
			float oi0 = 0.f, oi1 = 0.f, oj0 = 0.f, oj1 = 0.f, ok0 = 0.f, ok1 = 0.f;
			float4 nA = sA[lane];
			float nB = sB[lane];
			typedef int rows_i4 __attribute__((ext_vector_type(4)));
			typedef float rows_f4 __attribute__((ext_vector_type(4)));
			typedef const volatile __attribute__((address_space(3))) rows_i4* lflags;
			typedef const volatile __attribute__((address_space(3))) rows_f4* lface;
			auto block = [&](int m, auto edge_tag) {
				constexpr bool EDGE = decltype(edge_tag)::value;
				const int xq = 8 * m - 2 - skew;
				rows_f4 f[8];      // {j h0, j h1, k h0, k h1} x {first, second half}
				const int r3 = m % 3;
				bool have2;
				for (;;) {
					const rows_i4 fg = *(lflags)s_flags;
#pragma unroll
					for (int v = 0; v < 2; v++) {
						f[v * 2 + 0] = *(lface)&sFj[v][m & 1][c][0];
						f[v * 2 + 1] = *(lface)&sFj[v][m & 1][c][4];
						f[4 + v * 2 + 0] = *(lface)&sFk[v][m & 1][b][0];
						f[4 + v * 2 + 1] = *(lface)&sFk[v][m & 1][b][4];
					}
					const int rdy = r3 == 0 ? fg.x : (r3 == 1 ? fg.y : fg.z);
					have2 = fg.w >= 2 * m + 2;
					if (((m >= nchunks) || rdy >= m + 1) && fg.w >= 2 * m + 1) break;
					if (++spins > FLOW_SPIN_LIMIT) break;
					__builtin_amdgcn_s_sleep(1);
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
				const int base = (8 * m) & (ROWS_RING - 1);
#pragma unroll
				for (int s = 0; s < 8; s++) {
					if (s == 4 && !have2) {
						for (;;) {
							const int fw = *(const volatile __attribute__((address_space(3))) int*)&s_flags[3];
#pragma unroll
							for (int v = 0; v < 2; v++) {
								f[v * 2 + 1] = *(lface)&sFj[v][m & 1][c][4];
								f[4 + v * 2 + 1] = *(lface)&sFk[v][m & 1][b][4];
							}
							if (fw >= 2 * m + 2 || ++spins > FLOW_SPIN_LIMIT) break;
							__builtin_amdgcn_s_sleep(1);
						}
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
					}
					const float4 cA = nA;
					const bool fl = nB != 0.f;
					const int row = (base + s) * 64 + lane;
					const int nrow = (s < 7) ? (base + s + 1) * 64 + lane : ((base + 8) & (ROWS_RING - 1)) * 64 + lane;
					nA = sA[nrow];
					nB = sB[nrow];
					const float dj0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oj0), 0x111, 0xf, 0xf, true));
					const float dj1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(oj1), 0x111, 0xf, 0xf, true));
					const float sk0 = __shfl_up(ok0, 8, 64), sk1 = __shfl_up(ok1, 8, 64);
					const int hs = s >> 2, es = s & 3;
					const float ij0 = (b == 0) ? f[0 + hs][es] : dj0, ij1 = (b == 0) ? f[2 + hs][es] : dj1;
					const float ik0 = (c == 0) ? f[4 + hs][es] : sk0, ik1 = (c == 0) ? f[6 + hs][es] : sk1;
					const float ii0 = oi0, ii1 = oi1;
					const bool valid = !EDGE || ((unsigned)(xq + s) < (unsigned)X8);
					const float a0 = cA.x, ai = cA.y, aj = cA.z, ak = cA.w;
					float ap = 0.f;
					if (fl) {
						float e = a0 - ii0 - ij0 - ik0;
						const float s3 = ii1 + ij1 + ik1;
						// e -= tau * ( ... + 0. ): fp64 product and subtraction, conjugategrad.cpp:84-88
						const float tau = 0.97f;
						e = (float)((double)e - (double)tau * ((double)s3 + 0.));
						if (e < 0.25f * a0) e = a0;
						ap = (float)(1. / (double)sqrtf(e));
					}
					const float ti_ = ai * ap, tj_ = aj * ap, tk_ = ak * ap;
					const float ap2 = ap * ap;
					oi0 = valid ? ti_ * ti_ : 0.f;
					oj0 = valid ? tj_ * tj_ : 0.f;
					ok0 = valid ? tk_ * tk_ : 0.f;
					oi1 = valid ? ai * (aj + ak) * ap2 : 0.f;
					oj1 = valid ? aj * (ai + ak) * ap2 : 0.f;
					ok1 = valid ? ak * (ai + aj) * ap2 : 0.f;
					if (valid) sA[row].x = ap;
					if (s == 3) {
						asm volatile("" ::: "memory");
						*(volatile __attribute__((address_space(3))) int*)&s_half = 2 * m + 1;
					}
				}
				asm volatile("" ::: "memory");
				*(volatile __attribute__((address_space(3))) int*)&s_done = m + 1;
				*(volatile __attribute__((address_space(3))) int*)&s_half = 2 * m + 2;
			};
#pragma unroll 1
			for (int m = 0; m <= nchunks + 1; m++) {
				const bool interior = (m >= 2 && m <= nchunks - 1);
				if (interior) block(m, std::false_type{});
				else block(m, std::true_type{});
			}
		}
		__syncthreads();
	}
	if (spins > FLOW_SPIN_LIMIT) atomicExch(&ctl->err, 1);
	if (threadIdx.x == 0) {
		const int f = atomicAdd(&ctl->finished, 1);
		if (f == (int)gridDim.x - 1) {
			xt[0] = 0;
			ctl->finished = 0;
		}
	}
}

// one byte per cell: bit 0 fluid, bits 1..3 "Ai / Aj / Ak is -1"; ok[0] is cleared when a coefficient is neither +0 nor -1
// (second-order boundaries, a caller's own matrix): the sweeps then read the four arrays themselves
__global__ void __launch_bounds__(BLOCK)
k_mic_pack(int64_t n, const int32_t* __restrict__ flags, const float* __restrict__ A0, const float* __restrict__ Ai,
           const float* __restrict__ Aj, const float* __restrict__ Ak, unsigned char* __restrict__ pack, int* __restrict__ ok) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= n) return;
	const unsigned ui = __float_as_uint(Ai[idx]), uj = __float_as_uint(Aj[idx]), uk = __float_as_uint(Ak[idx]);
	const unsigned M1 = 0xBF800000u;   // -1.0f
	const bool good = (ui == 0u || ui == M1) && (uj == 0u || uj == M1) && (uk == 0u || uk == M1);
	if (!good) ok[0] = 0;
	const bool fl = (flags[idx] & MF_FLUID) != 0;
	// bits 4-7: the diagonal A0 of a fluid cell when it is a small non-negative integer (MakeLaplaceMatrix counts the non-obstacle
	// neighbours: 0..6) -- ok[1] stays non-zero when that holds for every fluid cell, and ApplyMatrix then does not read A0 at all
	unsigned a0code = 0u;
	if (A0 == nullptr) {
		if (idx == 0) ok[1] = 0;
	} else if (fl) {
		const float v = A0[idx];
		const int iv = (int)v;
		if (__float_as_uint(v) == __float_as_uint((float)iv) && iv >= 0 && iv <= 15) a0code = (unsigned)iv;   // bit pattern: -0.0f is not the +0 the nibble rebuilds
		else ok[1] = 0;
	}
	pack[idx] = (unsigned char)((fl ? 1u : 0u) | (ui == M1 ? 2u : 0u) | (uj == M1 ? 4u : 0u) | (uk == M1 ? 8u : 0u) | (a0code << 4));
}

// bempty[tk * nbj + tj] = 1 when the 8x8 bundle of x-rows (tj, tk) needs no sweep: it has no fluid cell, and nothing couples
// into it -- the rows just below it in j and k carry Aj == 0 / Ak == 0 (what the backward substitution multiplies the bundle's
// pass-through values with; MakeLaplaceMatrix writes exactly that next to non-fluid cells).  In the forward substitution the
// bundle's faces are (val * A) * Aprecond with Aprecond == 0 in every non-fluid cell (mf_mic_init clears the grid and
// writes fluid cells only).  One workgroup per bundle.
__global__ void __launch_bounds__(BLOCK)
k_bundle_empty(Dim d, int nbj, const int32_t* __restrict__ flags, const float* __restrict__ Aj, const float* __restrict__ Ak,
               int* __restrict__ bempty) {
	const int tj = blockIdx.x % nbj, tk = blockIdx.x / nbj;
	const int j0 = tj * 8, k0 = tk * 8;
	__shared__ int s_live;
	if (threadIdx.x == 0) s_live = 0;
	__syncthreads();
	bool live = false;
	const int cells = 64 * d.sx;
	for (int q = threadIdx.x; q < cells && !live; q += BLOCK) {
		const int x = q % d.sx, r = q / d.sx, j = j0 + (r & 7), k = k0 + (r >> 3);
		if (j >= d.sy || k >= d.sz) continue;
		const int64_t idx = (int64_t)x + d.Y * j + d.Z * k;
		if (flags[idx] & MF_FLUID) live = true;
		if ((r & 7) == 0 && j0 > 0 && Aj[idx - d.Y] != 0.f) live = true;
		if ((r >> 3) == 0 && k0 > 0 && Ak[idx - d.Z] != 0.f) live = true;
	}
	if (live) s_live = 1;
	__syncthreads();
	if (threadIdx.x == 0) {
		bempty[blockIdx.x] = s_live ? 0 : 1;
		if (!s_live) atomicAdd(&bempty[gridDim.x], 1);      // number of empty bundles (slot behind the map)
	}
}

// dot(dst, var1) share of the bundles the backward sweep leaves out (dotpart[sid] of an empty bundle; the sweep writes the others): one
// workgroup per bundle and x-block, lanes along x (coalesced), rows in order, block_sum -- a fixed order, so the same bits on every
// run.  (In a pressure solve the share is exactly +0: the residual vanishes outside the fluid.)
__global__ void __launch_bounds__(BLOCK)
k_mic_empty_dot(Dim d, int nbj, int nbk, int X8, const int* __restrict__ bempty, const float* __restrict__ dst, const float* __restrict__ var1,
                const CgScalars* __restrict__ sc, double* __restrict__ dotpart) {
	if (sc && sc->done) return;
	const int sid = blockIdx.x;
	const int tjl = sid % nbj, tkl = (sid / nbj) % nbk, xb = sid / (nbj * nbk);
	const int tj = nbj - 1 - tjl, tk = nbk - 1 - tkl;          // the backward sweep's logical -> physical bundle
	if (!bempty[tk * nbj + tj]) return;
	const int xoff = xb * X8;
	const int xlim = d.sx - xoff < X8 ? d.sx - xoff : X8;
	double acc = 0.0;
	for (int r = threadIdx.x >> 6; r < 64; r += BLOCK / 64) {
		const int j = tj * 8 + (r & 7), k = tk * 8 + (r >> 3);
		if (j >= d.sy || k >= d.sz) continue;
		const int64_t rowbase = d.Y * j + d.Z * k + xoff;
		for (int x = threadIdx.x & 63; x < xlim; x += 64) acc += (double)(dst[rowbase + x] * var1[rowbase + x]);
	}
	acc = block_sum(acc);
	if (threadIdx.x == 0) dotpart[sid] = acc;
}

// The system handle of the sweeps (per device; built by mf_mic_init_blocked / mf_pack_matrix, used by the apply sweeps of the
// grids it was built for): sweep mode, preconditioner blocks, bundle order, hand-off buffers, packed operands, empty-bundle map.
struct FlowState {
	int mode = 2;                // sweep mode this system was initialised under: 2 "rows", 0 "levels" (mf_set_mic_mode / MF_MIC_MODE)
	int nbj = 0, nbk = 0, nblocks = 0, nchunks = 0;
	int* border = nullptr;       // ticket order of the forward sweep, then of the backward sweep (nblocks entries each)
	int jb = 0;                  // bundles per j-block the order was built for
	int nxb = 0;                 // x-blocks per row (1 = whole rows)
	int* rows_xt = nullptr;      // [2]: ticket counter of the forward / backward sweep
	// bundles without a fluid cell (and without coupling into them) need no sweep at all: built by mf_mic_init for the grids
	// it was given, used by the apply sweeps only when they are given the same grids
	int* bempty = nullptr;
	int bempty_cap = 0;
	int nempty_host = -1;        // number of empty bundles of the registered system as the host knows it (-1: not read back yet)
	const void *be_flags = nullptr, *be_Ap = nullptr, *be_Aj = nullptr, *be_Ak = nullptr;
	// preconditioner blocks of the system mf_mic_init_blocked was given (0 = uncut): the apply sweeps use them only when they
	// are called with the same flags / Aprecond / Aj / Ak (be_*), any other system is swept as the uncut reference algorithm
	int blk_rows = 0, blk_cells = 0;
	// packed operands of the apply sweeps (k_mic_pack), valid for the grids mf_mic_init was given
	unsigned char* pack = nullptr;
	int* pack_ok = nullptr;
	size_t pack_cap = 0;
	const void *pk_flags = nullptr, *pk_Ai = nullptr, *pk_Aj = nullptr, *pk_Ak = nullptr, *pk_A0 = nullptr;
	// a second set of packed bytes, built on request (mf_pack_matrix) for the plain mf_apply_matrix entry point
	unsigned char* upack = nullptr;
	int* upack_ok = nullptr;
	size_t upack_cap = 0;
	int upack_ok_host = 0;
	const void *up_flags = nullptr, *up_Ai = nullptr, *up_Aj = nullptr, *up_Ak = nullptr, *up_A0 = nullptr;
	unsigned long long *sxj = nullptr, *sxk = nullptr;
	unsigned long long *sxj1 = nullptr, *sxk1 = nullptr;      // second hand-off value of the init sweep (k_mic_rows_init)
	size_t sx_cap = 0;
	unsigned sgen = 0;
	FlowCtl* ctl = nullptr;
};
static FlowState g_flow[16];

static int rows_prepare(const Dim& d, FlowState** out, hipStream_t st, int jblock_rows, int xblock_cells) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	const int nbj = (d.sy + 7) / 8, nbk = (d.sz + 7) / 8;
	// x-blocks of xblock_cells cells (independent systems, the caller has cut Ai) -- or the whole row
	const int xcells = (xblock_cells > 0 && xblock_cells < d.sx) ? xblock_cells : ((d.sx + 7) / 8) * 8;
	const int nchunks = xcells / 8, nxb = (d.sx + xcells - 1) / xcells;
	if (nbj > 4095 || nbk > 4095 || nxb > 127) return fail("grid too large for the MIC bundle order table");
	int jb = jblock_rows > 0 ? jblock_rows / 8 : nbj;
	if (jb < 1 || jb > nbj) jb = nbj;
	if (!f.ctl) {
		MF_HIP(hipMalloc((void**)&f.ctl, sizeof(FlowCtl)));
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
	}
	if (f.nbj != nbj || f.nbk != nbk || f.nchunks != nchunks || f.jb != jb || f.nxb != nxb) {
		MF_HIP(hipStreamSynchronize(st));
		const int nb = nbj * nbk * nxb;
		int* h = (int*)malloc(sizeof(int) * 2 * nb);
		// tickets in topological order of each sweep: key = position of the bundle inside its j-block along the sweep
		// direction + tkl (anti-diagonals of the block-local dependency graph); ONE queue per sweep (see k_mic_rows)
		for (int rev = 0; rev < 2; rev++) {
			int q = 0;
			for (int L = 0; L <= nbj + nbk - 2; L++)
				for (int bk = 0; bk < nbk; bk++)
					for (int bjl = 0; bjl < nbj; bjl++) {
						const int tj = rev ? nbj - 1 - bjl : bjl;   // physical bundle row
						const int b0 = (tj / jb) * jb, b1 = (b0 + jb < nbj ? b0 + jb : nbj);
						const int posj = rev ? (b1 - 1 - tj) : (tj - b0);
						if (posj + bk == L)
							for (int xb = 0; xb < nxb; xb++) h[rev * nb + q++] = bjl | (bk << 12) | (xb << 24);
					}
		}
		if (!f.rows_xt) MF_HIP(hipMalloc((void**)&f.rows_xt, 2 * sizeof(int)));
		MF_HIP(hipMemset(f.rows_xt, 0, 2 * sizeof(int)));
		if (f.border) MF_HIP(hipFree(f.border));
		MF_HIP(hipMalloc((void**)&f.border, sizeof(int) * 2 * nb));
		MF_HIP(hipMemcpy(f.border, h, sizeof(int) * 2 * nb, hipMemcpyHostToDevice));
		free(h);
		// one granule per (bundle, x', face lane) and face
		const size_t need = (size_t)nb * 8 * (8 * (size_t)nchunks + 2 * ROWS_PAD) * sizeof(unsigned long long);
		if (need > f.sx_cap) {
			// (fine-grained memory, hipExtMallocWithFlags, was tried for these: 0.45 instead of 0.71 us per idle hand-off in
			// tools/micro/pingpong_scalar.hip, but no change of the sweep time -- 575.0 vs 576.4 us per apply at 256^3; so were four
			// scalar-path poller waves (s_load_dwordx16 glc, 0.45 us per hand-off in the micro-benchmark): bit-exact, 700+ us)
			for (unsigned long long** q : {&f.sxj, &f.sxk, &f.sxj1, &f.sxk1}) {
				if (*q) MF_HIP(hipFree(*q));
				MF_HIP(hipMalloc((void**)q, need));
			}
			f.sx_cap = need;
		}
		for (unsigned long long* q : {f.sxj, f.sxk, f.sxj1, f.sxk1}) MF_HIP(hipMemset(q, 0, f.sx_cap));
		MF_HIP(hipMemset(f.ctl, 0, sizeof(FlowCtl)));
		f.sgen = 0;
		f.nbj = nbj;
		f.nbk = nbk;
		f.nblocks = nb;
		f.nchunks = nchunks;
		f.jb = jb;
		f.nxb = nxb;
	}
	*out = &f;
	return 0;
}
// next launch generation of the hand-off granules (tag = generation: the buffers are cleared only when the 32-bit counter wraps)
static int rows_next_gen(FlowState* f, hipStream_t st) {
	f->sgen++;
	if (f->sgen == 0) {
		for (unsigned long long* q : {f->sxj, f->sxk, f->sxj1, f->sxk1}) MF_HIP(hipMemsetAsync(q, 0, f->sx_cap, st));
		f->sgen = 1;
	}
	return 0;
}

// How the sweeps are parallelised (both give the bits of the serial sweep):
// 2 "rows"  : one launch per sweep, a 7-wave workgroup per 8x8 bundle of x-rows streaming along x (default for 3D grids)
// 0 "levels": one launch per tile hyperplane (no inter-workgroup waiting at all; the conservative fallback, 2.2 ms per 256^3 apply)
// mf_set_mic_mode / MF_MIC_MODE choose the mode the NEXT mf_mic_init registers its system under; the apply sweeps of that system
// follow the handle, a system the handle does not know is swept in the requested mode.
// (Round 2 also carried "tiles", single-wave 8^3 tiles in one launch -- 1.32 ms -- and "rows-sb", 2 x 2 bundles per workgroup --
// 1.05 ms; both bit-exact, both slower than "rows", deleted in round 3.)
static int g_mic_mode = -1;     // requested mode: 0 levels, 2 rows
extern "C" int mf_set_mic_mode(const char* name) {
	if (!name || !*name) g_mic_mode = -1;
	else if (!strcmp(name, "levels")) g_mic_mode = 0;
	else if (!strcmp(name, "rows")) g_mic_mode = 2;
	else return fail("mf_set_mic_mode: unknown mode (rows | levels)");
	return 0;
}
static int mic_mode_() {
	if (g_mic_mode < 0) {
		const char* e = getenv("MF_MIC_MODE");
		g_mic_mode = (e && !strcmp(e, "levels")) ? 0 : 2;
	}
	return g_mic_mode;
}
extern "C" int mf_pack_matrix(int sx, int sy, int sz, const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	hipStream_t st = (hipStream_t)stream;
	f.upack_ok_host = 0;
	if (!d.is3d || (d.sx % 4) != 0) return 0;
	if ((size_t)d.n > f.upack_cap) {
		MF_HIP(hipStreamSynchronize(st));
		if (f.upack) MF_HIP(hipFree(f.upack));
		MF_HIP(hipMalloc((void**)&f.upack, (size_t)d.n + 64));
		f.upack_cap = (size_t)d.n;
	}
	if (!f.upack_ok) MF_HIP(hipMalloc((void**)&f.upack_ok, 2 * sizeof(int)));
	MF_HIP(hipMemsetAsync(f.upack_ok, 1, 2 * sizeof(int), st));
	hipLaunchKernelGGL(k_mic_pack, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d.n, flags, A0, Ai, Aj, Ak, f.upack, f.upack_ok);
	MF_LAUNCH_CHECK();
	int ok[2] = {0, 0};
	MF_HIP(hipMemcpyAsync(ok, f.upack_ok, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	f.upack_ok_host = ok[0] != 0;
	f.up_A0 = (ok[0] && ok[1] && A0) ? A0 : nullptr;      // the diagonal this set of bytes carries in bits 4-7, if any
	f.up_flags = flags;
	f.up_Ai = Ai;
	f.up_Aj = Aj;
	f.up_Ak = Ak;
	return 0;
}

// set by mic_launch_dot for the duration of one backward-sweep launch
static thread_local double* g_dot_request = nullptr;
static thread_local int g_dot_count = 0;
static thread_local bool g_dot_empty_ext = false;
static thread_local BetaTail g_dot_tail = BetaTail{nullptr, 0, nullptr, nullptr, 0};
static thread_local bool g_dot_tail_done = false;
// x-range the caller asks the sweeps of the registered system to keep to (mf::mic_set_trim): first cell, chunks of 8 cells; 0 = whole rows
static thread_local int g_trim_xoff = 0, g_trim_chunks = 0;
template <int MODE>
static int launch_mic(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                      const float* Aj, const float* Ak, const CgScalars* sc, hipStream_t st) {
	const int nti = (d.sx + 7) / 8, ntj = (d.sy + 7) / 8, ntk = (d.sz + 7) / 8;
	const int levels = nti + ntj + ntk - 2;
	const bool vec = (d.sx % 4 == 0) && al16(flags) && al16(dst) && al16(var1) && al16(Ai) && al16(Aj) && al16(Ak) && (MODE == 0 || al16(Ap));
	if constexpr (MODE != 0) {
		int dev_ = 0;
		MF_HIP(hipGetDevice(&dev_));
		const FlowState& f0 = g_flow[dev_];
		// the system mf_mic_init registered: its mode and its preconditioner blocks; any other system: requested mode, uncut
		const bool same_system = f0.be_flags == flags && f0.be_Ap == Ap && f0.be_Aj == Aj && f0.be_Ak == Ak;
		const int mode = same_system ? f0.mode : mic_mode_();
		if (mode == 2 && d.is3d) {
			FlowState* f;
			MF_TRY(rows_prepare(d, &f, st, same_system ? f0.blk_rows : 0, same_system ? f0.blk_cells : 0));
			MF_TRY(rows_next_gen(f, st));
			static int ncu = 0;
			if (!ncu) {
				// one bundle per CU measured best (round 1: 834 us per apply with 256 workgroups, 1040 us with 512; round 2, with a
				// packed-only <= 128-VGPR variant and a 32- / 48-step ring for two workgroups per CU: 445 vs 412 us, 217 vs 204 us per
				// sweep -- mid-sweep the 256 bundles already stream ~4 TB/s, the rest of the sweep is the dependency chain)
				int dev = 0;
				ncu = 256;
				(void)hipGetDevice(&dev);
				(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
				if (ncu < 1) ncu = 1;
			}
			const int grid = f->nblocks < ncu ? f->nblocks : ncu;
			const int* be = (f->bempty && same_system) ? f->bempty : nullptr;
			static const bool nopack = getenv("MF_MIC_NOPACK") != nullptr;       // debugging: sweeps on the four coefficient arrays
			const bool use_pack = !nopack && f->pack && (d.sx % 8 == 0) && f->pk_flags == flags && f->pk_Ai == Ai && f->pk_Aj == Aj && f->pk_Ak == Ak;
			const unsigned char* pk = use_pack ? f->pack : nullptr;
			double* dotp = (MODE == 2 && al16(var1) && f->nblocks <= MAX_BLOCKS) ? g_dot_request : nullptr;
			g_dot_count = dotp ? f->nblocks : 0;
			const int empty_ext = (MODE == 2 && dotp && be && g_dot_empty_ext) ? 1 : 0;
			if (MODE == 2 && dotp && be && f->nblocks > grid && !empty_ext) {
				if (f->nempty_host < 0) {
					// once per system: does any bundle sit out the sweeps?  (smoke scenes: none -- no extra launch per iteration)
					static thread_local std::vector<int> hb;
					hb.resize((size_t)f->nbj * f->nbk);
					MF_HIP(hipMemcpyAsync(hb.data(), f->bempty, sizeof(int) * hb.size(), hipMemcpyDeviceToHost, st));
					MF_HIP(hipStreamSynchronize(st));
					int cnt = 0;
					for (int v : hb) cnt += v != 0;
					f->nempty_host = cnt;
				}
				if (f->nempty_host > 0 && f->nblocks > grid)
					hipLaunchKernelGGL(k_mic_empty_dot, dim3(f->nblocks), dim3(BLOCK), 0, st, d, f->nbj, f->nbk, f->nchunks * 8, be, dst, var1, sc, dotp);
			}
			// rows trimmed to the x-range of the fluid (mic_set_trim; only for the registered system on its packed bytes)
			int nch = f->nchunks, xoff0 = 0;
			if (same_system && use_pack && g_trim_chunks > 0 && g_trim_chunks < f->nchunks && f->nxb == 1) {
				nch = g_trim_chunks;
				xoff0 = g_trim_xoff;
			}
			BetaTail ktail = BetaTail{nullptr, 0, nullptr, nullptr, 0};
			if (MODE == 2 && dotp && (g_dot_tail.sc || g_dot_tail.sum_out) && !(be && f->nblocks > grid && !empty_ext && f->nempty_host > 0)) {
				// (not with the separate empty-share kernel: its partials are plain stores of another launch -- fine -- but keep it simple)
				ktail = g_dot_tail;
				ktail.nsig += f->nblocks;      // the sweep's own partials come first
				g_dot_tail_done = true;
			}
			if (vec)
				hipLaunchKernelGGL((k_mic_rows<MODE, true>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, nch, xoff0, f->border + (MODE == 2 ? f->nblocks : 0), f->ctl, f->rows_xt + (MODE == 2 ? 1 : 0), f->sxj, f->sxk, f->sgen, flags, dst, var1, Ap, Ai, Aj, Ak, sc, dotp, be, pk, f->pack_ok, empty_ext, ktail);
			else
				hipLaunchKernelGGL((k_mic_rows<MODE, false>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, nch, xoff0, f->border + (MODE == 2 ? f->nblocks : 0), f->ctl, f->rows_xt + (MODE == 2 ? 1 : 0), f->sxj, f->sxk, f->sgen, flags, dst, var1, Ap, Ai, Aj, Ak, sc, dotp, be, pk, f->pack_ok, empty_ext, ktail);
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


// one-block sum of the per-bundle dot partials (index order) -- or, without fusion, a plain dot over the grid
static __global__ void __launch_bounds__(BLOCK) k_mic_fin_sum(int nb, const double* __restrict__ partials, double* __restrict__ out) {
	double acc = strided_sum(partials, nb);
	acc = block_sum(acc);
	if (threadIdx.x == 0) out[0] = acc;
}
static __global__ void __launch_bounds__(BLOCK) k_mic_plain_dot(int64_t n, const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partials) {
	double acc = 0.0;
	for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) acc += (double)(a[i] * b[i]);
	acc = block_sum(acc);
	if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
static __global__ void __launch_bounds__(BLOCK) k_mic_fin_maxabs_live(int nb, const float* __restrict__ fpart, double* __restrict__ out, const CgScalars* __restrict__ live) {
	if (live && live->done) return;
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

namespace mf {
void mic_set_trim(int xoff_cells, int nchunks) {
	g_trim_xoff = xoff_cells;
	g_trim_chunks = nchunks;
}
int mic_launch(int mode, const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
               const float* Aj, const float* Ak, const CgScalars* sc, hipStream_t st) {
	if (mode == 0) return launch_mic<0>(d, flags, dst, var1, Ap, Ai, Aj, Ak, sc, st);
	if (mode == 1) return launch_mic<1>(d, flags, dst, var1, Ap, Ai, Aj, Ak, sc, st);
	return launch_mic<2>(d, flags, dst, var1, Ap, Ai, Aj, Ak, sc, st);
}
// backward sweep with GridDotProduct(dst, var1) fused: *ndot = number of partials written to dotpart (0: not fused in
// this mode -- the caller runs its own dot kernel)
int mic_launch_dot(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                   const float* Aj, const float* Ak, const CgScalars* sc, double* dotpart, int* ndot, hipStream_t st, bool empty_ext,
                   BetaTail tail, bool* tail_done) {
	g_dot_request = dotpart;
	g_dot_empty_ext = empty_ext;
	g_dot_tail = tail;
	g_dot_tail_done = false;
	g_dot_count = 0;
	const int rc = launch_mic<2>(d, flags, dst, var1, Ap, Ai, Aj, Ak, sc, st);
	g_dot_request = nullptr;
	g_dot_empty_ext = false;
	g_dot_tail = BetaTail{nullptr, 0, nullptr, nullptr, 0};
	if (tail_done) *tail_done = g_dot_tail_done;
	*ndot = g_dot_count;
	return rc;
}
int mic_apply_dot_fold(const Dim& d, const int32_t* flags, float* dst, const float* var1, const float* Ap, const float* Ai,
                       const float* Aj, const float* Ak, double* dot_dev, int nbr, const float* fpart, double* maxabs_dev,
                       const CgScalars* live, hipStream_t st) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	double* part = ws->partials + 2 * MAX_BLOCKS;
	// live->done (the z-slab solver's iterations queued past the stop): both sweeps return at once, *dot_dev / *maxabs_dev keep the
	// values of the stopping iteration (dst and var1 have not changed since)
	MF_TRY(mic_launch(1, d, flags, dst, var1, Ap, Ai, Aj, Ak, live, st));
	int nsig = 0;
	bool folded = false;
	BetaTail tail = BetaTail{nullptr, nbr, fpart, nullptr, 0, dot_dev, maxabs_dev, live};
	MF_TRY(mic_launch_dot(d, flags, dst, var1, Ap, Ai, Aj, Ak, live, part, &nsig, st, false, tail, &folded));
	if (folded) return 0;
	if (nsig == 0) {
		nsig = blocks_for(d.n, BLOCK * 4, 2048);
		hipLaunchKernelGGL(k_mic_plain_dot, dim3(nsig), dim3(BLOCK), 0, st, d.n, dst, var1, part);
	}
	hipLaunchKernelGGL(k_mic_fin_sum, dim3(1), dim3(BLOCK), 0, st, nsig, part, dot_dev);
	if (nbr > 0) hipLaunchKernelGGL(k_mic_fin_maxabs_live, dim3(1), dim3(BLOCK), 0, st, nbr, fpart, maxabs_dev, live);
	MF_LAUNCH_CHECK();
	return 0;
}
// the packed flags/Ai/Aj/Ak bytes mf_mic_init built for exactly these grids, if every coefficient was +0 or -1 (one small
// device-to-host read, i.e. a stream synchronisation: call it once per solve); *pack = nullptr otherwise
int mic_pack_query(const Dim& d, const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                   const unsigned char** pack, bool* a0_packed, hipStream_t st) {
	*pack = nullptr;
	*a0_packed = false;
	static const bool nopack = getenv("MF_MIC_NOPACK") != nullptr;
	if (nopack || !d.is3d) return 0;
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	if (f.mode != 2 || !f.pack || !f.pack_ok || f.pk_flags != flags || f.pk_Ai != Ai || f.pk_Aj != Aj || f.pk_Ak != Ak) return 0;
	int ok[2] = {0, 0};
	MF_HIP(hipMemcpyAsync(ok, f.pack_ok, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
	MF_HIP(hipStreamSynchronize(st));
	if (ok[0]) *pack = f.pack;
	*a0_packed = ok[0] && ok[1] && f.pk_A0 == A0 && A0 != nullptr;
	return 0;
}
// the empty-bundle map of the system registered for (flags, Ap, Aj, Ak) in "rows" mode, if that system has empty bundles and its sweeps
// draw several tickets per workgroup (then the shares of those bundles in the fused dot are worth summing elsewhere): one small read-back
// per system.  *bempty = nullptr otherwise.
int mic_empty_map(const Dim& d, const int32_t* flags, const float* Ap, const float* Aj, const float* Ak, const int** bempty, int* nbj, hipStream_t st,
                  bool any_size) {
	*bempty = nullptr;
	*nbj = 0;
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState& f = g_flow[dev];
	if (!d.is3d || f.mode != 2 || !f.bempty || f.be_flags != flags || f.be_Ap != Ap || f.be_Aj != Aj || f.be_Ak != Ak) return 0;
	if (f.nbj != (d.sy + 7) / 8 || f.nbk != (d.sz + 7) / 8 || f.nxb != 1) return 0;
	int ncu = 256;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	if (f.nblocks <= ncu && !any_size) return 0;
	if (f.nempty_host < 0) {
		static thread_local std::vector<int> hb;
		hb.resize((size_t)f.nbj * f.nbk);
		MF_HIP(hipMemcpyAsync(hb.data(), f.bempty, sizeof(int) * hb.size(), hipMemcpyDeviceToHost, st));
		MF_HIP(hipStreamSynchronize(st));
		int cnt = 0;
		for (int v : hb) cnt += v != 0;
		f.nempty_host = cnt;
	}
	if (f.nempty_host > 0) {
		*bempty = f.bempty;
		*nbj = f.nbj;
	}
	return 0;
}
// packed bytes built by mf_pack_matrix for exactly these grids (no synchronisation: the verdict was read when they were built)
const unsigned char* mic_pack_user(const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak, bool* a0_packed) {
	*a0_packed = false;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) return nullptr;
	FlowState& f = g_flow[dev];
	if (!f.upack || !f.upack_ok_host || f.up_flags != flags || f.up_Ai != Ai || f.up_Aj != Aj || f.up_Ak != Ak) return nullptr;
	*a0_packed = f.up_A0 != nullptr && f.up_A0 == A0;
	return f.upack;
}
int mic_flow_error() {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	if (!g_flow[dev].ctl) return 0;
	FlowCtl fc;
	MF_HIP(hipMemcpy(&fc, g_flow[dev].ctl, sizeof fc, hipMemcpyDeviceToHost));
	if (fc.err) {
		MF_HIP(hipMemset(g_flow[dev].ctl, 0, sizeof(FlowCtl)));
		return fail("MIC dataflow sweep: a workgroup timed out waiting for its predecessor faces");
	}
	return 0;
}
int mic_mode() { return mic_mode_(); }      // the requested mode (0 levels, 2 rows)
// matrix-free set-up (mf_solve_pressure_fused): the handle's packed-byte and empty-bundle buffers for a d-sized system, the map
// preset to "empty" (the set-up kernel clears the entry of every bundle it finds a fluid cell in)
int mic_fused_begin(const Dim& d, hipStream_t st, unsigned char** pack, int** bempty, int* nbj) {
	if (!d.is3d || (d.sx % 8) != 0) return fail("mic_fused_begin: needs a 3D grid with sx % 8 == 0");
	FlowState* f;
	MF_TRY(rows_prepare(d, &f, st, 0, 0));
	if (f->nblocks + 1 > f->bempty_cap) {
		MF_HIP(hipStreamSynchronize(st));
		if (f->bempty) MF_HIP(hipFree(f->bempty));
		MF_HIP(hipMalloc((void**)&f->bempty, sizeof(int) * (f->nblocks + 1)));
		f->bempty_cap = f->nblocks + 1;
	}
	if ((size_t)d.n > f->pack_cap) {
		MF_HIP(hipStreamSynchronize(st));
		if (f->pack) MF_HIP(hipFree(f->pack));
		MF_HIP(hipMalloc((void**)&f->pack, (size_t)d.n + 64));
		f->pack_cap = (size_t)d.n;
	}
	if (!f->pack_ok) MF_HIP(hipMalloc((void**)&f->pack_ok, 2 * sizeof(int)));
	MF_HIP(hipMemsetD32Async((hipDeviceptr_t)f->bempty, 1, f->nblocks + 1, st));
	MF_HIP(hipMemsetAsync(f->pack_ok, 1, 2 * sizeof(int), st));      // both verdicts hold by construction
	f->be_flags = nullptr;
	*pack = f->pack;
	*bempty = f->bempty;
	*nbj = f->nbj;
	return 0;
}
// ... and the MIC factor from those bytes (k_mic_rows_init), registered as the system of (flags, Aprecond) with no coefficient arrays
int mic_fused_finish(const Dim& d, const int32_t* flags, float* Aprecond, hipStream_t st) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	FlowState* f;
	MF_TRY(rows_prepare(d, &f, st, 0, 0));
	MF_HIP(hipMemsetAsync(Aprecond, 0, sizeof(float) * d.n, st));
	MF_TRY(rows_next_gen(f, st));
	int ncu = 256;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	const int grid = f->nblocks < ncu ? f->nblocks : ncu;
	if (!al16(Aprecond)) return fail("mic_fused_finish: Aprecond must be 16-byte aligned");
	hipLaunchKernelGGL((k_mic_rows_init<true>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, f->nchunks, f->border, f->ctl, f->rows_xt, f->sxj, f->sxk, f->sxj1, f->sxk1, f->sgen, flags, Aprecond, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const unsigned char*)f->pack);
	MF_LAUNCH_CHECK();
	f->mode = 2;
	f->blk_rows = f->blk_cells = 0;
	f->be_flags = flags;
	f->be_Ap = Aprecond;
	f->be_Aj = f->be_Ak = nullptr;
	f->nempty_host = -1;
	f->pk_flags = flags;
	f->pk_A0 = f->pk_Ai = f->pk_Aj = f->pk_Ak = nullptr;
	return 0;
}
}  // namespace mf

extern "C" {

int mf_mic_init(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0, const float* Ai,
                const float* Aj, const float* Ak, void* stream) {
	return mf_mic_init_blocked(sx, sy, sz, flags, Aprecond, A0, Ai, Aj, Ak, 0, 0, stream);
}
int mf_mic_init_blocked(int sx, int sy, int sz, const int32_t* flags, float* Aprecond, const float* A0, const float* Ai,
                        const float* Aj, const float* Ak, int rows_j, int cells_x, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (rows_j < 0 || (rows_j % 8) != 0) return fail("mf_mic_init_blocked: rows_j must be a non-negative multiple of 8");
	if (cells_x < 0 || (cells_x % 8) != 0) return fail("mf_mic_init_blocked: cells_x must be a non-negative multiple of 8");
	const Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(Aprecond, 0, sizeof(float) * d.n, st));
	const int mode = mic_mode_();
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	g_flow[dev].mode = mode;
	g_flow[dev].be_flags = nullptr;        // no registered system until this call has built one
	if (mode != 2) return mic_launch(0, d, flags, Aprecond, A0, nullptr, Ai, Aj, Ak, nullptr, st);   // one launch per tile hyperplane
	// "rows": one dataflow sweep (k_mic_rows_init)
	FlowState* f;
	MF_TRY(rows_prepare(d, &f, st, rows_j, cells_x));
	MF_TRY(rows_next_gen(f, st));
	int ncu = 256;
	(void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
	const int grid = f->nblocks < ncu ? f->nblocks : ncu;
	const bool vec = (d.sx % 4 == 0) && al16(flags) && al16(Aprecond) && al16(A0) && al16(Ai) && al16(Aj) && al16(Ak);
	if (vec)
		hipLaunchKernelGGL((k_mic_rows_init<true>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, f->nchunks, f->border, f->ctl, f->rows_xt, f->sxj, f->sxk, f->sxj1, f->sxk1, f->sgen, flags, Aprecond, A0, Ai, Aj, Ak, (const unsigned char*)nullptr);
	else
		hipLaunchKernelGGL((k_mic_rows_init<false>), dim3(grid), dim3(ROWS_THREADS), 0, st, d, f->nbj, f->nbk, f->jb, f->nblocks, f->nchunks, f->border, f->ctl, f->rows_xt, f->sxj, f->sxk, f->sxj1, f->sxk1, f->sgen, flags, Aprecond, A0, Ai, Aj, Ak, (const unsigned char*)nullptr);
	MF_LAUNCH_CHECK();
	// which row bundles the apply sweeps of THIS system may leave out (valid for the grids given here)
	f->blk_rows = rows_j;
	f->blk_cells = cells_x;
	if (f->nblocks + 1 > f->bempty_cap) {
		MF_HIP(hipStreamSynchronize(st));
		if (f->bempty) MF_HIP(hipFree(f->bempty));
		MF_HIP(hipMalloc((void**)&f->bempty, sizeof(int) * (f->nblocks + 1)));
		f->bempty_cap = f->nblocks + 1;
	}
	MF_HIP(hipMemsetAsync(f->bempty + f->nbj * f->nbk, 0, sizeof(int), st));
	hipLaunchKernelGGL(k_bundle_empty, dim3(f->nbj * f->nbk), dim3(BLOCK), 0, st, d, f->nbj, flags, Aj, Ak, f->bempty);
	MF_LAUNCH_CHECK();
	f->be_flags = flags;
	f->be_Ap = Aprecond;
	f->be_Aj = Aj;
	f->be_Ak = Ak;
	f->nempty_host = -1;
	// packed operands for the apply sweeps
	if ((size_t)d.n > f->pack_cap) {
		MF_HIP(hipStreamSynchronize(st));
		if (f->pack) MF_HIP(hipFree(f->pack));
		MF_HIP(hipMalloc((void**)&f->pack, (size_t)d.n + 64));
		f->pack_cap = (size_t)d.n;
	}
	if (!f->pack_ok) MF_HIP(hipMalloc((void**)&f->pack_ok, 2 * sizeof(int)));
	MF_HIP(hipMemsetAsync(f->pack_ok, 1, 2 * sizeof(int), st));     // non-zero = valid until k_mic_pack clears it
	hipLaunchKernelGGL(k_mic_pack, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d.n, flags, A0, Ai, Aj, Ak, f->pack, f->pack_ok);
	MF_LAUNCH_CHECK();
	f->pk_A0 = A0;
	f->pk_flags = flags;
	f->pk_Ai = Ai;
	f->pk_Aj = Aj;
	f->pk_Ak = Ak;
	return 0;
}
int mf_mic_apply(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1, const float* Aprecond,
                 const float* Ai, const float* Aj, const float* Ak, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	MF_TRY(mic_launch(1, d, flags, dst, var1, Aprecond, Ai, Aj, Ak, nullptr, (hipStream_t)stream));
	return mic_launch(2, d, flags, dst, var1, Aprecond, Ai, Aj, Ak, nullptr, (hipStream_t)stream);
}

int mf_mic_apply_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1, const float* Aprecond,
                         const float* Ai, const float* Aj, const float* Ak, double* dot_dev, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	return mf::mic_apply_dot_fold(d, flags, dst, var1, Aprecond, Ai, Aj, Ak, dot_dev, 0, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

// a dataflow sweep that gives up waiting for a face (FLOW_SPIN_LIMIT) latches an error flag on the device; mf_cg_solve
// looks at it itself, callers that drive mf_mic_apply directly (the z-slab solver) ask here once per solve
int mf_mic_check(void* stream) {
	MF_HIP(hipStreamSynchronize((hipStream_t)stream));
	return mic_flow_error();
}

}  // extern "C"
