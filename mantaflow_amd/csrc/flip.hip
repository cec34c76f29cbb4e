// flip.hip -- FLIP particle <-> grid transfers and particle advection (gfx950).
// Reference: source/plugin/flip.cpp:607-742, source/particle.h:458-550, source/util/integrator.h:26-78,
// source/util/interpol.h:96-213.  Particle positions / velocities are SoA (x[], y[], z[] with stride pstride).
#include "common.h"
#include <float.h>
#include <stdlib.h>

using namespace mf;

static inline unsigned nblk_n(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK > 0 ? (n + BLOCK - 1) / BLOCK : 1); }

__device__ __forceinline__ bool skip_particle(const int32_t* __restrict__ pflag, const int32_t* __restrict__ ptype, int exclude, int64_t p) {
	// !p.isActive(idx) || (ptype && ((*ptype)[idx] & exclude)), flip.cpp:630,712,727
	return (pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude));
}

// ---------------------------------------------------------------------------------------------------------
// particle -> grid scatter.  setInterpolMAC, interpol.h:166-213: per component c the 8 trilinear weights are
// products t*(s*f) formed in this order; value contribution = w * val.
// ATOMIC = true : fp32 atomicAdd (global_atomic_add_f32), order of summation is not defined
// ATOMIC = false: plain read-modify-write; only legal from the single sequential wave of the deterministic path
// ---------------------------------------------------------------------------------------------------------
template <bool ATOMIC>
__device__ __forceinline__ void add_to(float* p, float v) {
	if (ATOMIC)
		atomicAdd(p, v);
	else
		*p += v;
}
template <bool ATOMIC>
__device__ __forceinline__ void scatter8(float* __restrict__ ref, float* __restrict__ sum, int64_t Y, int64_t Z, float ta,
                                         float tb, float sa, float sb, float fa, float fb, float val, bool zfirst) {
	const float s0f0 = sa * fa, s1f0 = sb * fa, s0f1 = sa * fb, s1f1 = sb * fb;
	const float w0 = ta * s0f0, wx = ta * s1f0, wy = tb * s0f0, wxy = tb * s1f0;
	const float wz = ta * s0f1, wxz = ta * s1f1, wyz = tb * s0f1, wxyz = tb * s1f1;
	if (zfirst) {
		add_to<ATOMIC>(sum + Z, wz); add_to<ATOMIC>(sum + 1 + Z, wxz); add_to<ATOMIC>(sum + Y + Z, wyz); add_to<ATOMIC>(sum + 1 + Y + Z, wxyz);
		add_to<ATOMIC>(ref + Z, wz * val); add_to<ATOMIC>(ref + 1 + Z, wxz * val); add_to<ATOMIC>(ref + Y + Z, wyz * val); add_to<ATOMIC>(ref + 1 + Y + Z, wxyz * val);
		add_to<ATOMIC>(sum, w0); add_to<ATOMIC>(sum + 1, wx); add_to<ATOMIC>(sum + Y, wy); add_to<ATOMIC>(sum + 1 + Y, wxy);
		add_to<ATOMIC>(ref, w0 * val); add_to<ATOMIC>(ref + 1, wx * val); add_to<ATOMIC>(ref + Y, wy * val); add_to<ATOMIC>(ref + 1 + Y, wxy * val);
	} else {
		add_to<ATOMIC>(sum, w0); add_to<ATOMIC>(sum + 1, wx); add_to<ATOMIC>(sum + Y, wy); add_to<ATOMIC>(sum + 1 + Y, wxy);
		add_to<ATOMIC>(sum + Z, wz); add_to<ATOMIC>(sum + 1 + Z, wxz); add_to<ATOMIC>(sum + Y + Z, wyz); add_to<ATOMIC>(sum + 1 + Y + Z, wxyz);
		add_to<ATOMIC>(ref, w0 * val); add_to<ATOMIC>(ref + 1, wx * val); add_to<ATOMIC>(ref + Y, wy * val); add_to<ATOMIC>(ref + 1 + Y, wxy * val);
		add_to<ATOMIC>(ref + Z, wz * val); add_to<ATOMIC>(ref + 1 + Z, wxz * val); add_to<ATOMIC>(ref + Y + Z, wyz * val); add_to<ATOMIC>(ref + 1 + Y + Z, wxyz * val);
	}
}
template <bool ATOMIC>
__device__ __forceinline__ void p2g_mac_one(const Dim& d, float* __restrict__ vel, float* __restrict__ weight, float x, float y,
                                            float z, float ux, float uy, float uz) {
	const Bi b = build_index(d, x, y, z), s = build_index_shift(d, x, y, z);
	const int64_t n = d.n;
	const int64_t ix = ((int64_t)b.zi * d.sy + b.yi) * d.sx + s.xi;
	scatter8<ATOMIC>(vel + ix, weight + ix, d.Y, d.Z, b.t0, b.t1, s.s0, s.s1, b.f0, b.f1, ux, true);
	const int64_t iy = ((int64_t)b.zi * d.sy + s.yi) * d.sx + b.xi;
	scatter8<ATOMIC>(vel + n + iy, weight + n + iy, d.Y, d.Z, s.t0, s.t1, b.s0, b.s1, b.f0, b.f1, uy, true);
	const int64_t iz = ((int64_t)s.zi * d.sy + b.yi) * d.sx + b.xi;
	scatter8<ATOMIC>(vel + 2 * n + iz, weight + 2 * n + iz, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, s.f0, s.f1, uz, false);
}
// knMapLinearVec3ToMACGrid, flip.cpp:619-633 -- one thread per particle, fp32 atomics.
// Particles that follow each other in memory mostly sit in the same cell (they are sampled cell by cell and stay
// spatially coherent), i.e. neighbouring lanes hit the same 8 corners: before the atomics, runs of equal base index
// inside aligned groups of 8 lanes are summed with three DPP row-shift steps, so a cell-coherent wave issues up to 8x
// fewer atomics (global_atomic_add_f32 runs at ~1.3 TB/s of added bytes chip-wide, MI355X_MICROARCH.md).
__device__ __forceinline__ float dpp_shl(float v, int s) {
	const int i = __float_as_int(v);
	int r;
	if (s == 1) r = __builtin_amdgcn_update_dpp(0, i, 0x101, 0xf, 0xf, true);
	else if (s == 2) r = __builtin_amdgcn_update_dpp(0, i, 0x102, 0xf, 0xf, true);
	else r = __builtin_amdgcn_update_dpp(0, i, 0x104, 0xf, 0xf, true);
	return __int_as_float(r);
}
__device__ __forceinline__ int dpp_shl_i(int v, int s) {
	if (s == 1) return __builtin_amdgcn_update_dpp(-2, v, 0x101, 0xf, 0xf, false);
	if (s == 2) return __builtin_amdgcn_update_dpp(-2, v, 0x102, 0xf, 0xf, false);
	return __builtin_amdgcn_update_dpp(-2, v, 0x104, 0xf, 0xf, false);
}
// one component: 8 weights + 8 weighted values, keyed by the base cell index `key` (-1 = inactive particle)
__device__ __forceinline__ void p2g_component_grouped(float* __restrict__ ref, float* __restrict__ sum, int key, int64_t Y, int64_t Z,
                                                      float ta, float tb, float sa, float sb, float fa, float fb, float val) {
	const float s0f0 = sa * fa, s1f0 = sb * fa, s0f1 = sa * fb, s1f1 = sb * fb;
	float w[8] = {ta * s0f0, ta * s1f0, tb * s0f0, tb * s1f0, ta * s0f1, ta * s1f1, tb * s0f1, tb * s1f1};
	float v[8];
#pragma unroll
	for (int q = 0; q < 8; q++) v[q] = w[q] * val;
	const int lane = threadIdx.x & 63, g = lane & 7;
	// run id inside the aligned group of 8 lanes: a run = consecutive lanes with the same key (equal keys that are
	// separated by another key are different runs and are not merged)
	const int prev = __builtin_amdgcn_update_dpp(-2, key, 0x111, 0xf, 0xf, false);  // row_shr:1
	const bool head = (g == 0) || (prev != key);
	int rid = head ? 1 : 0;
	{
		int t = __builtin_amdgcn_update_dpp(0, rid, 0x111, 0xf, 0xf, false);
		rid += (g >= 1) ? t : 0;
		t = __builtin_amdgcn_update_dpp(0, rid, 0x112, 0xf, 0xf, false);
		rid += (g >= 2) ? t : 0;
		t = __builtin_amdgcn_update_dpp(0, rid, 0x114, 0xf, 0xf, false);
		rid += (g >= 4) ? t : 0;
	}
#pragma unroll
	for (int st = 1; st <= 4; st <<= 1) {
		const bool take = (g + st < 8) && (dpp_shl_i(rid, st) == rid);
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const float wn = dpp_shl(w[q], st), vn = dpp_shl(v[q], st);
			w[q] += take ? wn : 0.f;
			v[q] += take ? vn : 0.f;
		}
	}
	if (head && key >= 0) {
		const int64_t off[8] = {0, 1, Y, 1 + Y, Z, 1 + Z, Y + Z, 1 + Y + Z};
#pragma unroll
		for (int q = 0; q < 8; q++) {
			atomicAdd(sum + key + off[q], w[q]);
			atomicAdd(ref + key + off[q], v[q]);
		}
	}
}
__global__ void __launch_bounds__(BLOCK)
k_p2g_mac_atomic(Dim d, float* __restrict__ vel, float* __restrict__ weight, int64_t np, int64_t ps, const float* __restrict__ pos,
                 const int32_t* __restrict__ pflag, const float* __restrict__ pvel, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	const bool act = (p < np) && !skip_particle(pflag, ptype, exclude, p);
	const int64_t q = act ? p : 0;
	const float x = pos[q], y = pos[ps + q], z = pos[2 * ps + q];
	const float ux = pvel[q], uy = pvel[ps + q], uz = pvel[2 * ps + q];
	const Bi b = build_index(d, x, y, z), s = build_index_shift(d, x, y, z);
	const int64_t n = d.n;
	const int kx = act ? (int)(((int64_t)b.zi * d.sy + b.yi) * d.sx + s.xi) : -1;
	const int ky = act ? (int)(((int64_t)b.zi * d.sy + s.yi) * d.sx + b.xi) : -1;
	const int kz = act ? (int)(((int64_t)s.zi * d.sy + b.yi) * d.sx + b.xi) : -1;
	p2g_component_grouped(vel, weight, kx, d.Y, d.Z, b.t0, b.t1, s.s0, s.s1, b.f0, b.f1, ux);
	p2g_component_grouped(vel + n, weight + n, ky, d.Y, d.Z, s.t0, s.t1, b.s0, b.s1, b.f0, b.f1, uy);
	p2g_component_grouped(vel + 2 * n, weight + 2 * n, kz, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, s.f0, s.f1, uz);
}
// Block-privatised scatter: a block of 256 consecutive particles (spatially coherent in practice) first accumulates its
// 256 x 48 contributions in an LDS hash table keyed by the flat grid address (ds atomics), then flushes each occupied slot
// with ONE global atomic pair.  A cell-coherent block touches a few hundred distinct addresses, so global atomics drop by
// an order of magnitude; when the table overflows (incoherent particle order) the contribution goes straight to global.
constexpr int P2G_SLOTS = 2048;
__device__ __forceinline__ void lds_scatter(int* __restrict__ keys, float* __restrict__ tw, float* __restrict__ tv, float* __restrict__ gref,
                                            float* __restrict__ gsum, int addr, float w, float v) {
	unsigned h = ((unsigned)addr * 2654435761u) >> 21;  // 11 bits
#pragma unroll 1
	for (int probe = 0; probe < 12; probe++) {
		const int k = atomicCAS(&keys[h], -1, addr);
		if (k == -1 || k == addr) {
			atomicAdd(&tw[h], w);
			atomicAdd(&tv[h], v);
			return;
		}
		h = (h + 1) & (P2G_SLOTS - 1);
	}
	atomicAdd(gsum + addr, w);
	atomicAdd(gref + addr, v);
}
__global__ void __launch_bounds__(BLOCK)
k_p2g_mac_lds(Dim d, float* __restrict__ vel, float* __restrict__ weight, int64_t np, int64_t ps, const float* __restrict__ pos,
              const int32_t* __restrict__ pflag, const float* __restrict__ pvel, const int32_t* __restrict__ ptype, int exclude) {
	__shared__ int keys[P2G_SLOTS];
	__shared__ float tw[P2G_SLOTS], tv[P2G_SLOTS];
	for (int i = threadIdx.x; i < P2G_SLOTS; i += BLOCK) {
		keys[i] = -1;
		tw[i] = 0.f;
		tv[i] = 0.f;
	}
	__syncthreads();
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p < np && !skip_particle(pflag, ptype, exclude, p)) {
		const float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
		const Bi b = build_index(d, x, y, z), s = build_index_shift(d, x, y, z);
		const int n = (int)d.n, Y = (int)d.Y, Z = (int)d.Z;
#define P2G_COMP(BASE, SA, SB, TA, TB, FA, FB, VAL)                                                                          \
	{                                                                                                                        \
		const int base = (BASE);                                                                                             \
		const float val = (VAL);                                                                                             \
		_Pragma("unroll") for (int c = 0; c < 8; c++) {                                                                      \
			const float w = ((c & 2) ? (TB) : (TA)) * (((c & 1) ? (SB) : (SA)) * ((c & 4) ? (FB) : (FA)));                   \
			lds_scatter(keys, tw, tv, vel, weight, base + (c & 1) + ((c & 2) ? Y : 0) + ((c & 4) ? Z : 0), w, w * val);     \
		}                                                                                                                    \
	}
		P2G_COMP((b.zi * d.sy + b.yi) * d.sx + s.xi, s.s0, s.s1, b.t0, b.t1, b.f0, b.f1, pvel[p])
		P2G_COMP(n + (b.zi * d.sy + s.yi) * d.sx + b.xi, b.s0, b.s1, s.t0, s.t1, b.f0, b.f1, pvel[ps + p])
		P2G_COMP(2 * n + (s.zi * d.sy + b.yi) * d.sx + b.xi, b.s0, b.s1, b.t0, b.t1, s.f0, s.f1, pvel[2 * ps + p])
#undef P2G_COMP
	}
	__syncthreads();
	for (int i = threadIdx.x; i < P2G_SLOTS; i += BLOCK) {
		const int k = keys[i];
		if (k >= 0) {
			atomicAdd(weight + k, tw[i]);
			atomicAdd(vel + k, tv[i]);
		}
	}
}
// the reference's KERNEL(pts, single) order: one thread walks the particles in index order
__global__ void k_p2g_mac_sequential(Dim d, float* vel, float* weight, int64_t np, int64_t ps, const float* pos,
                                     const int32_t* pflag, const float* pvel, const int32_t* ptype, int exclude) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	for (int64_t p = 0; p < np; p++) {
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		p2g_mac_one<false>(d, vel, weight, pos[p], pos[ps + p], pos[2 * ps + p], pvel[p], pvel[ps + p], pvel[2 * ps + p]);
	}
}
// weight.stomp(1e-6) ; vel.safeDivide(weight) ; velOld.copyFrom(vel)  (flip.cpp:653-659) fused: 3n scalars
__global__ void __launch_bounds__(BLOCK)
k_p2g_mac_finish(int64_t n3, float* __restrict__ vel, float* __restrict__ velOld, float* __restrict__ weight) {   // velOld nullable (APIC)
	for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n3; i += (int64_t)gridDim.x * BLOCK) {
		float w = weight[i];
		if (w < 1e-6f) w = 0.f;
		weight[i] = w;
		float v = vel[i];
		v = (w != 0.f) ? (v / w) : v;
		vel[i] = v;
		if (velOld) velOld[i] = v;
	}
}

// knApicMapLinearMACGridToVec3, apic.cpp:112-173: per particle and face the trilinear sample and its gradient weights
template <int COMP>
__device__ __forceinline__ void apic_g2p_face(const Dim& d, const float* __restrict__ vg, float px, float py, float pz, float out[4]) {
	const ApicFace a = apic_face<COMP>(d, px, py, pz);
	float v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int64_t node = a.gidx + i + j * d.Y + k * d.Z;
				const float val = (node >= 0 && node < d.n) ? vg[node] : 0.f;
				const float gi = i ? 1.f : -1.f, gj = j ? 1.f : -1.f, gk = k ? 1.f : -1.f;
				v += a.W[0][i] * a.W[1][j] * a.W[2][k] * val;
				g0 += gi * a.W[1][j] * a.W[2][k] * val;
				g1 += a.W[0][i] * gj * a.W[2][k] * val;
				g2 += a.W[0][i] * a.W[1][j] * gk * val;
			}
	out[0] = v; out[1] = g0; out[2] = g1; out[3] = g2;
}
__global__ void __launch_bounds__(BLOCK)
k_g2p_apic(Dim d, const float* __restrict__ vel, int64_t np, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ pflag,
           float* __restrict__ pvel, float* __restrict__ cpx, float* __restrict__ cpy, float* __restrict__ cpz,
           const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) return;
	const float px = pos[p], py = pos[ps + p], pz = pos[2 * ps + p];
	float u[4], v[4], w[4] = {0.f, 0.f, 0.f, 0.f};
	apic_g2p_face<0>(d, vel, px, py, pz, u);
	apic_g2p_face<1>(d, vel + d.n, px, py, pz, v);
	if (d.is3d) apic_g2p_face<2>(d, vel + 2 * d.n, px, py, pz, w);
	pvel[p] = u[0]; pvel[ps + p] = v[0]; pvel[2 * ps + p] = w[0];
	cpx[p] = u[1]; cpx[ps + p] = u[2]; cpx[2 * ps + p] = u[3];
	cpy[p] = v[1]; cpy[ps + p] = v[2]; cpy[2 * ps + p] = v[3];
	cpz[p] = w[1]; cpz[ps + p] = w[2]; cpz[2 * ps + p] = w[3];
}

// setInterpol, interpol.h:96-113 (cell-centred target, NCOMP planes, weights into a separate Real grid)
template <bool ATOMIC, int NCOMP>
__device__ __forceinline__ void p2g_cell_one(const Dim& d, float* __restrict__ target, float* __restrict__ wsum, float x, float y,
                                             float z, const float v[NCOMP]) {
	const Bi b = build_index(d, x, y, z);
	const int64_t Y = d.Y, Z = d.Z;
	const int64_t idx = (int64_t)b.xi + Y * b.yi + Z * b.zi;
	const float s0f0 = b.s0 * b.f0, s1f0 = b.s1 * b.f0, s0f1 = b.s0 * b.f1, s1f1 = b.s1 * b.f1;
	const float w0 = b.t0 * s0f0, wx = b.t0 * s1f0, wy = b.t1 * s0f0, wxy = b.t1 * s1f0;
	const float wz = b.t0 * s0f1, wxz = b.t0 * s1f1, wyz = b.t1 * s0f1, wxyz = b.t1 * s1f1;
	float* sum = wsum + idx;
	add_to<ATOMIC>(sum + Z, wz); add_to<ATOMIC>(sum + 1 + Z, wxz); add_to<ATOMIC>(sum + Y + Z, wyz); add_to<ATOMIC>(sum + 1 + Y + Z, wxyz);
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		float* ref = target + c * d.n + idx;
		add_to<ATOMIC>(ref + Z, wz * v[c]); add_to<ATOMIC>(ref + 1 + Z, wxz * v[c]); add_to<ATOMIC>(ref + Y + Z, wyz * v[c]); add_to<ATOMIC>(ref + 1 + Y + Z, wxyz * v[c]);
	}
	add_to<ATOMIC>(sum, w0); add_to<ATOMIC>(sum + 1, wx); add_to<ATOMIC>(sum + Y, wy); add_to<ATOMIC>(sum + 1 + Y, wxy);
#pragma unroll
	for (int c = 0; c < NCOMP; c++) {
		float* ref = target + c * d.n + idx;
		add_to<ATOMIC>(ref, w0 * v[c]); add_to<ATOMIC>(ref + 1, wx * v[c]); add_to<ATOMIC>(ref + Y, wy * v[c]); add_to<ATOMIC>(ref + 1 + Y, wxy * v[c]);
	}
}
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_p2g_cell_atomic(Dim d, float* __restrict__ target, float* __restrict__ wsum, int64_t np, int64_t ps, const float* __restrict__ pos,
                  const int32_t* __restrict__ pflag, const float* __restrict__ psrc) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (pflag[p] & MF_PDELETE) return;
	float v[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) v[c] = psrc[c * ps + p];
	p2g_cell_one<true, NCOMP>(d, target, wsum, pos[p], pos[ps + p], pos[2 * ps + p], v);
}
template <int NCOMP>
__global__ void k_p2g_cell_sequential(Dim d, float* target, float* wsum, int64_t np, int64_t ps, const float* pos,
                                      const int32_t* pflag, const float* psrc) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	for (int64_t p = 0; p < np; p++) {
		if (pflag[p] & MF_PDELETE) continue;
		float v[NCOMP];
		for (int c = 0; c < NCOMP; c++) v[c] = psrc[c * ps + p];
		p2g_cell_one<false, NCOMP>(d, target, wsum, pos[p], pos[ps + p], pos[2 * ps + p], v);
	}
}
// knSafeDivReal, flip.cpp:607-615
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK) k_safe_div_real(int64_t n, float* __restrict__ target, const float* __restrict__ wsum) {
	for (int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BLOCK) {
		const float w = wsum[i];
#pragma unroll
		for (int c = 0; c < NCOMP; c++) {
			const float t = target[c * n + i];
			target[c * n + i] = (w < 1e-6f) ? 0.f : ((w != 0.f) ? (t / w) : t);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------
// grid -> particle gathers
// ---------------------------------------------------------------------------------------------------------
// knMapLinearMACGridToVec3_PIC, flip.cpp:709-716
__global__ void __launch_bounds__(BLOCK)
k_g2p_pic(Dim d, const float* __restrict__ vel, int64_t np, int64_t ps, const float* __restrict__ pos,
          const int32_t* __restrict__ pflag, float* __restrict__ pvel, const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (skip_particle(pflag, ptype, exclude, p)) return;
	float vx, vy, vz;
	interpol_mac(d, vel, pos[p], pos[ps + p], pos[2 * ps + p], vx, vy, vz);
	pvel[p] = vx;
	pvel[ps + p] = vy;
	pvel[2 * ps + p] = vz;
}
// knMapLinearMACGridToVec3_FLIP, flip.cpp:724-736:
//   pvel = flipRatio*(v + (v2 - v1)) + (1.0 - flipRatio)*v2 ; the second scalar is a double, so that product is
//   formed in fp64 and rounded to fp32 (vectorbase.h:282-284)
__global__ void __launch_bounds__(BLOCK)
k_g2p_flip(Dim d, const float* __restrict__ vel, const float* __restrict__ velOld, int64_t np, int64_t ps,
           const float* __restrict__ pos, const int32_t* __restrict__ pflag, float* __restrict__ pvel, float flipRatio,
           const int32_t* __restrict__ ptype, int exclude) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (skip_particle(pflag, ptype, exclude, p)) return;
	const float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
	float a[3], b[3];
	interpol_mac(d, velOld, x, y, z, a[0], a[1], a[2]);
	interpol_mac(d, vel, x, y, z, b[0], b[1], b[2]);
	const double om = 1.0 - (double)flipRatio;
#pragma unroll
	for (int c = 0; c < 3; c++) {
		const float v = pvel[c * ps + p];
		const float delta = b[c] - a[c];
		const float t1 = flipRatio * (v + delta);
		const float t2 = (float)(om * (double)b[c]);
		pvel[c * ps + p] = t1 + t2;
	}
}
// knMapFromGrid<T>, flip.cpp:693-698
template <int NCOMP>
__global__ void __launch_bounds__(BLOCK)
k_g2p_cell(Dim d, const float* __restrict__ src, int64_t np, int64_t ps, const float* __restrict__ pos,
           const int32_t* __restrict__ pflag, float* __restrict__ ptarget) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	if (pflag[p] & MF_PDELETE) return;
	const Bi b = build_index(d, pos[p], pos[ps + p], pos[2 * ps + p]);
	const int64_t base = (int64_t)b.xi + d.Y * b.yi + d.Z * b.zi;
#pragma unroll
	for (int c = 0; c < NCOMP; c++) ptarget[c * ps + p] = tri8(src + c * d.n + base, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, b.f0, b.f1);
}

// ---------------------------------------------------------------------------------------------------------
// ParticleSystem::advectInGrid, particle.h:526-550 -- the four GridAdvectKernel runs, the host-side RK loops
// of integratePointSet (integrator.h:26-78, incl. the fork's extra `uTotal += u` at line 55) and the final
// KnClampPositions / KnDeleteInObstacle are fused: each particle is independent, the grid is read-only.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool in_bounds_pos(const Dim& d, float x, float y, float z, int bnd) {
	const int i = (int)x, j = (int)y, k = (int)z;  // toVec3i truncation, grid.h:65
	bool r = i >= bnd && j >= bnd && i < d.sx - bnd && j < d.sy - bnd;
	if (d.is3d)
		r = r && (k >= bnd && k < d.gsz - bnd);   // positions are global coordinates: the z extent is the whole domain's
	else
		r = r && (k == 0);
	return r;
}
__device__ __forceinline__ int flag_at(const Dim& d, const int32_t* __restrict__ flags, float x, float y, float z) {
	int k = (int)z - d.zoff;                     // plane inside the slab window (identity without a window)
	k = k < 0 ? 0 : (k > d.sz - 1 ? d.sz - 1 : k);
	return flags[(int64_t)(int)x + d.Y * (int)y + d.Z * k];  // FlagGrid::getAt, grid.h:324
}
struct AdvArgs {
	float dt;
	int deleteInObstacle, stopInObstacle, skipNew, exclude;
};
// GridAdvectKernel::op, particle.h:458-481 ; u keeps its previous value where the reference leaves it untouched
__device__ __forceinline__ void advect_eval(const Dim& d, const int32_t* __restrict__ flags, const float* __restrict__ vel,
                                            const AdvArgs& a, bool excluded, float x, float y, float z, int& pf, float u[3]) {
	if ((pf & MF_PDELETE) || excluded || (a.skipNew && (pf & MF_PNEW))) {
		u[0] = u[1] = u[2] = 0.f;
		return;
	}
	if (a.deleteInObstacle || a.stopInObstacle) {
		if (!in_bounds_pos(d, x, y, z, 1) || (flag_at(d, flags, x, y, z) & MF_OBSTACLE)) {
			if (a.stopInObstacle) u[0] = u[1] = u[2] = 0.f;
			if (a.deleteInObstacle) pf |= MF_PDELETE;
			return;
		}
	}
	float vx, vy, vz;
	interpol_mac(d, vel, x, y, z, vx, vy, vz);
	u[0] = vx * a.dt;
	u[1] = vy * a.dt;
	u[2] = vz * a.dt;
}
__global__ void __launch_bounds__(BLOCK)
k_advect_in_grid(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, int64_t np, int64_t ps,
                 float* __restrict__ pos, int32_t* __restrict__ pflag, AdvArgs a, int mode, const int32_t* __restrict__ ptype) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (p >= np) return;
	int pf = pflag[p];
	const bool excluded = ptype && (ptype[p] & a.exclude);
	const float x0[3] = {pos[p], pos[ps + p], pos[2 * ps + p]};
	float x[3] = {x0[0], x0[1], x0[2]};
	float u[3] = {0.f, 0.f, 0.f};
	advect_eval(d, flags, vel, a, excluded, x[0], x[1], x[2], pf, u);
	if (mode == MF_INT_EULER) {
		for (int c = 0; c < 3; c++) x[c] = x[c] + u[c];
	} else if (mode == MF_INT_RK2) {
		for (int c = 0; c < 3; c++) x[c] = x0[c] + 0.5f * u[c];
		advect_eval(d, flags, vel, a, excluded, x[0], x[1], x[2], pf, u);
		for (int c = 0; c < 3; c++) x[c] = x0[c] + u[c];
	} else {
		float ut[3];
		for (int c = 0; c < 3; c++) {
			ut[c] = u[c];
			x[c] = x0[c] + 0.5f * u[c];
			ut[c] = ut[c] + u[c];  // integrator.h:55 (fork)
		}
		advect_eval(d, flags, vel, a, excluded, x[0], x[1], x[2], pf, u);
		for (int c = 0; c < 3; c++) {
			x[c] = x0[c] + 0.5f * u[c];
			ut[c] = ut[c] + 2.f * u[c];
		}
		advect_eval(d, flags, vel, a, excluded, x[0], x[1], x[2], pf, u);
		for (int c = 0; c < 3; c++) {
			x[c] = x0[c] + u[c];
			ut[c] = ut[c] + 2.f * u[c];
		}
		advect_eval(d, flags, vel, a, excluded, x[0], x[1], x[2], pf, u);
		const float sixth = (float)(1. / 6.);
		for (int c = 0; c < 3; c++) x[c] = x0[c] + sixth * (ut[c] + u[c]);
	}
	if (!a.deleteInObstacle) {
		// KnClampPositions, particle.h:507-523
		if (!(pf & MF_PDELETE)) {
			if (excluded) {
				for (int c = 0; c < 3; c++) x[c] = x0[c];
			} else {
				if (!in_bounds_pos(d, x[0], x[1], x[2], 0)) {
					const float hi[3] = {(float)d.sx - 1.f, (float)d.sy - 1.f, (float)d.gsz - 1.f};
					for (int c = 0; c < 3; c++) x[c] = x[c] < 0.f ? 0.f : (x[c] > hi[c] ? hi[c] : x[c]);
				}
				if (a.stopInObstacle && (flag_at(d, flags, x[0], x[1], x[2]) & MF_OBSTACLE)) {
					// bisectBacktracePos, particle.h:494-504: oldp*(1.-(s+ds)) + newp*(s+ds)
					float s = 0.f;
					for (int it = 1; it < 5; ++it) {
						const float ds = 1.f / (float)(1 << it);
						const float sb = s + ds;
						const double sa = 1. - (double)sb;
						const float tx = (float)((double)x0[0] * sa) + x[0] * sb;
						const float ty = (float)((double)x0[1] * sa) + x[1] * sb;
						const float tz = (float)((double)x0[2] * sa) + x[2] * sb;
						if (!(flag_at(d, flags, tx, ty, tz) & MF_OBSTACLE)) s += ds;
					}
					const double sa = 1. - (double)s;
					for (int c = 0; c < 3; c++) x[c] = (float)((double)x0[c] * sa) + x[c] * s;
				}
			}
		}
	} else {
		// KnDeleteInObstacle, particle.h:485-491
		if (!(pf & MF_PDELETE)) {
			if (!in_bounds_pos(d, x[0], x[1], x[2], 1) || (flag_at(d, flags, x[0], x[1], x[2]) & MF_OBSTACLE)) pf |= MF_PDELETE;
		}
	}
	pos[p] = x[0];
	pos[ps + p] = x[1];
	pos[2 * ps + p] = x[2];
	pflag[p] = pf;
}

extern "C" {

int mf_map_parts_to_mac_accum(int sx, int sy, int sz, float* vel, float* weight, int64_t np, int64_t ps, const float* pos,
                              const int32_t* pflag, const float* pvel, const int32_t* ptype, int exclude, int deterministic,
                              void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(weight, 0, sizeof(float) * 3 * d.n, st));
	MF_HIP(hipMemsetAsync(vel, 0, sizeof(float) * 3 * d.n, st));
	if (np > 0) {
		static const bool sequential = getenv("MF_P2G_SEQUENTIAL") != nullptr;   // the one-thread walk, kept as a cross-check
		if (deterministic && !sequential)
			MF_TRY(p2g_ordered_mac(d, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude, st));
		else if (deterministic)
			hipLaunchKernelGGL(k_p2g_mac_sequential, dim3(1), dim3(64), 0, st, d, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude);
		else if (3 * d.n < ((int64_t)1 << 31) && !getenv("MF_P2G_NOLDS"))
			hipLaunchKernelGGL(k_p2g_mac_lds, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude);
		else
			hipLaunchKernelGGL(k_p2g_mac_atomic, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude);
	}
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_map_parts_to_mac_finish(int64_t n3, float* vel, float* velOld, float* weight, void* stream) {
	if (n3 <= 0) return 0;
	hipLaunchKernelGGL(k_p2g_mac_finish, dim3(blocks_for(n3, BLOCK, 2048)), dim3(BLOCK), 0, (hipStream_t)stream, n3, vel, velOld, weight);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* velOld, float* weight, int64_t np, int64_t ps,
                        const float* pos, const int32_t* pflag, const float* pvel, const int32_t* ptype, int exclude,
                        int deterministic, void* stream) {
	MF_TRY(mf_map_parts_to_mac_accum(sx, sy, sz, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude, deterministic, stream));
	return mf_map_parts_to_mac_finish(3 * (int64_t)sx * sy * sz, vel, velOld, weight, stream);
}
int mf_apic_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* mass, int64_t np, int64_t ps, const float* pos,
                             const int32_t* pflag, const float* pvel, const float* cpx, const float* cpy, const float* cpz,
                             const int32_t* ptype, int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (!mass) return fail("mf_apic_map_parts_to_mac: mass grid required");
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(mass, 0, sizeof(float) * 3 * d.n, st));
	MF_HIP(hipMemsetAsync(vel, 0, sizeof(float) * 3 * d.n, st));
	if (np > 0) MF_TRY(p2g_ordered_apic(d, vel, mass, np, ps, pos, pflag, pvel, cpx, cpy, cpz, ptype, exclude, st));
	hipLaunchKernelGGL(k_p2g_mac_finish, dim3(blocks_for(3 * d.n, BLOCK, 2048)), dim3(BLOCK), 0, st, 3 * d.n, vel, (float*)nullptr, mass);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_apic_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t ps, const float* pos,
                             const int32_t* pflag, float* pvel, float* cpx, float* cpy, float* cpz, const int32_t* ptype,
                             int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_g2p_apic, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, np, ps, pos, pflag, pvel, cpx, cpy, cpz, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t ps, const float* pos,
                        const int32_t* pflag, float* pvel, const int32_t* ptype, int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_g2p_pic, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, np, ps, pos, pflag, pvel, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_flip_velocity_update(int sx, int sy, int sz, const float* vel, const float* velOld, int64_t np, int64_t ps,
                            const float* pos, const int32_t* pflag, float* pvel, float flipRatio, const int32_t* ptype,
                            int exclude, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_g2p_flip, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, vel, velOld, np, ps, pos, pflag, pvel, flipRatio, ptype, exclude);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_map_parts_to_grid(int sx, int sy, int sz, int ncomp, float* target, float* wtmp, int64_t np, int64_t ps,
                         const float* pos, const int32_t* pflag, const float* psrc, int deterministic, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (ncomp != 1 && ncomp != 3) return fail("ncomp must be 1 or 3");
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(target, 0, sizeof(float) * ncomp * d.n, st));
	MF_HIP(hipMemsetAsync(wtmp, 0, sizeof(float) * d.n, st));
	if (np > 0) {
		static const bool sequential = getenv("MF_P2G_SEQUENTIAL") != nullptr;
		if (deterministic && !sequential) {
			MF_TRY(p2g_ordered_cell(d, ncomp, target, wtmp, np, ps, pos, pflag, psrc, st));
		} else if (deterministic) {
			if (ncomp == 1)
				hipLaunchKernelGGL((k_p2g_cell_sequential<1>), dim3(1), dim3(64), 0, st, d, target, wtmp, np, ps, pos, pflag, psrc);
			else
				hipLaunchKernelGGL((k_p2g_cell_sequential<3>), dim3(1), dim3(64), 0, st, d, target, wtmp, np, ps, pos, pflag, psrc);
		} else {
			if (ncomp == 1)
				hipLaunchKernelGGL((k_p2g_cell_atomic<1>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, target, wtmp, np, ps, pos, pflag, psrc);
			else
				hipLaunchKernelGGL((k_p2g_cell_atomic<3>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, target, wtmp, np, ps, pos, pflag, psrc);
		}
	}
	if (ncomp == 1)
		hipLaunchKernelGGL((k_safe_div_real<1>), dim3(blocks_for(d.n, BLOCK, 2048)), dim3(BLOCK), 0, st, d.n, target, wtmp);
	else
		hipLaunchKernelGGL((k_safe_div_real<3>), dim3(blocks_for(d.n, BLOCK, 2048)), dim3(BLOCK), 0, st, d.n, target, wtmp);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_map_grid_to_parts(int sx, int sy, int sz, int ncomp, const float* source, int64_t np, int64_t ps, const float* pos,
                         const int32_t* pflag, float* ptarget, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	if (ncomp != 1 && ncomp != 3) return fail("ncomp must be 1 or 3");
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	if (ncomp == 1)
		hipLaunchKernelGGL((k_g2p_cell<1>), dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, source, np, ps, pos, pflag, ptarget);
	else
		hipLaunchKernelGGL((k_g2p_cell<3>), dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, source, np, ps, pos, pflag, ptarget);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_advect_in_grid(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t ps, float* pos,
                      int32_t* pflag, float dt, int integrationMode, int deleteInObstacle, int stopInObstacle, int skipNew,
                      const int32_t* ptype, int exclude, float* scratch, void* stream) {
	(void)scratch;  // the fused kernel keeps x0 / u / uTotal in registers
	MF_TRY(check_dim(sx, sy, sz));
	if (integrationMode < 0 || integrationMode > 2) return fail("unknown integration type");
	if (np <= 0) return 0;
	const Dim d = mkdim(sx, sy, sz);
	AdvArgs a;
	a.dt = dt;
	a.deleteInObstacle = deleteInObstacle;
	a.stopInObstacle = stopInObstacle;
	a.skipNew = skipNew;
	a.exclude = exclude;
	hipLaunchKernelGGL(k_advect_in_grid, dim3(nblk_n(np)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, np, ps, pos, pflag, a, integrationMode, ptype);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
