// p2g_ordered.hip -- particle -> grid transfers that reproduce the reference's SERIAL scatter bit for bit, in parallel.
//
// knMapLinearVec3ToMACGrid / knMapLinear<T> are KERNEL(pts, single): one thread walks the particles in index order and
// adds w and w*val into the 8 nodes around each particle (setInterpolMAC / setInterpol, util/interpol.h:96-113, 166-213).
// Every node is therefore an independent fp32 accumulator that receives its contributions in increasing particle index.
// That is a gather: a node (i,j,k) of component c is reached exactly by the particles whose component base cell is
// (i-di, j-dj, k-dk), di,dj,dk in {0,1}, through the corner (di,dj,dk).
//
// Round 3: ONE binning pass serves all three MAC components, and it is ours (no library sort on the hot path).
//   1. bin by the plain base cell B = BUILD_INDEX(pos) (the cell-centred base): wave-aggregated histogram, exclusive scan,
//      placement with wave-aggregated cursors, then every cell's run is put into increasing particle index (a run is short:
//      insertion sort per cell, a workgroup sort for the rare crowded cell).  Result: start[cell], order[slot].
//   2. the sorted payload (per component a record {w0, w1, w2, v} and a link word to the next entries, p | e << 28) is written once; e = (ex, ey, ez) says for each axis whether the
//      SHIFTED base of BUILD_INDEX_SHIFT is B or B + 1 (it is always one of the two).  The base cell of MAC component X is
//      (B.x + ex, B.y, B.z), and likewise for Y, Z.
//   3. one thread per node merges, by particle index, the 3 x 2 x 2 runs of the cells B with B.c in {n.c-2, n.c-1, n.c} (the
//      outer two filtered by e: 12 sorted lists, 8 for a cell-centred grid) and accumulates  acc_w += w ; acc_v += w*val  in
//      exactly the reference's order.  Heads are (p << 4 | list) words, so the minimum names the list.
// No atomics in the sums, no order dependence, same bits as the reference on every run.  In 2D the two z-corners alias the
// same node (strideZ = 0); the per-particle order of those two adds follows the statement order of the reference.
#include "common.h"
#include <limits.h>

using namespace mf;

namespace {

static inline unsigned nblk_n(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK > 0 ? (n + BLOCK - 1) / BLOCK : 1); }

constexpr int PBITS = 28;                          // particle index bits of a payload word; e lives above
constexpr unsigned PMASK = (1u << PBITS) - 1u;
constexpr int SMALL_RUN = 32;                      // longest run the per-cell insertion sort takes
constexpr int BIG_LDS = 4096;                      // longest run the workgroup sort takes in LDS
constexpr int SCAN_ITEMS = 16;                     // per thread and scan block

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

// ---- binning ------------------------------------------------------------------------------------------------------------
// keys: KEYMODE 3 = plain base cell (serves MODE 0..3); KEYMODE 10 + c = flat APIC face index of component c.
template <int KEYMODE>
__device__ __forceinline__ int key_of(const Dim& d, float x, float y, float z) {
	int64_t key;
	if (KEYMODE == 3) {
		const Bi b = build_index(d, x, y, z);
		key = (int64_t)b.xi + d.sx * ((int64_t)b.yi + (int64_t)d.sy * b.zi);
	} else {
		key = apic_face<KEYMODE - 10>(d, x, y, z).gidx;
	}
	return (key >= 0 && key < d.n) ? (int)key : (int)d.n;        // a position that maps nowhere (NaN, APIC outside) is skipped
}

// lanes of a wave that hold the same key as their left neighbour form a run; its first lane speaks for it
struct Run {
	bool head;
	int len, first;
};
__device__ __forceinline__ Run run_of(int key) {
	const int lane = threadIdx.x & 63;
	const int prev = __shfl_up(key, 1);
	Run r;
	r.head = lane == 0 || prev != key;
	const unsigned long long m = __ballot(r.head);
	const unsigned long long below = m & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
	r.first = 63 - __clzll(below);
	const unsigned long long above = (lane == 63) ? 0ull : (m >> (lane + 1));
	r.len = above ? (__ffsll((long long)above)) : (64 - lane);
	return r;
}

template <int KEYMODE>
__global__ void __launch_bounds__(BLOCK)
k_bin_keys(Dim d, int64_t np, int64_t ps, const float* __restrict__ pos, const int32_t* __restrict__ pflag, const int32_t* __restrict__ ptype,
           int exclude, int32_t* __restrict__ keys, int32_t* __restrict__ counts) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	int key = (int)d.n + 1;          // lanes past the end: a key of their own, never counted
	if (p < np) {
		key = (int)d.n;
		if (!((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude)))) key = key_of<KEYMODE>(d, pos[p], pos[ps + p], pos[2 * ps + p]);
		keys[p] = key;
	}
	const Run r = run_of(key);
	if (r.head && key < d.n) atomicAdd(&counts[key], r.len);
}

// exclusive scan of counts[0..n] -> start[0..n] in three launches (block sums, their scan, the blocks)
__global__ void __launch_bounds__(BLOCK)
k_scan_sums(int64_t n, const int32_t* __restrict__ counts, int32_t* __restrict__ sums) {
	const int64_t base = (int64_t)blockIdx.x * BLOCK * SCAN_ITEMS;
	int s = 0;
	for (int q = 0; q < SCAN_ITEMS; q++) {
		const int64_t i = base + (int64_t)q * BLOCK + threadIdx.x;
		if (i < n) s += counts[i];
	}
	__shared__ int sh[BLOCK / 64];
	for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ int block_excl_scan(int v, int* total) {
	// exclusive scan of one value per thread over the block (BLOCK = 256 = 4 waves)
	__shared__ int shw[BLOCK / 64];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	int inc = v;
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(inc, o);
		if (lane >= o) inc += t;
	}
	__syncthreads();
	if (lane == 63) shw[w] = inc;
	__syncthreads();
	int off = 0;
	for (int q = 0; q < w; q++) off += shw[q];
	*total = shw[0] + shw[1] + shw[2] + shw[3];
	return off + inc - v;
}
__global__ void __launch_bounds__(BLOCK)
k_scan_top(int nb, int32_t* __restrict__ sums) {
	int carry = 0;
	for (int base = 0; base < nb; base += BLOCK) {
		const int i = base + threadIdx.x;
		const int v = i < nb ? sums[i] : 0;
		int tot;
		const int ex = block_excl_scan(v, &tot);
		if (i < nb) sums[i] = carry + ex;
		carry += tot;
		__syncthreads();
	}
}
__global__ void __launch_bounds__(BLOCK)
k_scan_blocks(int64_t n, const int32_t* __restrict__ counts, const int32_t* __restrict__ sums, int32_t* __restrict__ start) {
	// thread t owns SCAN_ITEMS consecutive entries
	const int64_t base = (int64_t)blockIdx.x * BLOCK * SCAN_ITEMS + (int64_t)threadIdx.x * SCAN_ITEMS;
	int v[SCAN_ITEMS], s = 0;
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; q++) {
		v[q] = (base + q < n) ? counts[base + q] : 0;
		s += v[q];
	}
	int tot;
	int run = sums[blockIdx.x] + block_excl_scan(s, &tot);
#pragma unroll
	for (int q = 0; q < SCAN_ITEMS; q++) {
		if (base + q < n) start[base + q] = run;
		run += v[q];
	}
}
// k_scan_sums reads strided, k_scan_blocks reads blocked: both cover [blockIdx * BLOCK * SCAN_ITEMS, +BLOCK * SCAN_ITEMS)

__global__ void __launch_bounds__(BLOCK)
k_bin_place(int64_t n, int64_t np, const int32_t* __restrict__ keys, const int32_t* __restrict__ start, int32_t* __restrict__ counts,
            int32_t* __restrict__ order) {
	const int64_t p = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	const int key = p < np ? keys[p] : (int)n + 1;
	const Run r = run_of(key);
	int old = 0;
	if (r.head && key < n) old = atomicSub(&counts[key], r.len);       // the run takes the slots [old - len, old) of its cell
	old = __shfl(old, r.first);
	const int hlen = __shfl(r.len, r.first);                             // r.len is the run's length on its first lane only
	if (key < n) order[start[key] + old - hlen + ((threadIdx.x & 63) - r.first)] = (int)p;
}

// every cell's run into increasing particle index; runs longer than SMALL_RUN go to the workgroup sort
__global__ void __launch_bounds__(BLOCK)
k_bin_sort_cells(int64_t n, const int32_t* __restrict__ start, int32_t* __restrict__ order, int32_t* __restrict__ nbig, int32_t* __restrict__ big) {
	const int64_t c = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (c >= n) return;
	const int a = start[c], e = start[c + 1];
	if (e - a < 2) return;
	if (e - a > SMALL_RUN) {
		big[atomicAdd(nbig, 1)] = (int)c;
		return;
	}
	for (int i = a + 1; i < e; i++) {
		const int v = order[i];
		int j = i - 1;
		if (order[j] <= v) continue;
		while (j >= a && order[j] > v) {
			order[j + 1] = order[j];
			j--;
		}
		order[j + 1] = v;
	}
}
// one workgroup per crowded cell: bitonic sort in LDS up to BIG_LDS entries, rank counting into `tmp` beyond that
__global__ void __launch_bounds__(BLOCK)
k_bin_sort_big(const int32_t* __restrict__ start, int32_t* __restrict__ order, const int32_t* __restrict__ nbig, const int32_t* __restrict__ big,
               int32_t* __restrict__ tmp) {
	__shared__ int sh[BIG_LDS];
	const int nb = *nbig;
	for (int b = blockIdx.x; b < nb; b += gridDim.x) {
		const int c = big[b];
		const int a = start[c], cnt = start[c + 1] - a;
		if (cnt <= BIG_LDS) {
			int m = 64;
			while (m < cnt) m <<= 1;
			for (int i = threadIdx.x; i < m; i += BLOCK) sh[i] = i < cnt ? order[a + i] : INT_MAX;
			__syncthreads();
			for (int k = 2; k <= m; k <<= 1)
				for (int j = k >> 1; j > 0; j >>= 1) {
					for (int i = threadIdx.x; i < m; i += BLOCK) {
						const int x = i ^ j;
						if (x > i) {
							const int u = sh[i], v = sh[x];
							const bool up = (i & k) == 0;
							if ((u > v) == up) {
								sh[i] = v;
								sh[x] = u;
							}
						}
					}
					__syncthreads();
				}
			for (int i = threadIdx.x; i < cnt; i += BLOCK) order[a + i] = sh[i];
			__syncthreads();
		} else {
			// particle indices are distinct: the rank of an entry is the number of smaller entries of its run
			for (int i0 = 0; i0 < cnt; i0 += BLOCK) {
				const int i = i0 + threadIdx.x;
				const int v = i < cnt ? order[a + i] : INT_MAX;
				int rank = 0;
				for (int t0 = 0; t0 < cnt; t0 += BIG_LDS) {
					const int tn = cnt - t0 < BIG_LDS ? cnt - t0 : BIG_LDS;
					__syncthreads();
					for (int q = threadIdx.x; q < tn; q += BLOCK) sh[q] = order[a + t0 + q];
					__syncthreads();
					for (int q = 0; q < tn; q++) rank += sh[q] < v;
				}
				if (i < cnt) tmp[a + rank] = v;
			}
			__syncthreads();
			for (int i = threadIdx.x; i < cnt; i += BLOCK) order[a + i] = tmp[a + i];
			__syncthreads();
		}
	}
}

// sorted payload, one record set per target kind T (0..2: MAC component, 3: cell-centred grid), so that a contribution costs the
// gather two loads whose addresses follow from the slot alone:
//   rec[T][slot]  = {w0, w1, w2, v}: the fractional weights of BUILD_INDEX (s1, t1, f1 after the clamps), the one along the
//                   component's own axis replaced by that of the shifted half of BUILD_INDEX_SHIFT, and the value to spread.
//                   The other weight of a pair is (float)(1. - (double)w1) in every case: that is how the reference forms it, and
//                   the clamps set the pairs (1,0) / (0,1).
//   link[T][slot] = p | e << 28 | dist << 29: this entry's particle, its e bit along T's axis, and the distance to the next entry of
//                   the same cell with the same e bit (0: none, 7: seven or more -- walk)
// e = (ex, ey, ez): shifted base = B + e per axis.  spe[slot] = p | e << PBITS.
template <int NV>
__global__ void __launch_bounds__(BLOCK)
k_bin_payload(Dim d, const int32_t* __restrict__ start, int64_t ps, const float* __restrict__ pos, const float* __restrict__ pval, int64_t vstride,
              const int32_t* __restrict__ order, bool mac, float4* __restrict__ rec, float* __restrict__ sv, int64_t cp, uint32_t* __restrict__ spe) {
	const int64_t slot = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (slot >= start[d.n]) return;          // the binned particles (skipped ones have no slot)
	const int p = order[slot];
	const float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
	const Bi b = build_index(d, x, y, z);
	const Bi s = build_index_shift(d, x, y, z);
	const unsigned e = (unsigned)(s.xi - b.xi) | ((unsigned)(s.yi - b.yi) << 1) | ((unsigned)(s.zi - b.zi) << 2);
	if (mac) {
		rec[slot] = make_float4(s.s1, b.t1, b.f1, pval[p]);
		rec[cp + slot] = make_float4(b.s1, s.t1, b.f1, pval[vstride + p]);
		rec[2 * cp + slot] = make_float4(b.s1, b.t1, s.f1, pval[2 * vstride + p]);
	} else {
		rec[slot] = make_float4(b.s1, b.t1, b.f1, pval[p]);
#pragma unroll
		for (int c = 1; c < NV; c++) sv[(c - 1) * cp + slot] = pval[c * vstride + p];
	}
	spe[slot] = (unsigned)p | (e << PBITS);
}
// link words of every slot: a look-ahead inside the slot's own cell run
__global__ void __launch_bounds__(BLOCK)
k_bin_links(Dim d, const int32_t* __restrict__ start, const int32_t* __restrict__ keys, const uint32_t* __restrict__ spe, bool mac, int64_t cp,
            uint32_t* __restrict__ link) {
	const int64_t slot = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (slot >= start[d.n]) return;
	const unsigned w = spe[slot];
	const int end = start[keys[w & PMASK] + 1];
	const unsigned own = w & PMASK;
	if (!mac) {
		link[slot] = own;
		return;
	}
	unsigned dist[3] = {0u, 0u, 0u};
	int found = 0;
	for (int64_t a = slot + 1; a < end && found != 7; a++) {
		const unsigned u = spe[a];
		const unsigned dd = a - slot < 7 ? (unsigned)(a - slot) : 7u;
#pragma unroll
		for (int c = 0; c < 3; c++)
			if (!(found & (1 << c)) && ((u ^ w) >> (PBITS + c) & 1u) == 0u) {
				dist[c] = dd;
				found |= 1 << c;
			}
	}
#pragma unroll
	for (int c = 0; c < 3; c++) link[c * cp + slot] = own | (((w >> (PBITS + c)) & 1u) << 28) | (dist[c] << 29);
}

// ---- the gather: one thread per node ------------------------------------------------------------------------------------
// where a node's merge reads its lists from: the sorted arrays in global memory ...
struct SrcGlobal {
	const float4* __restrict__ rec_;
	const uint32_t* __restrict__ link_;
	const float* __restrict__ sv_;
	int64_t cp;
	const int32_t* __restrict__ start;
	int sx, sy;
	// slots [a, e) of a cell's run; *delta: what turns such a slot into the global slot of its record
	__device__ __forceinline__ void range(int bx, int by, int bz, unsigned& a, unsigned& e, int& delta) const {
		const int64_t c = (int64_t)bx + sx * ((int64_t)by + (int64_t)sy * bz);
		a = (unsigned)start[c];
		e = (unsigned)start[c + 1];
		delta = 0;
	}
	__device__ __forceinline__ unsigned link(unsigned a) const { return link_[a]; }
	__device__ __forceinline__ float4 rec(unsigned g) const { return rec_[g]; }
	__device__ __forceinline__ float sv(int c, unsigned g) const { return sv_[c * cp + g]; }
	// a list's cursor in LDS: {slot, end | offsets << PBITS}
	typedef uint2 Cursor;
	static __device__ __forceinline__ Cursor cur_make(unsigned a, unsigned e, unsigned meta) { return make_uint2(a, e | (meta << PBITS)); }
	static __device__ __forceinline__ void cur_get(const Cursor& c, unsigned& a, unsigned& e, unsigned& meta) { a = c.x; e = c.y & PMASK; meta = c.y >> PBITS; }
	static __device__ __forceinline__ void cur_set(Cursor& c, unsigned a) { c.x = a; }
};
// ... or the wave's copy of the link words of its tile's neighbourhood in LDS (local slot numbers; records stay in global memory)
struct SrcWave {
	const float4* __restrict__ rec_;
	const uint32_t* link_;       // LDS
	const int* lstart;           // LDS: local slot of every cell of the neighbourhood (+ one past the end)
	const int* gstart;           // LDS: global slot of every cell's first entry
	int x0, y0, z0, bxn, byn;
	__device__ __forceinline__ void range(int bx, int by, int bz, unsigned& a, unsigned& e, int& delta) const {
		const int idx = (bx - x0) + bxn * ((by - y0) + byn * (bz - z0));
		a = (unsigned)lstart[idx];
		e = (unsigned)lstart[idx + 1];
		delta = gstart[idx] - (int)a;
	}
	__device__ __forceinline__ unsigned link(unsigned a) const { return link_[a]; }
	__device__ __forceinline__ float4 rec(unsigned g) const { return rec_[g]; }
	__device__ __forceinline__ float sv(int, unsigned) const { return 0.f; }
	// local slots fit 12 bits (WCAP <= 4095): slot | end << 12 | offsets << 24 in one word -- LDS per wave is what bounds the occupancy
	typedef unsigned Cursor;
	static __device__ __forceinline__ Cursor cur_make(unsigned a, unsigned e, unsigned meta) { return a | (e << 12) | (meta << 24); }
	static __device__ __forceinline__ void cur_get(const Cursor& c, unsigned& a, unsigned& e, unsigned& meta) { a = c & 4095u; e = (c >> 12) & 4095u; meta = c >> 24; }
	static __device__ __forceinline__ void cur_set(Cursor& c, unsigned a) { c = (c & ~4095u) | a; }
};

// ---- the merge of one node: NCOMP values per particle (1 for a MAC component / Real grid, 3 for a Vec3 grid) ----
// The kernel takes as long as one node's chain  minimum -> cursor -> next head  times its contributions: list heads live in
// registers as (p << 4 | list) words (one v_min chain finds the next particle AND its list), the cursors of the 12 lists live
// in LDS ([list][thread]: a dynamic list index costs one ds_read instead of a 12-way select), the next head comes from link
// words (no filter loop), and the weights / value of a contribution are consumed one iteration later, so that their load never
// sits on the chain.
template <int MODE, int NCOMP, int NT, class Src>
__device__ __forceinline__ void merge_node(const Dim& d, int i, int j, int k, const Src& src, typename Src::Cursor (*s_ce)[NT], int (*s_delta)[NT],
                                           float* __restrict__ ref, int64_t rstride, float* __restrict__ sum) {
	constexpr int NL = (MODE == 3) ? 8 : 12;
	constexpr unsigned INF = 0xffffffffu;
	// list q: offsets (o0, o1, o2) below the node along (shifted axis, next axis, next axis) for a MAC component -- 0 / 1 / 2
	// along the shifted axis, of which offset 2 takes the entries with e = 1, offset 0 those with e = 0, offset 1 both -- and
	// along (x, y, z) for a cell-centred grid.  meta = o0 | o1 << 2 | o2 << 3.
	unsigned head[NL];
#pragma unroll
	for (int q = 0; q < NL; q++) {
		int o[3];
		unsigned meta;
		if constexpr (MODE == 3) {
			o[0] = q & 1; o[1] = (q >> 1) & 1; o[2] = q >> 2;
			meta = (unsigned)(o[0] | (o[1] << 2) | (o[2] << 3));
		} else {
			constexpr int a1 = (MODE + 1) % 3, a2 = (MODE + 2) % 3;
			o[MODE] = q % 3; o[a1] = (q / 3) & 1; o[a2] = q / 6;
			meta = (unsigned)(o[MODE] | (o[a1] << 2) | (o[a2] << 3));
		}
		const int bx = i - o[0], by = j - o[1], bz = k - o[2];
		const bool ok = bx >= 0 && by >= 0 && bz >= 0 && (d.is3d || o[2] == 0);
		const unsigned want = (MODE == 3) ? 2u : ((meta & 3u) == 2u ? 1u : ((meta & 3u) == 0u ? 0u : 2u));
		unsigned a = 0, e = 0;
		int delta = 0;
		if (ok) src.range(bx, by, bz, a, e, delta);
		unsigned h = INF;        // first entry that passes the filter
		while (a < e) {
			const unsigned y = src.link(a);
			if (want == 2u || ((y >> 28) & 1u) == want) {
				h = ((y & PMASK) << 4) | (unsigned)q;
				break;
			}
			a++;
		}
		s_ce[q][threadIdx.x] = Src::cur_make(a, e, meta);
		s_delta[q][threadIdx.x] = delta;
		head[q] = h;
	}
	float acc_w = 0.f, acc_v[NCOMP];
#pragma unroll
	for (int c = 0; c < NCOMP; c++) acc_v[c] = 0.f;
	// the contribution whose record is still in flight (consumed one iteration after its load was issued; two iterations, three
	// register slots in rotation, was built as well: 1.49 instead of 1.42 ms per mapPartsToMAC -- the record is not what the chain
	// waits for)
	bool pend = false;
	float4 prec = make_float4(0.f, 0.f, 0.f, 0.f);
	float pv1 = 0.f, pv2 = 0.f;
	int pd0 = 0, pd1 = 0, pd2 = 0;
	auto accumulate = [&](const float4& r, float v1, float v2, int d0, int d1, int d2) {
		const float sw = d0 ? r.x : (float)(1. - (double)r.x);
		const float tw = d1 ? r.y : (float)(1. - (double)r.y);
		const float g1 = r.z, g0 = (float)(1. - (double)r.z);
		float v[3] = {r.w, v1, v2};
		if (d.is3d) {
			const float w = tw * (sw * (d2 ? g1 : g0));
			acc_w += w;
#pragma unroll
			for (int cc = 0; cc < NCOMP; cc++) acc_v[cc] += w * v[cc];
		} else {
			// strideZ == 0: both z-corners land on this node, in the reference's statement order
			const float wa = tw * (sw * ((MODE == 2) ? g0 : g1));
			const float wb = tw * (sw * ((MODE == 2) ? g1 : g0));
			acc_w += wa;
			acc_w += wb;
#pragma unroll
			for (int cc = 0; cc < NCOMP; cc++) {
				acc_v[cc] += wa * v[cc];
				acc_v[cc] += wb * v[cc];
			}
		}
	};
	for (;;) {
		unsigned best = head[0];
#pragma unroll
		for (int q = 1; q < NL; q++) best = head[q] < best ? head[q] : best;
		if (best == INF) break;
		const unsigned bq = best & 15u;
		typename Src::Cursor ce = s_ce[bq][threadIdx.x];
		unsigned slot, e, meta;
		Src::cur_get(ce, slot, e, meta);
		const unsigned g = (unsigned)((int)slot + s_delta[bq][threadIdx.x]);       // the record's global slot
		const unsigned osh = meta & 3u;
		const bool both = (MODE == 3) || osh == 1u;
		const unsigned lk = src.link(slot);
		// where the winner's list goes on: slot + 1 for a list that takes every entry, slot + dist (from the link word) for one that
		// takes the entries with one e bit; the next head is that entry's particle -- one more link word
		const unsigned dist = both ? 1u : (lk >> 29);
		unsigned a = slot + dist;
		const unsigned ny = (dist != 0u && a < e) ? src.link(a) : 0u;
		const float4 r = src.rec(g);
		float v1 = 0.f, v2 = 0.f;
		if (NCOMP == 3) {
			v1 = src.sv(0, g);
			v2 = src.sv(1, g);
		}
		if (pend) accumulate(prec, pv1, pv2, pd0, pd1, pd2);
		unsigned h = (dist != 0u && a < e) ? (((ny & PMASK) << 4) | bq) : INF;
		if (!both && dist == 7u && h != INF) {
			// seven or more slots away: the entry at slot + 7 need not be the one -- walk
			const unsigned want = osh == 2u ? 1u : 0u;
			h = INF;
			while (a < e) {
				const unsigned y = src.link(a);
				if (((y >> 28) & 1u) == want) {
					h = ((y & PMASK) << 4) | bq;
					break;
				}
				a++;
			}
		}
		Src::cur_set(ce, a);
		s_ce[bq][threadIdx.x] = ce;
#pragma unroll
		for (int q = 0; q < NL; q++) head[q] = (bq == (unsigned)q) ? h : head[q];
		// corner of this node as seen from the particle: offset - e along the shifted axis, the offset itself elsewhere
		if constexpr (MODE == 3) {
			pd0 = meta & 1; pd1 = (meta >> 2) & 1; pd2 = (meta >> 3) & 1;
		} else {
			int dd[3];
			constexpr int a1 = (MODE + 1) % 3, a2 = (MODE + 2) % 3;
			dd[MODE] = (int)osh - (int)((lk >> 28) & 1u);
			dd[a1] = (meta >> 2) & 1;
			dd[a2] = (meta >> 3) & 1;
			pd0 = dd[0]; pd1 = dd[1]; pd2 = dd[2];
		}
		prec = r;
		pv1 = v1;
		pv2 = v2;
		pend = true;
	}
	if (pend) accumulate(prec, pv1, pv2, pd0, pd1, pd2);
	const int64_t node = (int64_t)i + d.sx * ((int64_t)j + (int64_t)d.sy * k);
	sum[node] = acc_w;
#pragma unroll
	for (int cc = 0; cc < NCOMP; cc++) ref[cc * rstride + node] = acc_v[cc];
}

// one thread per node; a wave owns a compact GX x GY x GZ tile of nodes, so that the runs its 64 merges walk (a cell's run feeds up
// to 12 nodes of the tile) are pulled into the CU's L1 once.  Lists read from global memory: 2-D grids, Vec3 sources, and the
// fallback of k_gather_wave.
constexpr int GBLOCK = 64, GX = 4, GY = 4, GZ = 4;
template <int MODE, int NCOMP>
__global__ void __launch_bounds__(GBLOCK)
k_gather(Dim d, int ntx, int nty, const float4* __restrict__ rec, const uint32_t* __restrict__ link, const float* __restrict__ sv, int64_t cp,
         const int32_t* __restrict__ start, float* __restrict__ ref, int64_t rstride, float* __restrict__ sum) {
	__shared__ SrcGlobal::Cursor s_ce[(MODE == 3) ? 8 : 12][GBLOCK];
	__shared__ int s_delta[(MODE == 3) ? 8 : 12][GBLOCK];
	const int tile = xcd_swizzle(blockIdx.x, gridDim.x), t = threadIdx.x;
	const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
	// 2-D grids: the tile is GX*GZ x GY x 1
	const int i = d.is3d ? tx * GX + (t % GX) : tx * (GX * GZ) + (t % (GX * GZ));
	const int j = d.is3d ? ty * GY + (t / GX) % GY : ty * GY + t / (GX * GZ);
	const int k = d.is3d ? tz * GZ + t / (GX * GY) : 0;
	if (i >= d.sx || j >= d.sy || k >= d.sz) return;
	const SrcGlobal src{rec, link, sv, cp, start, d.sx, d.sy};
	merge_node<MODE, NCOMP, GBLOCK>(d, i, j, k, src, s_ce, s_delta, ref, rstride, sum);
}

// 3-D grids, one value per particle: the wave first copies the link words of every cell its 64 nodes reach -- (GX + 2) x (GY + 1) x
// (GZ + 1) cells for the X component, each row of cells one contiguous piece of the sorted arrays -- into LDS (6 KB; with the
// cursors 16 KB per wave: 10 waves per CU -- the merge is a serial chain per node, only more waves per SIMD hide its latencies), so that the chain  minimum -> cursor -> link -> next head  never leaves the CU; only the
// record of a contribution (off the chain, consumed one iteration later) comes from global memory.  With the links in global
// memory every contribution pulls its own cache line: 67 M lines for 1.7 M load instructions per component at 128^3 / 3.8 M
// particles, 17 % L1 hits -- the kernel ran at the L2 -> L1 request rate.  (Round 3 also tried a 128-node workgroup staging records
// AND links, 79 KB: one wave per SIMD left, 0.61 ms against 0.35 ms.)  A tile whose neighbourhood holds more than WCAP particles
// merges from global memory.
constexpr int WCAP = 1536;
static_assert(WCAP <= 4095, "local slots are 12-bit fields of SrcWave::Cursor");
template <int MODE>
__global__ void __launch_bounds__(GBLOCK)
k_gather_wave(Dim d, int ntx, int nty, const float4* __restrict__ rec, const uint32_t* __restrict__ link, const int32_t* __restrict__ start,
              float* __restrict__ ref, float* __restrict__ sum) {
	constexpr int NL = (MODE == 3) ? 8 : 12;
	constexpr int LX = (MODE == 0) ? 2 : 1, LY = (MODE == 1) ? 2 : 1, LZ = (MODE == 2) ? 2 : 1;
	constexpr int BXN = GX + LX, BYN = GY + LY, BZN = GZ + LZ, NCELL = BXN * BYN * BZN, NROW = BYN * BZN;
	constexpr int PER = (NCELL + GBLOCK - 1) / GBLOCK;
	__shared__ SrcWave::Cursor s_ce[NL][GBLOCK];
	__shared__ int s_delta[NL][GBLOCK];
	__shared__ __attribute__((aligned(16))) uint32_t s_link[WCAP];
	__shared__ int s_lstart[NCELL + 1], s_gstart[NCELL];
	const int t = threadIdx.x;
	const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
	const int tx = tile % ntx, ty = (tile / ntx) % nty, tz = tile / (ntx * nty);
	const int x0 = tx * GX - LX, y0 = ty * GY - LY, z0 = tz * GZ - LZ;
	// most tiles of a liquid scene have no particle anywhere near: the cells of a row of the neighbourhood are consecutive entries of
	// `start`, so one lane per row (two loads) tells whether the whole neighbourhood is empty before the per-cell table is built
	{
		static_assert(NROW <= GBLOCK, "one lane per row of the neighbourhood");
		int rowcnt = 0;
		if (t < NROW) {
			const int by = y0 + t % BYN, bz = z0 + t / BYN;
			const int xa = x0 < 0 ? 0 : x0, xb = x0 + BXN > d.sx ? d.sx : x0 + BXN;      // cells [xa, xb) of the row
			if (by >= 0 && bz >= 0 && by < d.sy && bz < d.sz && xa < xb) {
				const int64_t c = d.sx * ((int64_t)by + (int64_t)d.sy * bz);
				rowcnt = start[c + xb] - start[c + xa];
			}
		}
		if (!__any(rowcnt != 0)) {
			const int i = tx * GX + (t % GX), j = ty * GY + (t / GX) % GY, k = tz * GZ + t / (GX * GY);
			if (i < d.sx && j < d.sy && k < d.sz) {
				const int64_t node = (int64_t)i + d.sx * ((int64_t)j + (int64_t)d.sy * k);
				sum[node] = 0.f;
				ref[node] = 0.f;
			}
			return;
		}
	}
	// cell table: run length of every cell of the neighbourhood (0 outside the grid), exclusive scan over the wave -> local slots
	int cnt[PER], gst[PER], tot = 0;
#pragma unroll
	for (int q = 0; q < PER; q++) {
		const int idx = t * PER + q;
		cnt[q] = 0;
		gst[q] = 0;
		if (idx < NCELL) {
			const int bx = x0 + idx % BXN, by = y0 + (idx / BXN) % BYN, bz = z0 + idx / (BXN * BYN);
			if (bx >= 0 && by >= 0 && bz >= 0 && bx < d.sx && by < d.sy && bz < d.sz) {
				const int64_t c = (int64_t)bx + d.sx * ((int64_t)by + (int64_t)d.sy * bz);
				gst[q] = start[c];
				cnt[q] = start[c + 1] - gst[q];
			}
		}
		tot += cnt[q];
	}
	int inc = tot;
	for (int o = 1; o < 64; o <<= 1) {
		const int u = __shfl_up(inc, o);
		if (t >= o) inc += u;
	}
	const int total = __shfl(inc, 63);
	int run = inc - tot;
#pragma unroll
	for (int q = 0; q < PER; q++) {
		const int idx = t * PER + q;
		if (idx < NCELL) {
			s_lstart[idx] = run;
			s_gstart[idx] = gst[q];
		}
		run += cnt[q];
	}
	if (t == 0) s_lstart[NCELL] = total;
	__syncthreads();
	const int i = tx * GX + (t % GX), j = ty * GY + (t / GX) % GY, k = tz * GZ + t / (GX * GY);
	const bool inside = i < d.sx && j < d.sy && k < d.sz;
	if (total > WCAP) {
		if (inside) {
			const SrcGlobal src{rec, link, nullptr, 0, start, d.sx, d.sy};
			// the fallback's cursors hold global slots: two words each, in the space of the link copy it does not make
			static_assert(sizeof(SrcGlobal::Cursor) * NL * GBLOCK <= sizeof(uint32_t) * WCAP, "fallback cursors alias s_link");
			merge_node<MODE, 1, GBLOCK>(d, i, j, k, src, (SrcGlobal::Cursor(*)[GBLOCK])s_link, s_delta, ref, 0, sum);
		}
		return;
	}
	if (total == 0) {
		if (inside) {
			const int64_t node = (int64_t)i + d.sx * ((int64_t)j + (int64_t)d.sy * k);
			sum[node] = 0.f;
			ref[node] = 0.f;
		}
		return;
	}
	// the rows of the neighbourhood: cells along x are consecutive in the sorted arrays; entry t (and t + 64, ...) of RU rows per
	// round trip
	constexpr int RU = 8;
	const int xin = x0 < 0 ? -x0 : 0;          // first cell of a row inside the grid (cells outside are empty)
	for (int r0 = 0; r0 < NROW; r0 += RU) {
		uint32_t vl[RU];
		int dst[RU];
#pragma unroll
		for (int u = 0; u < RU; u++) {
			const int r = r0 + u;
			dst[u] = -1;
			if (r < NROW) {
				const int first = r * BXN;
				const int l0 = s_lstart[first], len = s_lstart[first + BXN] - l0;
				if (t < len) {
					dst[u] = l0 + t;
					vl[u] = link[s_gstart[first + xin] + t];
				}
			}
		}
#pragma unroll
		for (int u = 0; u < RU; u++)
			if (dst[u] >= 0) s_link[dst[u]] = vl[u];
	}
	for (int r = 0; r < NROW; r++) {            // rows longer than the wave
		const int first = r * BXN;
		const int l0 = s_lstart[first], len = s_lstart[first + BXN] - l0;
		const int g0 = s_gstart[first + xin];
		for (int q = t + GBLOCK; q < len; q += GBLOCK) s_link[l0 + q] = link[g0 + q];
	}
	__syncthreads();
	if (inside) {
		const SrcWave src{rec, s_link, s_lstart, s_gstart, x0, y0, z0, BXN, BYN};
		merge_node<MODE, 1, GBLOCK>(d, i, j, k, src, s_ce, s_delta, ref, 0, sum);
	}
}

// ---- APIC (knApicMapLinearVec3ToMACGrid, apic.cpp:19-90): same gather, other weights.  The reference addresses the 8 nodes
// by FLAT index gidx + dX[i] + dY[j] + dZ[k] with no bounds check; the gather follows the flat arithmetic (a node n is fed by
// the base indices n - (i + j*Y + k*Z)), faces whose base index lies outside [0, n) are skipped like in the oracle.
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
	int32_t* keys = nullptr;     // [cap_p]
	int32_t* order = nullptr;    // [cap_p]
	int32_t* tmp = nullptr;      // [cap_p]   (rank-counting sort of a crowded cell)
	float* pay = nullptr;        // [18 * cap_p]  rec[3] (float4), link[3] (uint2) in slot order; extra value planes of a Vec3 source alias rec[1]
	uint32_t* spe = nullptr;     // [cap_p]
	int32_t* counts = nullptr;   // [cap_n + 1]
	int32_t* start = nullptr;    // [cap_n + 1]
	int32_t* big = nullptr;      // [cap_n]
	int32_t* sums = nullptr;     // [scan blocks of cap_n + 1]
	int32_t* nbig = nullptr;
	int64_t cap_p = 0, cap_n = 0;
};
Scratch g_scratch[16];

int get_scratch(int64_t np, int64_t n, Scratch** out) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	Scratch& s = g_scratch[dev];
	if (np > s.cap_p || n > s.cap_n) MF_HIP(hipDeviceSynchronize());
	if (!s.nbig) MF_HIP(hipMalloc((void**)&s.nbig, sizeof(int32_t)));
	if (np > s.cap_p) {
		for (void* q : {(void*)s.keys, (void*)s.order, (void*)s.tmp, (void*)s.pay, (void*)s.spe})
			if (q) MF_HIP(hipFree(q));
		s.cap_p = np + np / 8 + 1024;
		MF_HIP(hipMalloc((void**)&s.keys, sizeof(int32_t) * s.cap_p));
		MF_HIP(hipMalloc((void**)&s.order, sizeof(int32_t) * s.cap_p));
		MF_HIP(hipMalloc((void**)&s.tmp, sizeof(int32_t) * s.cap_p));
		MF_HIP(hipMalloc((void**)&s.pay, sizeof(float) * 18 * s.cap_p));
		MF_HIP(hipMalloc((void**)&s.spe, sizeof(uint32_t) * s.cap_p));
	}
	if (n > s.cap_n) {
		for (void* q : {(void*)s.counts, (void*)s.start, (void*)s.big, (void*)s.sums})
			if (q) MF_HIP(hipFree(q));
		s.cap_n = n;
		MF_HIP(hipMalloc((void**)&s.counts, sizeof(int32_t) * (s.cap_n + 1)));
		MF_HIP(hipMalloc((void**)&s.start, sizeof(int32_t) * (s.cap_n + 1)));
		MF_HIP(hipMalloc((void**)&s.big, sizeof(int32_t) * s.cap_n));
		MF_HIP(hipMalloc((void**)&s.sums, sizeof(int32_t) * ((s.cap_n + 1) / (BLOCK * SCAN_ITEMS) + 2)));
	}
	*out = &s;
	return 0;
}

// start[0..n], order[0..start[n]) : the particles of every key in increasing particle index
template <int KEYMODE>
int bin_particles(const Dim& d, int64_t np, int64_t ps, const float* pos, const int32_t* pflag, const int32_t* ptype, int exclude, Scratch* s,
                  hipStream_t st) {
	const int64_t n1 = d.n + 1;
	const int nsb = (int)((n1 + BLOCK * SCAN_ITEMS - 1) / (BLOCK * SCAN_ITEMS));
	MF_HIP(hipMemsetAsync(s->counts, 0, sizeof(int32_t) * n1, st));
	MF_HIP(hipMemsetAsync(s->nbig, 0, sizeof(int32_t), st));
	hipLaunchKernelGGL((k_bin_keys<KEYMODE>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, np, ps, pos, pflag, ptype, exclude, s->keys, s->counts);
	hipLaunchKernelGGL(k_scan_sums, dim3(nsb), dim3(BLOCK), 0, st, n1, s->counts, s->sums);
	hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(BLOCK), 0, st, nsb, s->sums);
	hipLaunchKernelGGL(k_scan_blocks, dim3(nsb), dim3(BLOCK), 0, st, n1, s->counts, s->sums, s->start);
	hipLaunchKernelGGL(k_bin_place, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d.n, np, s->keys, s->start, s->counts, s->order);
	hipLaunchKernelGGL(k_bin_sort_cells, dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d.n, s->start, s->order, s->nbig, s->big);
	hipLaunchKernelGGL(k_bin_sort_big, dim3(256), dim3(BLOCK), 0, st, s->start, s->order, s->nbig, s->big, s->tmp);
	MF_LAUNCH_CHECK();
	return 0;
}

int check_np(int64_t np) {
	if (np >= ((int64_t)1 << PBITS)) return fail("ordered P2G: more than 2^%d particles; use the atomic mode (setDeterministicP2G(False))", PBITS);
	return 0;
}

template <int COMP>
int run_apic(const Dim& d, int64_t np, int64_t ps, const float* pos, const int32_t* pflag, const int32_t* ptype, int exclude,
             const float* pvc, const float* cp, float* vel, float* mass, hipStream_t st) {
	MF_TRY(check_np(np));
	Scratch* s;
	MF_TRY(get_scratch(np, d.n, &s));
	MF_TRY((bin_particles<10 + COMP>(d, np, ps, pos, pflag, ptype, exclude, s, st)));
	hipLaunchKernelGGL((k_gather_apic<COMP>), dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d, ps, pos, pvc, cp, s->order, s->start, vel, mass);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // namespace

namespace mf {

// vel / weight: SoA MAC grids (already zeroed or not: every node is written)
int p2g_ordered_mac(const Dim& d, float* vel, float* weight, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                    const float* pvel, const int32_t* ptype, int exclude, hipStream_t st) {
	MF_TRY(check_np(np));
	Scratch* s;
	MF_TRY(get_scratch(np, d.n, &s));
	MF_TRY((bin_particles<3>(d, np, ps, pos, pflag, ptype, exclude, s, st)));
	const int64_t cp = s->cap_p;
	float4* rec = (float4*)s->pay;
	uint32_t* link = (uint32_t*)(s->pay + 12 * cp);
	// the number of binned particles is start[n], known on the device only: the payload kernels are launched over np slots and
	// bound themselves
	hipLaunchKernelGGL((k_bin_payload<3>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, s->start, ps, pos, pvel, ps, s->order, true, rec, (float*)nullptr, cp, s->spe);
	hipLaunchKernelGGL(k_bin_links, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, s->start, s->keys, s->spe, true, cp, link);
	const int ntx = d.is3d ? (d.sx + GX - 1) / GX : (d.sx + GX * GZ - 1) / (GX * GZ), nty = (d.sy + GY - 1) / GY, ntz = d.is3d ? (d.sz + GZ - 1) / GZ : 1;
	const unsigned nb = (unsigned)(ntx * nty * ntz);
	if (d.is3d) {
		hipLaunchKernelGGL((k_gather_wave<0>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec, link, s->start, vel, weight);
		hipLaunchKernelGGL((k_gather_wave<1>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec + cp, link + cp, s->start, vel + d.n, weight + d.n);
		hipLaunchKernelGGL((k_gather_wave<2>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec + 2 * cp, link + 2 * cp, s->start, vel + 2 * d.n, weight + 2 * d.n);
	} else {
		hipLaunchKernelGGL((k_gather<0, 1>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec, link, (const float*)nullptr, cp, s->start, vel, d.n, weight);
		hipLaunchKernelGGL((k_gather<1, 1>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec + cp, link + cp, (const float*)nullptr, cp, s->start, vel + d.n, d.n, weight + d.n);
		hipLaunchKernelGGL((k_gather<2, 1>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec + 2 * cp, link + 2 * cp, (const float*)nullptr, cp, s->start, vel + 2 * d.n, d.n, weight + 2 * d.n);
	}
	MF_LAUNCH_CHECK();
	return 0;
}
// target: ncomp planes; wsum: Real grid
int p2g_ordered_cell(const Dim& d, int ncomp, float* target, float* wsum, int64_t np, int64_t ps, const float* pos,
                     const int32_t* pflag, const float* psrc, hipStream_t st) {
	MF_TRY(check_np(np));
	Scratch* s;
	MF_TRY(get_scratch(np, d.n, &s));
	MF_TRY((bin_particles<3>(d, np, ps, pos, pflag, nullptr, 0, s, st)));
	const int64_t cp = s->cap_p;
	float4* rec = (float4*)s->pay;
	float* sv = s->pay + 4 * cp;                 // the second and third value plane of a Vec3 source
	uint32_t* link = (uint32_t*)(s->pay + 12 * cp);
	const int ntx = d.is3d ? (d.sx + GX - 1) / GX : (d.sx + GX * GZ - 1) / (GX * GZ), nty = (d.sy + GY - 1) / GY, ntz = d.is3d ? (d.sz + GZ - 1) / GZ : 1;
	const unsigned nb = (unsigned)(ntx * nty * ntz);
	if (ncomp == 1)
		hipLaunchKernelGGL((k_bin_payload<1>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, s->start, ps, pos, psrc, ps, s->order, false, rec, sv, cp, s->spe);
	else
		hipLaunchKernelGGL((k_bin_payload<3>), dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, s->start, ps, pos, psrc, ps, s->order, false, rec, sv, cp, s->spe);
	hipLaunchKernelGGL(k_bin_links, dim3(nblk_n(np)), dim3(BLOCK), 0, st, d, s->start, s->keys, s->spe, false, cp, link);
	if (ncomp == 1 && d.is3d)
		hipLaunchKernelGGL((k_gather_wave<3>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec, link, s->start, target, wsum);
	else if (ncomp == 1)
		hipLaunchKernelGGL((k_gather<3, 1>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec, link, sv, cp, s->start, target, d.n, wsum);
	else
		hipLaunchKernelGGL((k_gather<3, 3>), dim3(nb), dim3(GBLOCK), 0, st, d, ntx, nty, rec, link, sv, cp, s->start, target, d.n, wsum);
	MF_LAUNCH_CHECK();
	return 0;
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
