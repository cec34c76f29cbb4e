/*
 * oracle/manta_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the hot path of zoharl3/mantaflow (semi-Lagrangian / MacCormack advection,
 * GridCg pressure projection with the modified-incomplete-Cholesky preconditioner, FLIP particle<->grid
 * transfers) behind the C ABI of include/manta_hip.h, with HOST pointers.  Every function cites the
 * reference file:line it follows and keeps the reference's evaluation order and float/double promotion
 * points, so that results are bit-identical to the reference compiled without FMA contraction
 * (gcc, x86-64, -O3; the reference build uses no -march flag).  Compile with -ffp-contract=off.
 *
 * It is NOT the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * It is pinned against the reference itself (oracle/_ref/libmanta_ref.so, built by oracle/ref.mk) in
 * tests/test_oracle_vs_reference.py and against the committed golden vectors in tests/golden/.
 *
 * Layouts: see include/manta_hip.h (Vec3/MAC grids and particle vectors are structure-of-arrays).
 * Loops that the reference runs under `#pragma omp for` (preprocessor/codegen_kernel.cpp:213-299) carry the
 * same pragma here; kernels the reference runs single-threaded (MIC sweeps conjugategrad.cpp:135-159, the
 * particle->grid scatter flip.cpp:619) are single-threaded here too.
 */
#include "../include/manta_hip.h"
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_err[512];
static int fail(const char* msg) {
	snprintf(g_err, sizeof g_err, "%s", msg);
	return 1;
}
const char* mf_last_error(void) { return g_err; }
const char* mf_backend(void) { return "oracle"; }
int mf_abi_version(void) { return MF_ABI_VERSION; }
/* the restatement has no liquid-scene shortcut: it sweeps and streams every cell (include/manta_hip.h: mf_cg_last_shortcut) */
int mf_cg_last_shortcut(int32_t* out) {
	out[0] = out[1] = out[2] = 0;
	return 0;
}

typedef struct {
	int sx, sy, sz;
	int is3d;
	int zoff, gsz;      /* z-slab window (multi-GPU tests): planes [zoff, zoff+sz) of a global grid of gsz planes */
	int64_t X, Y, Z, n; /* strides (Z == 0 in 2-D, grid.cpp:56) */
} Dim;
static _Thread_local int g_slab_zoff = 0, g_slab_gsz = 0;
static _Thread_local int g_slab_src_zoff = 0, g_slab_src_gsz = 0;
int mf_mic_check(void* stream) {
	(void)stream;
	return 0;
}
int mf_pack_matrix(int sx, int sy, int sz, const int32_t* flags, const float* A0, const float* Ai, const float* Aj, const float* Ak, void* st) {
	(void)A0;
	(void)sx; (void)sy; (void)sz; (void)flags; (void)Ai; (void)Aj; (void)Ak; (void)st; /* an accelerator of the HIP library only */
	return 0;
}
int mf_mic_apply_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* var1, const float* Aprecond,
                         const float* Ai, const float* Aj, const float* Ak, double* dot, void* st) {
	int rc = mf_mic_apply(sx, sy, sz, flags, dst, var1, Aprecond, Ai, Aj, Ak, st);
	if (rc) return rc;
	const int64_t n = (int64_t)sx * sy * sz;
	double acc = 0.0;
	for (int64_t i = 0; i < n; i++) acc += (double)(dst[i] * var1[i]);
	dot[0] = acc;
	return 0;
}
int mf_mic_init_blocked(int sx, int sy, int sz, const int32_t* flags, float* Ap, const float* A0, const float* Ai,
                        const float* Aj, const float* Ak, int rows_j, int cells_x, void* st) {
	(void)rows_j; /* the serial sweep needs no schedule: it runs over whatever (cut) coefficients it is given */
	(void)cells_x;
	return mf_mic_init(sx, sy, sz, flags, Ap, A0, Ai, Aj, Ak, st);
}
int mf_set_mic_mode(const char* name) {
	(void)name;
	return 0;
}
int mf_set_slab_window(int zoff, int gsz) {
	if (gsz < 0 || zoff < 0 || (gsz > 0 && zoff >= gsz)) return fail("invalid slab window");
	g_slab_zoff = zoff;
	g_slab_gsz = gsz;
	return 0;
}
int mf_set_slab_window_source(int zoff, int gsz) {
	if (gsz < 0 || zoff < 0 || (gsz > 0 && zoff >= gsz)) return fail("invalid source slab window");
	g_slab_src_zoff = zoff;
	g_slab_src_gsz = gsz;
	return 0;
}
static Dim mkdim(int sx, int sy, int sz) {
	Dim d;
	d.sx = sx;
	d.sy = sy;
	d.sz = sz;
	d.is3d = sz > 1;
	d.zoff = g_slab_gsz > 0 ? g_slab_zoff : 0;
	d.gsz = g_slab_gsz > 0 ? g_slab_gsz : sz;
	d.X = 1;
	d.Y = sx;
	d.Z = d.is3d ? (int64_t)sx * sy : 0;
	d.n = (int64_t)sx * sy * sz;
	return d;
}
static Dim mkdim_src(int sx, int sy, int sz) { /* the source grid of a two-grid call: its own window */
	Dim d = mkdim(sx, sy, sz);
	d.zoff = g_slab_src_gsz > 0 ? g_slab_src_zoff : 0;
	d.gsz = g_slab_src_gsz > 0 ? g_slab_src_gsz : sz;
	return d;
}
#define IDX(d, i, j, k) ((int64_t)(i) + (d).Y * (j) + (d).Z * (k))
/* FOR_IJK_BND, kernel.h:39-42 */
#define K0(d, b) ((d).is3d ? (b) : 0)
#define K1(d, b) ((d).is3d ? (d).sz - (b) : 1)

/* ================================================================================================
 * element-wise ops
 * ============================================================================================== */
int mf_fill_f32(int64_t n, float* a, float v, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) a[i] = v;
	return 0;
}
int mf_fill_i32(int64_t n, int32_t* a, int32_t v, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) a[i] = v;
	return 0;
}
int mf_copy_f32(int64_t n, float* dst, const float* src, void* s) {
	(void)s;
	memcpy(dst, src, sizeof(float) * n); /* grid.cpp:230 */
	return 0;
}
/* gridScaledAdd, grid.h:514: me[idx] += factor * other[idx] (all fp32) */
int mf_grid_scaled_add(int64_t n, float* me, const float* other, float factor, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] += factor * other[i];
	return 0;
}
/* UpdateSearchVec, conjugategrad.cpp:193-196 */
int mf_update_search_vec(int64_t n, float* dst, const float* src, float factor, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) dst[i] = src[i] + factor * dst[i];
	return 0;
}
/* GridDotProduct, conjugategrad.cpp:175-178: fp32 product added to an fp64 accumulator.  The reference
 * combines thread-local sums under `omp critical`; here: fixed chunks of 4096 summed in index order. */
static double dot64(int64_t n, const float* a, const float* b) {
	const int64_t CH = 4096;
	int64_t nch = (n + CH - 1) / CH;
	double* part = (double*)malloc(sizeof(double) * (nch ? nch : 1));
#pragma omp parallel for
	for (int64_t c = 0; c < nch; c++) {
		double acc = 0.0;
		int64_t e = (c + 1) * CH < n ? (c + 1) * CH : n;
		for (int64_t i = c * CH; i < e; i++) acc += (a[i] * b[i]);
		part[c] = acc;
	}
	double r = 0.0;
	for (int64_t c = 0; c < nch; c++) r += part[c];
	free(part);
	return r;
}
int mf_grid_dot(int64_t n, const float* a, const float* b, double* r, void* s) {
	(void)s;
	*r = dot64(n, a, b);
	return 0;
}
/* GridSumSqr, commonkernels.h:32-35: square of the value converted to double */
static double sumsqr64(int64_t n, const float* a) {
	const int64_t CH = 4096;
	int64_t nch = (n + CH - 1) / CH;
	double* part = (double*)malloc(sizeof(double) * (nch ? nch : 1));
#pragma omp parallel for
	for (int64_t c = 0; c < nch; c++) {
		double acc = 0.0;
		int64_t e = (c + 1) * CH < n ? (c + 1) * CH : n;
		for (int64_t i = c * CH; i < e; i++) acc += (double)a[i] * (double)a[i];
		part[c] = acc;
	}
	double r = 0.0;
	for (int64_t c = 0; c < nch; c++) r += part[c];
	free(part);
	return r;
}
int mf_grid_sum_sqr(int64_t n, const float* a, double* r, void* s) {
	(void)s;
	*r = sumsqr64(n, a);
	return 0;
}
/* CompMinReal / CompMaxReal, grid.cpp:185-196 */
static void minmax(int64_t n, const float* a, float* mn, float* mx) {
	float lo = FLT_MAX, hi = -FLT_MAX;
#pragma omp parallel for reduction(min : lo) reduction(max : hi)
	for (int64_t i = 0; i < n; i++) {
		if (a[i] < lo) lo = a[i];
		if (a[i] > hi) hi = a[i];
	}
	*mn = lo;
	*mx = hi;
}
int mf_grid_min_max(int64_t n, const float* a, float* mn, float* mx, void* s) {
	(void)s;
	minmax(n, a, mn, mx);
	return 0;
}
/* Grid<Real>::getMaxAbs, grid.cpp:356-360 */
static float maxabs(int64_t n, const float* a) {
	float lo, hi;
	minmax(n, a, &lo, &hi);
	lo = fabsf(lo);
	hi = fabsf(hi);
	return lo > hi ? lo : hi;
}
int mf_grid_max_abs(int64_t n, const float* a, float* r, void* s) {
	(void)s;
	*r = maxabs(n, a);
	return 0;
}
/* Grid<Vec3>::getMaxAbs = sqrt(CompMaxVec), grid.cpp:218-224,367-369; normSquare vectorbase.h:392-395 */
int mf_grid_max_abs_vec3(int64_t n, const float* a, float* r, void* s) {
	(void)s;
	float hi = -FLT_MAX;
#pragma omp parallel for reduction(max : hi)
	for (int64_t i = 0; i < n; i++) {
		float x = a[i], y = a[n + i], z = a[2 * n + i];
		float q = x * x + y * y + z * z;
		if (q > hi) hi = q;
	}
	*r = sqrtf(hi);
	return 0;
}
int mf_grid_stomp(int64_t n, float* a, float th, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++)
		if (a[i] < th) a[i] = 0;
	return 0;
}
int mf_grid_safe_divide(int64_t n, float* me, const float* other, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] = (other[i]) ? (me[i] / other[i]) : me[i];
	return 0;
}
int mf_grid_add_const(int64_t n, float* me, float v, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] += v;
	return 0;
}
int mf_grid_mult_const(int64_t n, float* me, float v, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] *= v;
	return 0;
}
int mf_grid_clamp(int64_t n, float* me, float lo, float hi, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) {
		float v = me[i];
		me[i] = v < lo ? lo : (v > hi ? hi : v);
	}
	return 0;
}
int mf_grid_add(int64_t n, float* me, const float* o, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] += o[i];
	return 0;
}
int mf_grid_sub(int64_t n, float* me, const float* o, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] -= o[i];
	return 0;
}
int mf_grid_mult(int64_t n, float* me, const float* o, void* s) {
	(void)s;
#pragma omp parallel for
	for (int64_t i = 0; i < n; i++) me[i] *= o[i];
	return 0;
}

/* ================================================================================================
 * pressure projection
 * ============================================================================================== */
/* ApplyMatrix, conjugategrad.h:118-133; 2-D variant :136-151.  KERNEL(idx): every cell. */
int mf_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                    const float* Ai, const float* Aj, const float* Ak, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t X = d.X, Y = d.Y, Z = d.Z;
#pragma omp parallel for
	for (int64_t idx = 0; idx < d.n; idx++) {
		if (!(flags[idx] & MF_FLUID)) {
			dst[idx] = src[idx];
			continue;
		}
		float r = src[idx] * A0[idx] + src[idx - X] * Ai[idx - X] + src[idx + X] * Ai[idx] + src[idx - Y] * Aj[idx - Y] +
		          src[idx + Y] * Aj[idx];
		if (d.is3d) r = r + src[idx - Z] * Ak[idx - Z] + src[idx + Z] * Ak[idx];
		dst[idx] = r;
	}
	return 0;
}

/* MakeLaplaceMatrix, conjugategrad.h:154-187 (bnd=1) */
int mf_make_laplace_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak,
                           const float* fr, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				if (!fr) {
					if (!(flags[idx - d.X] & MF_OBSTACLE)) A0[idx] += 1.;
					if (!(flags[idx + d.X] & MF_OBSTACLE)) A0[idx] += 1.;
					if (!(flags[idx - d.Y] & MF_OBSTACLE)) A0[idx] += 1.;
					if (!(flags[idx + d.Y] & MF_OBSTACLE)) A0[idx] += 1.;
					if (d.is3d && !(flags[idx - d.Z] & MF_OBSTACLE)) A0[idx] += 1.;
					if (d.is3d && !(flags[idx + d.Z] & MF_OBSTACLE)) A0[idx] += 1.;
					if (flags[idx + d.X] & MF_FLUID) Ai[idx] = -1.;
					if (flags[idx + d.Y] & MF_FLUID) Aj[idx] = -1.;
					if (d.is3d && (flags[idx + d.Z] & MF_FLUID)) Ak[idx] = -1.;
				} else {
					const float *fx = fr, *fy = fr + n, *fz = fr + 2 * n;
					A0[idx] += fx[idx];
					A0[idx] += fx[idx + d.X];
					A0[idx] += fy[idx];
					A0[idx] += fy[idx + d.Y];
					if (d.is3d) A0[idx] += fz[idx];
					if (d.is3d) A0[idx] += fz[idx + d.Z];
					if (flags[idx + d.X] & MF_FLUID) Ai[idx] = -fx[idx + d.X];
					if (flags[idx + d.Y] & MF_FLUID) Aj[idx] = -fy[idx + d.Y];
					if (d.is3d && (flags[idx + d.Z] & MF_FLUID)) Ak[idx] = -fz[idx + d.Z];
				}
			}
	return 0;
}

/* ghost-fluid helpers, plugin/pressure.cpp:115-133 */
static inline float thetaHelper(float inside, float outside) {
	const float denom = inside - outside;
	if (denom > -1e-04) return 0.5;
	float q = inside / denom;
	float m = q < 1.f ? q : 1.f; /* std::min(Real(1), q) */
	return 0.f < m ? m : 0.f;    /* std::max(Real(0), m) */
}
static inline float ghostFluidHelper(int64_t idx, int64_t offset, const float* phi, float gfClamp) {
	float alpha = thetaHelper(phi[idx], phi[idx + offset]);
	if (alpha < gfClamp) return gfClamp;
	return (float)(1. - (1. / alpha));
}
static inline float surfTensHelper(int64_t idx, int64_t offset, const float* phi, const float* curv, float surfTens,
                                   float gfClamp) {
	return surfTens * (curv[idx + offset] - ghostFluidHelper(idx, offset, phi, gfClamp) * curv[idx]);
}
static inline int ghostFluidWasClamped(int64_t idx, int64_t offset, const float* phi, float gfClamp) {
	const float alpha = thetaHelper(phi[idx], phi[idx + offset]);
	return alpha < gfClamp;
}

/* MakeRhs, plugin/pressure.cpp:32-84 (bnd=1, reduce +) */
int mf_make_rhs(int sx, int sy, int sz, const int32_t* flags, float* rhs, const float* vel, const float* perCellCorr,
                const float* fr, const float* ob, const float* phi, const float* curv, float surfTens, float gfClamp,
                int32_t* cnt_out, double* sum_out, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	const float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
	int nk = K1(d, 1) - K0(d, 1);
	if (nk < 0) nk = 0;
	double* psum = (double*)calloc(nk ? nk : 1, sizeof(double));
	int* pcnt = (int*)calloc(nk ? nk : 1, sizeof(int));
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++) {
		double sum = 0;
		int cnt = 0;
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) {
					rhs[idx] = 0;
					continue;
				}
				float set = 0;
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
				if (perCellCorr) set += perCellCorr[idx];
				sum += set;
				cnt++;
				rhs[idx] = set;
			}
		psum[k - K0(d, 1)] = sum;
		pcnt[k - K0(d, 1)] = cnt;
	}
	double sum = 0;
	int cnt = 0;
	for (int k = 0; k < nk; k++) {
		sum += psum[k];
		cnt += pcnt[k];
	}
	free(psum);
	free(pcnt);
	if (cnt_out) *cnt_out = cnt;
	if (sum_out) *sum_out = sum;
	return 0;
}

/* ApplyGhostFluidDiagonal, plugin/pressure.cpp:136-151 */
int mf_apply_ghost_fluid_diagonal(int sx, int sy, int sz, float* A0, const int32_t* flags, const float* phi,
                                  float gfClamp, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t X = d.X, Y = d.Y, Z = d.Z;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				if (flags[idx - X] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, -X, phi, gfClamp);
				if (flags[idx + X] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, +X, phi, gfClamp);
				if (flags[idx - Y] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, -Y, phi, gfClamp);
				if (flags[idx + Y] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, +Y, phi, gfClamp);
				if (d.is3d) {
					if (flags[idx - Z] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, -Z, phi, gfClamp);
					if (flags[idx + Z] & MF_EMPTY) A0[idx] -= ghostFluidHelper(idx, +Z, phi, gfClamp);
				}
			}
	return 0;
}

/* knCorrectVelocity, plugin/pressure.cpp:87-109 */
int mf_correct_velocity(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* p, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				int f = flags[idx];
				if (f & MF_FLUID) {
					if (flags[idx - X] & MF_FLUID) vx[idx] -= (p[idx] - p[idx - X]);
					if (flags[idx - Y] & MF_FLUID) vy[idx] -= (p[idx] - p[idx - Y]);
					if (d.is3d && (flags[idx - Z] & MF_FLUID)) vz[idx] -= (p[idx] - p[idx - Z]);
					if (flags[idx - X] & MF_EMPTY) vx[idx] -= p[idx];
					if (flags[idx - Y] & MF_EMPTY) vy[idx] -= p[idx];
					if (d.is3d && (flags[idx - Z] & MF_EMPTY)) vz[idx] -= p[idx];
				} else if ((f & MF_EMPTY) && !(f & MF_OUTFLOW)) {
					if (flags[idx - X] & MF_FLUID)
						vx[idx] += p[idx - X];
					else
						vx[idx] = 0.f;
					if (flags[idx - Y] & MF_FLUID)
						vy[idx] += p[idx - Y];
					else
						vy[idx] = 0.f;
					if (d.is3d) {
						if (flags[idx - Z] & MF_FLUID)
							vz[idx] += p[idx - Z];
						else
							vz[idx] = 0.f;
					}
				}
			}
	return 0;
}

/* knCorrectVelocityGhostFluid, plugin/pressure.cpp:154-187 */
int mf_correct_velocity_ghost_fluid(int sx, int sy, int sz, float* vel, const int32_t* flags, const float* p,
                                    const float* phi, float gfClamp, const float* curv, float surfTens, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				int f = flags[idx];
				int fl = (f & MF_FLUID) != 0, emp = (f & MF_EMPTY) && !(f & MF_OUTFLOW);
				if (fl) {
					if (flags[idx - X] & MF_EMPTY) vx[idx] += p[idx] * ghostFluidHelper(idx, -X, phi, gfClamp);
					if (flags[idx - Y] & MF_EMPTY) vy[idx] += p[idx] * ghostFluidHelper(idx, -Y, phi, gfClamp);
					if (d.is3d && (flags[idx - Z] & MF_EMPTY)) vz[idx] += p[idx] * ghostFluidHelper(idx, -Z, phi, gfClamp);
				} else if (emp) {
					if (flags[idx - X] & MF_FLUID)
						vx[idx] -= p[idx - X] * ghostFluidHelper(idx - X, +X, phi, gfClamp);
					else
						vx[idx] = 0.f;
					if (flags[idx - Y] & MF_FLUID)
						vy[idx] -= p[idx - Y] * ghostFluidHelper(idx - Y, +Y, phi, gfClamp);
					else
						vy[idx] = 0.f;
					if (d.is3d) {
						if (flags[idx - Z] & MF_FLUID)
							vz[idx] -= p[idx - Z] * ghostFluidHelper(idx - Z, +Z, phi, gfClamp);
						else
							vz[idx] = 0.f;
					}
				}
				if (curv) {
					if (fl) {
						if (flags[idx - X] & MF_EMPTY) vx[idx] += surfTensHelper(idx, -X, phi, curv, surfTens, gfClamp);
						if (flags[idx - Y] & MF_EMPTY) vy[idx] += surfTensHelper(idx, -Y, phi, curv, surfTens, gfClamp);
						if (d.is3d && (flags[idx - Z] & MF_EMPTY))
							vz[idx] += surfTensHelper(idx, -Z, phi, curv, surfTens, gfClamp);
					} else if (emp) {
						vx[idx] -= (flags[idx - X] & MF_FLUID) ? surfTensHelper(idx - X, +X, phi, curv, surfTens, gfClamp) : 0.f;
						vy[idx] -= (flags[idx - Y] & MF_FLUID) ? surfTensHelper(idx - Y, +Y, phi, curv, surfTens, gfClamp) : 0.f;
						if (d.is3d)
							vz[idx] -= (flags[idx - Z] & MF_FLUID) ? surfTensHelper(idx - Z, +Z, phi, curv, surfTens, gfClamp) : 0.f;
					}
				}
			}
	return 0;
}

/* knReplaceClampedGhostFluidVels, plugin/pressure.cpp:198-214.  The reference reads vel[idx+-X] while other
 * threads may write them; reads only touch fluid cells' entries, writes only empty cells' -> race free. */
int mf_replace_clamped_ghost_fluid_vels(int sx, int sy, int sz, float* vel, const int32_t* flags, const float* p,
                                        const float* phi, float gfClamp, void* st) {
	(void)st;
	(void)p;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_EMPTY)) continue;
				if ((flags[idx - X] & MF_FLUID) && ghostFluidWasClamped(idx - X, +X, phi, gfClamp)) vx[idx] = vx[idx - X];
				if ((flags[idx - Y] & MF_FLUID) && ghostFluidWasClamped(idx - Y, +Y, phi, gfClamp)) vy[idx] = vy[idx - Y];
				if (d.is3d && (flags[idx - Z] & MF_FLUID) && ghostFluidWasClamped(idx - Z, +Z, phi, gfClamp))
					vz[idx] = vz[idx - Z];
				if ((flags[idx + X] & MF_FLUID) && ghostFluidWasClamped(idx + X, -X, phi, gfClamp)) vx[idx] = vx[idx + X];
				if ((flags[idx + Y] & MF_FLUID) && ghostFluidWasClamped(idx + Y, -Y, phi, gfClamp)) vy[idx] = vy[idx + Y];
				if (d.is3d && (flags[idx + Z] & MF_FLUID) && ghostFluidWasClamped(idx + Z, -Z, phi, gfClamp))
					vz[idx] = vz[idx + Z];
			}
	return 0;
}

/* CountEmptyCells, plugin/pressure.cpp:217-220 */
int mf_count_empty_cells(int64_t n, const int32_t* flags, int32_t* r, void* st) {
	(void)st;
	int c = 0;
#pragma omp parallel for reduction(+ : c)
	for (int64_t i = 0; i < n; i++)
		if (flags[i] & MF_EMPTY) c++;
	*r = c;
	return 0;
}

/* fixPressure, plugin/pressure.cpp:226-246 */
int mf_fix_pressure(int sx, int sy, int sz, int64_t p, float value, float* rhs, float* A0, float* Ai, float* Aj,
                    float* Ak, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t X = d.X, Y = d.Y, Z = d.Z;
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
	return 0;
}

/* InitPreconditionModifiedIncompCholesky2, conjugategrad.cpp:66-97.  Serial FOR_IJK over ALL cells; only
 * fluid cells are touched (fluid cells never sit on the outer layer in valid scenes, so i-1 etc. exist). */
static inline float sq(float a) { return a * a; }
int mf_mic_init(int sx, int sy, int sz, const int32_t* flags, float* Ap, const float* A0, const float* Ai,
                const float* Aj, const float* Ak, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	const int64_t X = d.X, Y = d.Y, Z = d.Z;
	memset(Ap, 0, sizeof(float) * d.n);
	const float tau = 0.97;
	const float sigma = 0.25;
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				float e = A0[idx] - sq(Ai[idx - X] * Ap[idx - X]) - sq(Aj[idx - Y] * Ap[idx - Y]) - sq(Ak[idx - Z] * Ap[idx - Z]);
				/* `tau * ( a + b + c + 0. )` : the trailing double literal promotes the product and the
				 * subtraction to fp64 (conjugategrad.cpp:84-88) */
				float s3 = Ai[idx - X] * (Aj[idx - X] + Ak[idx - X]) * sq(Ap[idx - X]) +
				           Aj[idx - Y] * (Ai[idx - Y] + Ak[idx - Y]) * sq(Ap[idx - Y]) +
				           Ak[idx - Z] * (Ai[idx - Z] + Aj[idx - Z]) * sq(Ap[idx - Z]);
				e = (float)((double)e - (double)tau * ((double)s3 + 0.));
				if (e < sigma * A0[idx]) e = A0[idx];
				Ap[idx] = (float)(1. / (double)sqrtf(e));
			}
	return 0;
}

/* ApplyPreconditionModifiedIncompCholesky2, conjugategrad.cpp:135-159 (serial forward + backward sweep) */
int mf_mic_apply(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* v, const float* Ap,
                 const float* Ai, const float* Aj, const float* Ak, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	if (!d.is3d) return fail("mICP only supports 3D grids so far");
	const int64_t X = d.X, Y = d.Y, Z = d.Z;
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				const float p = Ap[idx];
				dst[idx] = p * (v[idx] - dst[idx - X] * Ai[idx - X] * Ap[idx - X] - dst[idx - Y] * Aj[idx - Y] * Ap[idx - Y] -
				                dst[idx - Z] * Ak[idx - Z] * Ap[idx - Z]);
			}
	for (int k = sz - 1; k >= 0; k--)
		for (int j = sy - 1; j >= 0; j--)
			for (int i = sx - 1; i >= 0; i--) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				const float p = Ap[idx];
				dst[idx] = p * (dst[idx] - dst[idx + X] * Ai[idx] * p - dst[idx + Y] * Aj[idx] * p - dst[idx + Z] * Ak[idx] * p);
			}
	return 0;
}

/* GridCg<APPLYMAT>::doInit / iterate, conjugategrad.cpp:210-299; loop as in solvePressureSystem
 * (plugin/pressure.cpp:438-441) */
int mf_cg_solve(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* rhs, float* residual,
                float* search, float* tmp, const float* A0, const float* Ai, const float* Aj, const float* Ak,
                float* Ap, int pc, float accuracy, int maxIter, int useL2Norm, float* out, void* st) {
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	if (pc != MF_PC_NONE && pc != MF_PC_MICP) return fail("GridCg<APPLYMAT>::setICPreconditioner: Invalid method specified.");
	if (pc == MF_PC_MICP && !d.is3d) pc = MF_PC_NONE; /* conjugategrad.cpp:315-321 */
	int iterations = 0;
	float resNorm = 1e20f, sigma = 0.f;
	if (maxIter <= 0) { /* GridCg::solve, conjugategrad.cpp:302-307: iterate() (and with it doInit) is never reached */
		out[0] = 0.f;
		out[1] = resNorm;
		out[2] = 0.f;
		return 0;
	}
	/* doInit */
	memset(dst, 0, sizeof(float) * n);
	memcpy(residual, rhs, sizeof(float) * n);
	if (pc == MF_PC_MICP) {
		mf_mic_init(sx, sy, sz, flags, Ap, A0, Ai, Aj, Ak, st);
		mf_mic_apply(sx, sy, sz, flags, tmp, residual, Ap, Ai, Aj, Ak, st);
	} else
		memcpy(tmp, residual, sizeof(float) * n);
	memcpy(search, tmp, sizeof(float) * n);
	sigma = (float)dot64(n, tmp, residual);
	for (int iter = 0; iter < maxIter; iter++) {
		iterations++;
		mf_apply_matrix(sx, sy, sz, flags, tmp, search, A0, Ai, Aj, Ak, st);
		float dp = (float)dot64(n, tmp, search);
		float alpha = 0.;
		if (fabs(dp) > 0.) alpha = sigma / (float)dp;
		mf_grid_scaled_add(n, dst, search, alpha, st);
		mf_grid_scaled_add(n, residual, tmp, -alpha, st);
		if (pc == MF_PC_MICP)
			mf_mic_apply(sx, sy, sz, flags, tmp, residual, Ap, Ai, Aj, Ak, st);
		else
			memcpy(tmp, residual, sizeof(float) * n);
		if (useL2Norm)
			resNorm = (float)sumsqr64(n, residual);
		else
			resNorm = maxabs(n, residual);
		if (resNorm < accuracy) {
			sigma = resNorm;
			break;
		}
		float sigmaNew = (float)dot64(n, tmp, residual);
		float beta = sigmaNew / sigma;
		mf_update_search_vec(n, search, tmp, beta, st);
		sigma = sigmaNew;
		if (!(resNorm < 1e35)) {
			out[0] = (float)iterations;
			out[1] = resNorm;
			out[2] = sigma;
			return fail("GridCg::iterate: The CG solver diverged, residual norm > 1e30, stopping.");
		}
	}
	out[0] = (float)iterations;
	out[1] = resNorm;
	out[2] = sigma;
	return 0;
}

/* ================================================================================================
 * interpolation primitives, util/interpol.h
 * ============================================================================================== */
/* global plane index -> index inside the slab window, kept addressable in the ghost fringe; identity by default */
static inline int local_z(const Dim* d, int zi, int hi_off) {
	if (d->sz <= 1) return zi;
	zi -= d->zoff;
	int hi = d->sz - 1 - hi_off;
	return zi < 0 ? 0 : (zi > hi ? hi : zi);
}
typedef struct {
	int xi, yi, zi;
	float s0, s1, t0, t1, f0, f1;
} Bi;
/* BUILD_INDEX, interpol.h:52-69 (note: the fork's upper clamp tests px, not xi) */
static inline Bi build_index(const Dim* d, float x, float y, float z) {
	Bi b;
	float px = x - 0.5f, py = y - 0.5f, pz = z - 0.5f;
	b.xi = (int)px;
	b.yi = (int)py;
	b.zi = (int)pz;
	b.s1 = px - (float)b.xi;
	b.s0 = 1. - b.s1;
	b.t1 = py - (float)b.yi;
	b.t0 = 1. - b.t1;
	b.f1 = pz - (float)b.zi;
	b.f0 = 1. - b.f1;
	if (px < 0.) { b.xi = 0; b.s0 = 1.0; b.s1 = 0.0; }
	if (py < 0.) { b.yi = 0; b.t0 = 1.0; b.t1 = 0.0; }
	if (pz < 0.) { b.zi = 0; b.f0 = 1.0; b.f1 = 0.0; }
	if (px >= d->sx - 1) { b.xi = d->sx - 2; b.s0 = 0.0; b.s1 = 1.0; }
	if (py >= d->sy - 1) { b.yi = d->sy - 2; b.t0 = 0.0; b.t1 = 1.0; }
	if (d->gsz > 1) { if (pz >= d->gsz - 1) { b.zi = d->gsz - 2; b.f0 = 0.0; b.f1 = 1.0; } }
	b.zi = local_z(d, b.zi, 1);
	return b;
}
/* the shifted half of BUILD_INDEX_SHIFT, interpol.h:116-129 (upper clamp tests the integer index) */
static inline Bi build_index_shift(const Dim* d, float x, float y, float z) {
	Bi b;
	b.xi = (int)x;
	b.yi = (int)y;
	b.zi = (int)z;
	b.s1 = x - (float)b.xi;
	b.s0 = 1. - b.s1;
	b.t1 = y - (float)b.yi;
	b.t0 = 1. - b.t1;
	b.f1 = z - (float)b.zi;
	b.f0 = 1. - b.f1;
	if (x < 0) { b.xi = 0; b.s0 = 1.0; b.s1 = 0.0; }
	if (y < 0) { b.yi = 0; b.t0 = 1.0; b.t1 = 0.0; }
	if (z < 0) { b.zi = 0; b.f0 = 1.0; b.f1 = 0.0; }
	if (b.xi >= d->sx - 1) { b.xi = d->sx - 2; b.s0 = 0.0; b.s1 = 1.0; }
	if (b.yi >= d->sy - 1) { b.yi = d->sy - 2; b.t0 = 0.0; b.t1 = 1.0; }
	if (d->gsz > 1) { if (b.zi >= d->gsz - 1) { b.zi = d->gsz - 2; b.f0 = 0.0; b.f1 = 1.0; } }
	b.zi = local_z(d, b.zi, 1);
	return b;
}
/* interpol<T> / interpolComponent<c>, interpol.h:71-94 on one scalar plane */
static inline float interpol1(const Dim* d, const float* data, float x, float y, float z) {
	Bi b = build_index(d, x, y, z);
	const int64_t X = 1, Y = d->Y, Z = d->Z;
	int64_t idx = (int64_t)b.xi + Y * b.yi + Z * b.zi;
	return ((data[idx] * b.t0 + data[idx + Y] * b.t1) * b.s0 + (data[idx + X] * b.t0 + data[idx + X + Y] * b.t1) * b.s1) * b.f0 +
	       ((data[idx + Z] * b.t0 + data[idx + Y + Z] * b.t1) * b.s0 + (data[idx + X + Z] * b.t0 + data[idx + X + Y + Z] * b.t1) * b.s1) * b.f1;
}
/* ---- cubic interpolation, util/interpolHigh.h ------------------------------------------------------------------
 * cubicInterp<T> :22-39.  vec = 0: T = Real -- a2 and a3 are double expressions (3.0 * deltak - 2.0 * d0 - d1) rounded once;
 * vec = 1: one component of T = Vec3 -- every scalar * vector product is rounded to fp32 (vectorbase.h:277-284) before the fp32
 * vector sums.  The polynomial itself is fp32 either way, left to right. */
static inline float cubic_interp(float t, const float p[4], int vec) {
	const float d0 = (float)((double)(p[2] - p[0]) * 0.5), d1 = (float)((double)(p[3] - p[1]) * 0.5);
	const float dk = p[2] - p[1];
	float a2, a3;
	if (!vec) {
		a2 = (float)(3.0 * (double)dk - 2.0 * (double)d0 - (double)d1);
		a3 = (float)(-2.0 * (double)dk + (double)d0 + (double)d1);
	} else {
		a2 = ((float)(3.0 * (double)dk) - (float)(2.0 * (double)d0)) - d1;
		a3 = ((float)(-2.0 * (double)dk) + d0) + d1;
	}
	const float sq = t * t, cu = sq * t;
	return a3 * cu + a2 * sq + d0 * t + p[1];
}
/* interpolCubic<T> / interpolCubic2D<T> :42-167 on one scalar plane: 4 x 4 (x 4) points around the cell of pos - 0.5; where that
 * neighbourhood leaves the grid the reference falls back to the linear interpol().  Positions are global (z-slab window). */
static inline float interpol_cubic(const Dim* d, const float* data, float x, float y, float z, int vec) {
	const float px = x - 0.5f, py = y - 0.5f, pz = z - 0.5f;
	const int x1 = (int)px, y1 = (int)py, z1 = (int)pz;
	const int x0 = x1 - 1, x3 = x1 + 2, y0 = y1 - 1, y3 = y1 + 2, z0 = z1 - 1, z3 = z1 + 2;
	if (x0 < 0 || y0 < 0 || x3 >= d->sx || y3 >= d->sy || (d->is3d && (z0 < 0 || z3 >= d->gsz))) return interpol1(d, data, x, y, d->is3d ? z : z);
	const float xi = px - (float)x1, yi = py - (float)y1, zi = pz - (float)z1;
	float planes[4];
	const int nzp = d->is3d ? 4 : 1;
	for (int c = 0; c < nzp; c++) {
		const int64_t zb = d->is3d ? d->Z * (int64_t)(z0 + c - d->zoff) : 0;
		float rows[4];
		for (int b = 0; b < 4; b++) {
			const float* r = data + zb + d->Y * (int64_t)(y0 + b) + x0;
			const float q[4] = {r[0], r[1], r[2], r[3]};
			rows[b] = cubic_interp(xi, q, vec);
		}
		planes[c] = cubic_interp(yi, rows, vec);
	}
	return d->is3d ? cubic_interp(zi, planes, vec) : planes[0];
}
/* interpolCubicMAC :169-176, component c: interpolCubic<Vec3>(pos + 0.5 e_c)[c]; 0 for the z component of a 2-D grid */
static inline float interpol_cubic_mac(const Dim* d, const float* vel, int c, float x, float y, float z) {
	if (c == 2 && !d->is3d) return 0.f;
	return interpol_cubic(d, vel + (int64_t)c * d->n, c == 0 ? x + 0.5f : x, c == 1 ? y + 0.5f : y, c == 2 ? z + 0.5f : z, 1);
}
/* interpolMAC, interpol.h:131-164 */
static inline void interpol_mac(const Dim* d, const float* vel, float x, float y, float z, float out[3]) {
	Bi b = build_index(d, x, y, z), s = build_index_shift(d, x, y, z);
	const int64_t X = 1, Y = d->Y, Z = d->Z, n = d->n;
	{
		const float* r = vel + (((int64_t)b.zi * d->sy + b.yi) * d->sx + s.xi);
		out[0] = b.f0 * ((r[0] * b.t0 + r[Y] * b.t1) * s.s0 + (r[X] * b.t0 + r[X + Y] * b.t1) * s.s1) +
		         b.f1 * ((r[Z] * b.t0 + r[Z + Y] * b.t1) * s.s0 + (r[X + Z] * b.t0 + r[X + Y + Z] * b.t1) * s.s1);
	}
	{
		const float* r = vel + n + (((int64_t)b.zi * d->sy + s.yi) * d->sx + b.xi);
		out[1] = b.f0 * ((r[0] * s.t0 + r[Y] * s.t1) * b.s0 + (r[X] * s.t0 + r[X + Y] * s.t1) * b.s1) +
		         b.f1 * ((r[Z] * s.t0 + r[Z + Y] * s.t1) * b.s0 + (r[X + Z] * s.t0 + r[X + Y + Z] * s.t1) * b.s1);
	}
	{
		const float* r = vel + 2 * n + (((int64_t)s.zi * d->sy + b.yi) * d->sx + b.xi);
		out[2] = s.f0 * ((r[0] * b.t0 + r[Y] * b.t1) * b.s0 + (r[X] * b.t0 + r[X + Y] * b.t1) * b.s1) +
		         s.f1 * ((r[Z] * b.t0 + r[Z + Y] * b.t1) * b.s0 + (r[X + Z] * b.t0 + r[X + Y + Z] * b.t1) * b.s1);
	}
}
/* MACGrid::getCentered / getAtMACX/Y/Z, grid.h:460-506 */
static inline void get_centered(const Dim* d, const float* vel, int64_t idx, float v[3]) {
	const int64_t n = d->n;
	v[0] = 0.5 * (vel[idx] + vel[idx + 1]);
	v[1] = 0.5 * (vel[n + idx] + vel[n + idx + d->sx]);
	v[2] = 0.;
	if (d->is3d) v[2] = 0.5 * (vel[2 * n + idx] + vel[2 * n + idx + d->Z]);
}
static inline void get_at_mac_x(const Dim* d, const float* vel, int64_t idx, float v[3]) {
	const int64_t n = d->n, sx = d->sx, Z = d->Z;
	const float *y = vel + n, *z = vel + 2 * n;
	v[0] = vel[idx];
	v[1] = 0.25 * (y[idx] + y[idx - 1] + y[idx + sx] + y[idx + sx - 1]);
	v[2] = 0.;
	if (d->is3d) v[2] = 0.25 * (z[idx] + z[idx - 1] + z[idx + Z] + z[idx + Z - 1]);
}
static inline void get_at_mac_y(const Dim* d, const float* vel, int64_t idx, float v[3]) {
	const int64_t n = d->n, sx = d->sx, Z = d->Z;
	const float *x = vel, *z = vel + 2 * n;
	v[0] = 0.25 * (x[idx] + x[idx - sx] + x[idx + 1] + x[idx + 1 - sx]);
	v[1] = vel[n + idx];
	v[2] = 0.;
	if (d->is3d) v[2] = 0.25 * (z[idx] + z[idx - sx] + z[idx + Z] + z[idx + Z - sx]);
}
static inline void get_at_mac_z(const Dim* d, const float* vel, int64_t idx, float v[3]) {
	const int64_t n = d->n, sx = d->sx, Z = d->Z;
	const float *x = vel, *y = vel + n;
	v[0] = 0.25 * (x[idx] + x[idx - Z] + x[idx + 1] + x[idx + 1 - Z]);
	v[1] = 0.25 * (y[idx] + y[idx - Z] + y[idx + sx] + y[idx + sx - Z]);
	v[2] = vel[2 * n + idx];
}

/* ================================================================================================
 * advection, plugin/advection.cpp
 * ============================================================================================== */
/* SemiLagrange<T>, advection.cpp:25-42; ncomp scalar planes of `src` (1 = Real, 3 = centred Vec3) */
static int semi_lagrange(int sx, int sy, int sz, int ncomp, const float* vel, float* dst, const float* src, float dt,
                         int orderTrace, int orderSpace) {
	Dim d = mkdim(sx, sy, sz);
	if (orderTrace != 1 && orderTrace != 2) return fail("Unknown backtracing order");
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order");
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				float v[3], px, py, pz;
				const int kg = k + d.zoff;
				get_centered(&d, vel, idx, v);
				if (orderTrace == 1) {
					px = (i + 0.5f) - v[0] * dt;
					py = (j + 0.5f) - v[1] * dt;
					pz = (kg + 0.5f) - v[2] * dt;
				} else {
					float p1x = (i + 0.5f) - (float)((v[0] * dt) * 0.5);
					float p1y = (j + 0.5f) - (float)((v[1] * dt) * 0.5);
					float p1z = (kg + 0.5f) - (float)((v[2] * dt) * 0.5);
					float u[3];
					interpol_mac(&d, vel, p1x, p1y, p1z, u);
					px = (i + 0.5f) - u[0] * dt;
					py = (j + 0.5f) - u[1] * dt;
					pz = (kg + 0.5f) - u[2] * dt;
				}
				for (int c = 0; c < ncomp; c++)
					dst[c * d.n + idx] = orderSpace == 1 ? interpol1(&d, src + c * d.n, px, py, pz)
					                                     : interpol_cubic(&d, src + c * d.n, px, py, pz, ncomp == 3);   /* getInterpolatedHi, grid.h:153-159 */
			}
	return 0;
}
int mf_semi_lagrange_real(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                          int orderTrace, int orderSpace, void* st) {
	(void)st;
	memset(dst, 0, sizeof(float) * (size_t)sx * sy * sz);      /* the border: zeros of the reference's fresh temp grid */
	return semi_lagrange(sx, sy, sz, 1, vel, dst, src, dt, orderTrace, orderSpace);
}
int mf_semi_lagrange_vec3(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                          int orderTrace, int orderSpace, void* st) {
	(void)st;
	memset(dst, 0, sizeof(float) * 3 * (size_t)sx * sy * sz);
	return semi_lagrange(sx, sy, sz, 3, vel, dst, src, dt, orderTrace, orderSpace);
}
/* SemiLagrangeMAC, advection.cpp:45-78 */
/* MACGrid::getInterpolatedComponentHi<c>, grid.h:280-286 */
static inline float mac_component_hi(const Dim* d, const float* src, int c, float x, float y, float z, int orderSpace) {
	return orderSpace == 1 ? interpol1(d, src + (int64_t)c * d->n, x, y, z) : interpol_cubic_mac(d, src, c, x, y, z);
}
int mf_semi_lagrange_mac(int sx, int sy, int sz, const float* vel, float* dst, const float* src, float dt,
                         int orderTrace, int orderSpace, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	if (orderTrace != 1 && orderTrace != 2) return fail("Unknown backtracing order");
	if (orderSpace != 1 && orderSpace != 2) return fail("Unknown interpolation order");
	memset(dst, 0, sizeof(float) * 3 * (size_t)n);      /* the border: zeros of the reference's fresh temp grid */
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				const int kg = k + d.zoff;
				float v[3], r[3];
				if (orderTrace == 1) {
					get_at_mac_x(&d, vel, idx, v);
					r[0] = mac_component_hi(&d, src, 0, (i + 0.5f) - v[0] * dt, (j + 0.5f) - v[1] * dt, (kg + 0.5f) - v[2] * dt, orderSpace);
					get_at_mac_y(&d, vel, idx, v);
					r[1] = mac_component_hi(&d, src, 1, (i + 0.5f) - v[0] * dt, (j + 0.5f) - v[1] * dt, (kg + 0.5f) - v[2] * dt, orderSpace);
					get_at_mac_z(&d, vel, idx, v);
					r[2] = mac_component_hi(&d, src, 2, (i + 0.5f) - v[0] * dt, (j + 0.5f) - v[1] * dt, (kg + 0.5f) - v[2] * dt, orderSpace);
				} else {
					float u[3];
					const float p0x = (float)(i + 0.5), p0y = (float)(j + 0.5), p0z = (float)(kg + 0.5);
					get_at_mac_x(&d, src, idx, v);
					interpol_mac(&d, src, (float)i - (float)((v[0] * dt) * 0.5), (j + 0.5f) - (float)((v[1] * dt) * 0.5),
					             (kg + 0.5f) - (float)((v[2] * dt) * 0.5), u);
					r[0] = mac_component_hi(&d, src, 0, p0x - u[0] * dt, p0y - u[1] * dt, p0z - u[2] * dt, orderSpace);
					get_at_mac_y(&d, src, idx, v);
					interpol_mac(&d, src, (i + 0.5f) - (float)((v[0] * dt) * 0.5), (float)j - (float)((v[1] * dt) * 0.5),
					             (kg + 0.5f) - (float)((v[2] * dt) * 0.5), u);
					r[1] = mac_component_hi(&d, src, 1, p0x - u[0] * dt, p0y - u[1] * dt, p0z - u[2] * dt, orderSpace);
					get_at_mac_z(&d, src, idx, v);
					interpol_mac(&d, src, (i + 0.5f) - (float)((v[0] * dt) * 0.5), (j + 0.5f) - (float)((v[1] * dt) * 0.5),
					             (float)kg - (float)((v[2] * dt) * 0.5), u);
					r[2] = mac_component_hi(&d, src, 2, p0x - u[0] * dt, p0y - u[1] * dt, p0z - u[2] * dt, orderSpace);
				}
				dst[idx] = r[0];
				dst[n + idx] = r[1];
				dst[2 * n + idx] = r[2];
			}
	return 0;
}

/* MacCormackCorrect<T>, advection.cpp:82-92 (KERNEL(idx)).
 * Real:  dst += strength*0.5*(old-bwd)  -> the compound assignment is evaluated in fp64 (0.5 is a double).
 * Vec3:  S2*Vector3D rounds each product to fp32 first (vectorbase.h:282-284), then an fp32 add. */
int mf_maccormack_correct(int sx, int sy, int sz, int ncomp, const int32_t* flags, float* dst, const float* old,
                          const float* fwd, const float* bwd, float strength, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
#pragma omp parallel for
	for (int64_t idx = 0; idx < n; idx++) {
		int fl = flags[idx] & MF_FLUID;
		for (int c = 0; c < ncomp; c++) {
			int64_t q = c * n + idx;
			float v = fwd[q];
			if (fl) {
				if (ncomp == 1)
					v = (float)((double)v + strength * 0.5 * (old[q] - bwd[q]));
				else
					v = v + (float)(strength * 0.5 * (old[q] - bwd[q]));
			}
			dst[q] = v;
		}
	}
	return 0;
}
/* MacCormackCorrectMAC<Vec3>(isMAC=true), advection.cpp:95-116 (KERNEL(): all cells) */
int mf_maccormack_correct_mac(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* old,
                              const float* fwd, const float* bwd, float strength, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
#pragma omp parallel for
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				int skip[3] = {0, 0, 0};
				if (!(flags[idx] & MF_FLUID)) skip[0] = skip[1] = skip[2] = 1;
				if ((i > 0) && !(flags[idx - d.X] & MF_FLUID)) skip[0] = 1;
				if ((j > 0) && !(flags[idx - d.Y] & MF_FLUID)) skip[1] = 1;
				if ((k > 0) && !(flags[idx - d.Z] & MF_FLUID)) skip[2] = 1;
				for (int c = 0; c < 3; c++) {
					int64_t q = c * n + idx;
					if (skip[c])
						dst[q] = fwd[q];
					else
						dst[q] = (float)(fwd[q] + strength * 0.5 * (old[q] - bwd[q]));
				}
			}
	return 0;
}

#define CHECKFLAG(f) ((f) & (MF_FLUID | MF_EMPTY)) /* advection.cpp:140 */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* doClampComponent<T>, advection.cpp:145-187 (gridSize argument already holds size-1) */
static void do_clamp_component(const Dim* d, int ncomp, const int32_t* flags, float* dval, const float* orig,
                               const float* fwdv, float px, float py, float pz, const float vel[3], int clampMode) {
	float minv[3], maxv[3];
	for (int c = 0; c < ncomp; c++) {
		minv[c] = FLT_MAX;
		maxv[c] = -FLT_MAX;
	}
	int haveFl = 0;
	int pos[2][3];
	int numPos = 1;
	pos[0][0] = (int)(px - vel[0]);
	pos[0][1] = (int)(py - vel[1]);
	pos[0][2] = (int)(pz - vel[2]);
	if (clampMode == 1) {
		numPos = 2;
		pos[1][0] = (int)(px + vel[0]);
		pos[1][1] = (int)(py + vel[1]);
		pos[1][2] = (int)(pz + vel[2]);
	}
	const int gx = d->sx - 1, gy = d->sy - 1, gz = d->gsz - 1;
	for (int l = 0; l < numPos; l++) {
		const int i0 = clampi(pos[l][0], 0, gx - 1);
		const int j0 = clampi(pos[l][1], 0, gy - 1);
		const int k0 = local_z(d, clampi(pos[l][2], 0, d->is3d ? (gz - 1) : 1), 1);
		const int i1 = i0 + 1, j1 = j0 + 1, k1 = d->is3d ? (k0 + 1) : k0;
		const int ii[8] = {i0, i1, i0, i1, i0, i1, i0, i1};
		const int jj[8] = {j0, j0, j1, j1, j0, j0, j1, j1};
		const int kk[8] = {k0, k0, k0, k0, k1, k1, k1, k1};
		int nc = d->is3d ? 8 : 4;
		for (int q = 0; q < nc; q++) {
			int64_t idx = IDX(*d, ii[q], jj[q], kk[q]);
			if (CHECKFLAG(flags[idx])) {
				for (int c = 0; c < ncomp; c++) {
					float v = orig[c * d->n + idx];
					if (v < minv[c]) minv[c] = v;
					if (v > maxv[c]) maxv[c] = v;
				}
				haveFl = 1;
			}
		}
	}
	if (!haveFl) {
		for (int c = 0; c < ncomp; c++) dval[c] = fwdv[c];
		return;
	}
	if (clampMode == 1) {
		for (int c = 0; c < ncomp; c++) {
			float v = dval[c];
			dval[c] = v < minv[c] ? minv[c] : (v > maxv[c] ? maxv[c] : v);
		}
	} else {
		int outside = 0;
		for (int c = 0; c < ncomp; c++) outside |= (dval[c] < minv[c]) | (dval[c] > maxv[c]);
		if (outside)
			for (int c = 0; c < ncomp; c++) dval[c] = fwdv[c];
	}
}
/* MacCormackClamp<T>, advection.cpp:242-268 */
int mf_maccormack_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst,
                        const float* orig, const float* fwd, float dt, int clampMode, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	const int gx = sx - 1, gy = sy - 1, gz = d.gsz - 1;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				const int kg = k + d.zoff;
				float dval[3], fw[3], v[3], vd[3];
				for (int c = 0; c < ncomp; c++) {
					dval[c] = dst[c * n + idx];
					fw[c] = fwd[c * n + idx];
				}
				get_centered(&d, vel, idx, v);
				vd[0] = v[0] * dt;
				vd[1] = v[1] * dt;
				vd[2] = v[2] * dt;
				do_clamp_component(&d, ncomp, flags, dval, orig, fw, (float)i, (float)j, (float)kg, vd, clampMode);
				if (clampMode == 1) {
					/* Vec3(i,j,k) + Vec3(0.5,0.5,0.5) -+ vel*dt, truncated */
					int fx = (int)(((float)i + 0.5f) - vd[0]), fy = (int)(((float)j + 0.5f) - vd[1]), fz = (int)(((float)kg + 0.5f) - vd[2]);
					int bx = (int)(((float)i + 0.5f) + vd[0]), by = (int)(((float)j + 0.5f) + vd[1]), bz = (int)(((float)kg + 0.5f) + vd[2]);
					int bad = fx < 0 || fy < 0 || fz < 0 || bx < 0 || by < 0 || bz < 0 || fx > gx || fy > gy ||
					          ((fz > gz) && d.is3d) || bx > gx || by > gy || ((bz > gz) && d.is3d);
					if (!bad) bad = (flags[IDX(d, fx, fy, local_z(&d, fz, 0))] & MF_OBSTACLE) || (flags[IDX(d, bx, by, local_z(&d, bz, 0))] & MF_OBSTACLE);
					if (bad)
						for (int c = 0; c < ncomp; c++) dval[c] = fw[c];
				}
				for (int c = 0; c < ncomp; c++) dst[c * n + idx] = dval[c];
			}
	return 0;
}
/* doClampComponentMAC<c>, advection.cpp:192-236 */
static float do_clamp_component_mac(const Dim* d, int c, const int32_t* flags, float dst, const float* orig, float fwd,
                                    int i, int j, int k, const float vel[3], int clampMode) {
	float minv = FLT_MAX, maxv = -FLT_MAX;
	const float px = (float)i, py = (float)j, pz = (float)(k + d->zoff);
	int pos[2][3];
	int numPos = 1;
	pos[0][0] = (int)(px - vel[0]);
	pos[0][1] = (int)(py - vel[1]);
	pos[0][2] = (int)(pz - vel[2]);
	if (clampMode == 1) {
		numPos = 2;
		pos[1][0] = (int)(px + vel[0]);
		pos[1][1] = (int)(py + vel[1]);
		pos[1][2] = (int)(pz + vel[2]);
	}
	int o[3] = {i, j, k}, nb[3] = {i, j, k};
	nb[c] -= 1;
	if (clampMode == 2 && !(CHECKFLAG(flags[IDX(*d, o[0], o[1], o[2])]) && CHECKFLAG(flags[IDX(*d, nb[0], nb[1], nb[2])])))
		return fwd;
	const int gx = d->sx - 1, gy = d->sy - 1, gz = d->gsz - 1;
	const float* oc = orig + c * d->n;
	for (int l = 0; l < numPos; l++) {
		const int i0 = clampi(pos[l][0], 0, gx - 1);
		const int j0 = clampi(pos[l][1], 0, gy - 1);
		const int k0 = local_z(d, clampi(pos[l][2], 0, d->is3d ? (gz - 1) : 0), 1);
		const int i1 = i0 + 1, j1 = j0 + 1, k1 = d->is3d ? (k0 + 1) : k0;
		const int ii[8] = {i0, i1, i0, i1, i0, i1, i0, i1};
		const int jj[8] = {j0, j0, j1, j1, j0, j0, j1, j1};
		const int kk[8] = {k0, k0, k0, k0, k1, k1, k1, k1};
		int nc = d->is3d ? 8 : 4;
		for (int q = 0; q < nc; q++) {
			float v = oc[IDX(*d, ii[q], jj[q], kk[q])];
			if (v < minv) minv = v;
			if (v > maxv) maxv = v;
		}
	}
	if (clampMode == 1)
		dst = dst < minv ? minv : (dst > maxv ? maxv : dst);
	else if ((dst < minv) | (dst > maxv))
		dst = fwd;
	return dst;
}
/* MacCormackClampMAC, advection.cpp:271-288 */
int mf_maccormack_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst,
                            const float* orig, const float* fwd, float dt, int clampMode, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				float v[3], vd[3];
				get_at_mac_x(&d, vel, idx, v);
				vd[0] = v[0] * dt; vd[1] = v[1] * dt; vd[2] = v[2] * dt;
				float rx = do_clamp_component_mac(&d, 0, flags, dst[idx], orig, fwd[idx], i, j, k, vd, clampMode);
				get_at_mac_y(&d, vel, idx, v);
				vd[0] = v[0] * dt; vd[1] = v[1] * dt; vd[2] = v[2] * dt;
				float ry = do_clamp_component_mac(&d, 1, flags, dst[n + idx], orig, fwd[n + idx], i, j, k, vd, clampMode);
				float rz = dst[2 * n + idx];
				if (d.is3d) {
					get_at_mac_z(&d, vel, idx, v);
					vd[0] = v[0] * dt; vd[1] = v[1] * dt; vd[2] = v[2] * dt;
					rz = do_clamp_component_mac(&d, 2, flags, rz, orig, fwd[2 * n + idx], i, j, k, vd, clampMode);
				}
				dst[idx] = rx;
				dst[n + idx] = ry;
				dst[2 * n + idx] = rz;
			}
	return 0;
}

/* applyOutflowBC, advection.cpp:327-392 */
static int in_bounds(const Dim* d, int i, int j, int k) { return i >= 0 && j >= 0 && k >= 0 && i < d->sx && j < d->sy && k < d->sz; }
/* Particle positions are GLOBAL grid coordinates; with a z-slab window (mf_set_slab_window: multi-GPU tests) the grid holds
 * planes [zoff, zoff + sz) of gsz.  cell_of: cell of a position, k as the plane inside the window; 0 when the cell is outside
 * the domain or the window.  Without a window this is isInBounds(toVec3i(pos)). */
static int cell_of(const Dim* d, float x, float y, float z, int* i, int* j, int* k) {
	*i = (int)x;
	*j = (int)y;
	const int kg = (int)z;
	*k = kg - d->zoff;
	return *i >= 0 && *j >= 0 && kg >= 0 && *i < d->sx && *j < d->sy && kg < d->gsz && *k >= 0 && *k < d->sz;
}
/* domain-boundary test in global planes (a slab's outer ghost plane is not a domain wall) */
static int z_wall(const Dim* d, int k, int w) { return d->is3d && (k + d->zoff <= w || k + d->zoff >= d->gsz - 1 - w); }
/* the fused entry points of the HIP library: here simply the two reference kernels in sequence */
int mf_maccormack_correct_clamp(int sx, int sy, int sz, int ncomp, const int32_t* flags, const float* vel, float* dst, const float* orig,
                                const float* fwd, const float* bwd, float strength, float dt, int clampMode, void* st) {
	int rc = mf_maccormack_correct(sx, sy, sz, ncomp, flags, dst, orig, fwd, bwd, strength, st);
	if (rc) return rc;
	return mf_maccormack_clamp(sx, sy, sz, ncomp, flags, vel, dst, orig, fwd, dt, clampMode, st);
}
int mf_maccormack_correct_clamp_mac(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* dst, const float* orig,
                                    const float* fwd, const float* bwd, float strength, float dt, int clampMode, void* st) {
	int rc = mf_maccormack_correct_mac(sx, sy, sz, flags, dst, orig, fwd, bwd, strength, st);
	if (rc) return rc;
	return mf_maccormack_clamp_mac(sx, sy, sz, flags, vel, dst, orig, fwd, dt, clampMode, st);
}
int mf_apply_outflow_bc(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* velPrev, float* velDst,
                        float dtIn, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	/* applyOutflowBC takes double timeStep and passes max(1.0, timeStep*4) as Real */
	const float timeStep = (float)((1.0 > (double)dtIn * 4) ? 1.0 : (double)dtIn * 4);
#pragma omp parallel for
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_OUTFLOW)) continue;
				/* getBulkVel, advection.cpp:327-344 */
				float avg[3] = {0.f, 0.f, 0.f};
				int count = 0;
				int nmax = d.is3d ? 1 : 0;
				for (int nn = -nmax; nn <= nmax; nn++)
					for (int m = -1; m <= 1; m++)
						for (int l = -1; l <= 1; l++)
							if (in_bounds(&d, i + l, j + m, k + nn)) {
								int64_t q = IDX(d, i + l, j + m, k + nn);
								if (flags[q] & (MF_FLUID | MF_OUTFLOW)) {
									avg[0] += vel[q];
									avg[1] += vel[n + q];
									avg[2] += vel[2 * n + q];
									count++;
								}
							}
				if (count > 0) {
					avg[0] = avg[0] / count;
					avg[1] = avg[1] / count;
					avg[2] = avg[2] / count;
				}
				int dim = d.is3d ? 3 : 2;
				int cnt = 0;
				float acc[3] = {velDst[idx], velDst[n + idx], velDst[2 * n + idx]};
				for (int c = 0; c < dim; c++) {
					int low[3] = {i, j, k}, up[3] = {i, j, k}, flLow[3] = {i, j, k}, flUp[3] = {i, j, k};
					float factor = timeStep * ((float)1.0 > avg[c] ? (float)1.0 : avg[c]);
					low[c] = flLow[c] = low[c] - 1;
					up[c] = flUp[c] = up[c] + 1;
					for (int dd = 0; dd < 2; dd++) {
						int eL = in_bounds(&d, flLow[0], flLow[1], flLow[2]) && (flags[IDX(d, flLow[0], flLow[1], flLow[2])] & MF_FLUID);
						int eU = in_bounds(&d, flUp[0], flUp[1], flUp[2]) && (flags[IDX(d, flUp[0], flUp[1], flUp[2])] & MF_FLUID);
						if (eL || eU) {
							if (eL) {
								int64_t q = IDX(d, low[0], low[1], low[2]);
								for (int e = 0; e < 3; e++) acc[e] += ((vel[e * n + idx] - velPrev[e * n + idx]) / factor) + vel[e * n + q];
								cnt++;
							}
							if (eU) {
								int64_t q = IDX(d, up[0], up[1], up[2]);
								for (int e = 0; e < 3; e++) acc[e] += ((vel[e * n + idx] - velPrev[e * n + idx]) / factor) + vel[e * n + q];
								cnt++;
							}
							break;
						}
						flLow[c]--;
						flUp[c]++;
					}
				}
				if (cnt > 0) {
					float fc = (float)cnt;
					acc[0] /= fc;
					acc[1] /= fc;
					acc[2] /= fc;
				}
				velDst[idx] = acc[0];
				velDst[n + idx] = acc[1];
				velDst[2 * n + idx] = acc[2];
			}
	/* copyChangedVels, advection.cpp:385 */
#pragma omp parallel for
	for (int64_t idx = 0; idx < n; idx++)
		if (flags[idx] & MF_OUTFLOW) {
			vel[idx] = velDst[idx];
			vel[n + idx] = velDst[n + idx];
			vel[2 * n + idx] = velDst[2 * n + idx];
		}
	return 0;
}

/* ================================================================================================
 * FLIP transfers
 * ============================================================================================== */
static inline int skip_particle(const int32_t* pflag, const int32_t* ptype, int exclude, int64_t p) {
	/* !p.isActive(idx) || (ptype && ((*ptype)[idx] & exclude)), flip.cpp:630 */
	return (pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude));
}
/* setInterpolMAC, interpol.h:166-213: one component plane with its own base index and weights */
static inline void scatter8(float* ref, float* sum, int64_t Y, int64_t Z, float ta, float tb, float sa, float sb,
                            float fa, float fb, float val, int zfirst) {
	const int64_t X = 1;
	float s0f0 = sa * fa, s1f0 = sb * fa, s0f1 = sa * fb, s1f1 = sb * fb;
	float w0 = ta * s0f0, wx = ta * s1f0, wy = tb * s0f0, wxy = tb * s1f0;
	float wz = ta * s0f1, wxz = ta * s1f1, wyz = tb * s0f1, wxyz = tb * s1f1;
	if (zfirst) { /* X and Y components add the +Z corners first (matters when Z == 0 in 2-D) */
		sum[Z] += wz; sum[X + Z] += wxz; sum[Y + Z] += wyz; sum[X + Y + Z] += wxyz;
		ref[Z] += wz * val; ref[X + Z] += wxz * val; ref[Y + Z] += wyz * val; ref[X + Y + Z] += wxyz * val;
		sum[0] += w0; sum[X] += wx; sum[Y] += wy; sum[X + Y] += wxy;
		ref[0] += w0 * val; ref[X] += wx * val; ref[Y] += wy * val; ref[X + Y] += wxy * val;
	} else {
		sum[0] += w0; sum[X] += wx; sum[Y] += wy; sum[X + Y] += wxy;
		sum[Z] += wz; sum[X + Z] += wxz; sum[Y + Z] += wyz; sum[X + Y + Z] += wxyz;
		ref[0] += w0 * val; ref[X] += wx * val; ref[Y] += wy * val; ref[X + Y] += wxy * val;
		ref[Z] += wz * val; ref[X + Z] += wxz * val; ref[Y + Z] += wyz * val; ref[X + Y + Z] += wxyz * val;
	}
}
/* mapPartsToMAC, flip.cpp:637-661, in two halves (the slab decomposition adds ghost-plane sums in between) */
int mf_map_parts_to_mac_accum(int sx, int sy, int sz, float* vel, float* weight, int64_t np, int64_t ps, const float* pos,
                              const int32_t* pflag, const float* pvel, const int32_t* ptype, int exclude, int deterministic,
                              void* st) {
	(void)st;
	(void)deterministic;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	memset(weight, 0, sizeof(float) * 3 * n);
	memset(vel, 0, sizeof(float) * 3 * n);
	for (int64_t p = 0; p < np; p++) { /* KERNEL(pts, single): particle order, one thread (flip.cpp:619) */
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
		Bi b = build_index(&d, x, y, z), s = build_index_shift(&d, x, y, z);
		int64_t ix = ((int64_t)b.zi * sy + b.yi) * sx + s.xi;
		scatter8(vel + ix, weight + ix, d.Y, d.Z, b.t0, b.t1, s.s0, s.s1, b.f0, b.f1, pvel[p], 1);
		int64_t iy = ((int64_t)b.zi * sy + s.yi) * sx + b.xi;
		scatter8(vel + n + iy, weight + n + iy, d.Y, d.Z, s.t0, s.t1, b.s0, b.s1, b.f0, b.f1, pvel[ps + p], 1);
		int64_t iz = ((int64_t)s.zi * sy + b.yi) * sx + b.xi;
		scatter8(vel + 2 * n + iz, weight + 2 * n + iz, d.Y, d.Z, b.t0, b.t1, b.s0, b.s1, s.f0, s.f1, pvel[2 * ps + p], 0);
	}
	return 0;
}
int mf_map_parts_to_mac_finish(int64_t n3, float* vel, float* velOld, float* weight, void* st) {
	mf_grid_stomp(n3, weight, 1e-6f, st);       /* weight->stomp(Vec3(VECTOR_EPSILON)) */
	mf_grid_safe_divide(n3, vel, weight, st);   /* vel.safeDivide(*weight) */
	if (velOld) memcpy(velOld, vel, sizeof(float) * n3); /* velOld.copyFrom(vel) */
	return 0;
}
int mf_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* velOld, float* weight, int64_t np, int64_t ps,
                        const float* pos, const int32_t* pflag, const float* pvel, const int32_t* ptype, int exclude,
                        int deterministic, void* st) {
	mf_map_parts_to_mac_accum(sx, sy, sz, vel, weight, np, ps, pos, pflag, pvel, ptype, exclude, deterministic, st);
	return mf_map_parts_to_mac_finish(3 * (int64_t)sx * sy * sz, vel, velOld, weight, st);
}
/* mapMACToParts, flip.cpp:709-721 */
int mf_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t ps, const float* pos,
                        const int32_t* pflag, float* pvel, const int32_t* ptype, int exclude, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
#pragma omp parallel for
	for (int64_t p = 0; p < np; p++) {
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		float v[3];
		interpol_mac(&d, vel, pos[p], pos[ps + p], pos[2 * ps + p], v);
		pvel[p] = v[0];
		pvel[ps + p] = v[1];
		pvel[2 * ps + p] = v[2];
	}
	return 0;
}
/* ---- APIC transfers, plugin/apic.cpp ------------------------------------------------------------------------
 * Index/weight set-up shared by both directions (apic.cpp:29-33, 119-123): face index f* = (IndexInt)pos, centre index
 * c* = (IndexInt)(pos - 0.5) with the subtraction in double (0.5 is a double literal) and truncation toward zero,
 * wf = clamp(pos - f, 0, 1) in fp32, wc = clamp(Real(pos - c - 0.5), 0, 1) with the "- 0.5" in double. */
typedef struct {
	int64_t f[3], c[3];
	float wf[3], wc[3];
} ApicIdx;
static inline float clamp01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
static inline void apic_setup(const float pos[3], ApicIdx* a) {
	for (int q = 0; q < 3; q++) {
		a->f[q] = (int64_t)pos[q];
		a->c[q] = (int64_t)((double)pos[q] - 0.5);
		a->wf[q] = clamp01(pos[q] - (float)a->f[q]);
		a->wc[q] = clamp01((float)((double)(pos[q] - (float)a->c[q]) - 0.5));
	}
}
/* face comp (0 u, 1 v, 2 w): base index per axis, grid position of the base node, per-axis weight pairs */
static inline void apic_face(const ApicIdx* a, int comp, int64_t b[3], float gpos[3], float W[3][2]) {
	for (int q = 0; q < 3; q++) {
		const int onface = (q == comp);
		b[q] = onface ? a->f[q] : a->c[q];
		gpos[q] = onface ? (float)a->f[q] : (float)((double)a->c[q] + 0.5);
		const float w = onface ? a->wf[q] : a->wc[q];
		W[q][0] = 1.f - w;
		W[q][1] = w;
	}
}
/* apicMapPartsToMAC, apic.cpp:92-110 (knApicMapLinearVec3ToMACGrid :19-90, KERNEL(pts, single)).  The reference does not
 * bound-check the node index (":34 TODO"); a face whose base index falls outside the grid, and nodes past the end, are
 * skipped here (out-of-bounds writes in the reference).  mass: MAC grid (SoA), always written. */
int mf_apic_map_parts_to_mac(int sx, int sy, int sz, float* vel, float* mass, int64_t np, int64_t ps, const float* pos,
                             const int32_t* pflag, const float* pvel, const float* cpx, const float* cpy, const float* cpz,
                             const int32_t* ptype, int exclude, void* st) {
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	const float* cp[3] = {cpx, cpy, cpz};
	memset(mass, 0, sizeof(float) * 3 * n);
	memset(vel, 0, sizeof(float) * 3 * n);
	for (int64_t p = 0; p < np; p++) {
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		const float P[3] = {pos[p], pos[ps + p], pos[2 * ps + p]};
		ApicIdx a;
		apic_setup(P, &a);
		for (int comp = 0; comp < (d.is3d ? 3 : 2); comp++) {
			int64_t b[3];
			float gpos[3], W[3][2];
			apic_face(&a, comp, b, gpos, W);
			const int64_t gidx = b[0] + b[1] * d.Y + b[2] * d.Z;
			if (gidx < 0 || gidx >= n) continue;
			const float c0 = cp[comp][p], c1 = cp[comp][ps + p], c2 = cp[comp][2 * ps + p];
			const float vc = pvel[comp * ps + p];
			float* mg = mass + comp * n;
			float* vg = vel + comp * n;
			for (int i = 0; i < 2; i++)
				for (int j = 0; j < 2; j++)
					for (int k = 0; k < 2; k++) {
						const int64_t node = gidx + i + j * d.Y + k * d.Z;
						if (node >= n) continue;
						const float w = W[0][i] * W[1][j] * W[2][k];
						const float dx = (gpos[0] + (float)i) - P[0], dy = (gpos[1] + (float)j) - P[1], dz = (gpos[2] + (float)k) - P[2];
						mg[node] += w;
						vg[node] += w * vc;
						vg[node] += w * (c0 * dx + c1 * dy + c2 * dz);
					}
		}
	}
	mf_grid_stomp(3 * n, mass, 1e-6f, st);     /* mass->stomp(VECTOR_EPSILON) */
	mf_grid_safe_divide(3 * n, vel, mass, st); /* vel.safeDivide(*mass) */
	return 0;
}
/* apicMapMACGridToParts, apic.cpp:175-181 (knApicMapLinearMACGridToVec3 :112-173) */
int mf_apic_map_mac_to_parts(int sx, int sy, int sz, const float* vel, int64_t np, int64_t ps, const float* pos,
                             const int32_t* pflag, float* pvel, float* cpx, float* cpy, float* cpz, const int32_t* ptype,
                             int exclude, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	float* cp[3] = {cpx, cpy, cpz};
	const float gw[2] = {-1.f, 1.f};
#pragma omp parallel for
	for (int64_t p = 0; p < np; p++) {
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		const float P[3] = {pos[p], pos[ps + p], pos[2 * ps + p]};
		ApicIdx a;
		apic_setup(P, &a);
		for (int comp = 0; comp < 3; comp++) {
			float v = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
			if (comp < 2 || d.is3d) {
				int64_t b[3];
				float gpos[3], W[3][2];
				apic_face(&a, comp, b, gpos, W);
				const int64_t gidx = b[0] + b[1] * d.Y + b[2] * d.Z;
				for (int i = 0; i < 2; i++)
					for (int j = 0; j < 2; j++)
						for (int k = 0; k < 2; k++) {
							const int64_t node = gidx + i + j * d.Y + k * d.Z;
							const float vg = (node >= 0 && node < n) ? vel[comp * n + node] : 0.f;
							v += W[0][i] * W[1][j] * W[2][k] * vg;
							g0 += gw[i] * W[1][j] * W[2][k] * vg;
							g1 += W[0][i] * gw[j] * W[2][k] * vg;
							g2 += W[0][i] * W[1][j] * gw[k] * vg;
						}
			}
			pvel[comp * ps + p] = v;
			cp[comp][p] = g0;
			cp[comp][ps + p] = g1;
			cp[comp][2 * ps + p] = g2;
		}
	}
	return 0;
}
/* flipVelocityUpdate, flip.cpp:724-742: pvel = flipRatio*(v + delta) + (1.0 - flipRatio)*v2 ; the second
 * scalar is a double, so that product is rounded from fp64 (vectorbase.h:282-284) */
int mf_flip_velocity_update(int sx, int sy, int sz, const float* vel, const float* velOld, int64_t np, int64_t ps,
                            const float* pos, const int32_t* pflag, float* pvel, float flipRatio, const int32_t* ptype,
                            int exclude, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
#pragma omp parallel for
	for (int64_t p = 0; p < np; p++) {
		if (skip_particle(pflag, ptype, exclude, p)) continue;
		float v1[3], v2[3];
		interpol_mac(&d, velOld, pos[p], pos[ps + p], pos[2 * ps + p], v1);
		interpol_mac(&d, vel, pos[p], pos[ps + p], pos[2 * ps + p], v2);
		for (int c = 0; c < 3; c++) {
			float v = pvel[c * ps + p];
			float delta = v2[c] - v1[c];
			float a = flipRatio * (v + delta);
			float b = (float)((1.0 - flipRatio) * v2[c]);
			pvel[c * ps + p] = a + b;
		}
	}
	return 0;
}
/* mapPartsToGrid(+Vec3), flip.cpp:663-687; setInterpol interpol.h:96-113; knSafeDivReal flip.cpp:607-615 */
int mf_map_parts_to_grid(int sx, int sy, int sz, int ncomp, float* target, float* wtmp, int64_t np, int64_t ps,
                         const float* pos, const int32_t* pflag, const float* psrc, int deterministic, void* st) {
	(void)st;
	(void)deterministic;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = 1, Y = d.Y, Z = d.Z;
	memset(target, 0, sizeof(float) * ncomp * n);
	memset(wtmp, 0, sizeof(float) * n);
	for (int64_t p = 0; p < np; p++) {
		if (pflag[p] & MF_PDELETE) continue;
		Bi b = build_index(&d, pos[p], pos[ps + p], pos[2 * ps + p]);
		int64_t idx = (int64_t)b.xi + Y * b.yi + Z * b.zi;
		float s0f0 = b.s0 * b.f0, s1f0 = b.s1 * b.f0, s0f1 = b.s0 * b.f1, s1f1 = b.s1 * b.f1;
		float w0 = b.t0 * s0f0, wx = b.t0 * s1f0, wy = b.t1 * s0f0, wxy = b.t1 * s1f0;
		float wz = b.t0 * s0f1, wxz = b.t0 * s1f1, wyz = b.t1 * s0f1, wxyz = b.t1 * s1f1;
		float* sum = wtmp + idx;
		sum[Z] += wz; sum[X + Z] += wxz; sum[Y + Z] += wyz; sum[X + Y + Z] += wxyz;
		for (int c = 0; c < ncomp; c++) {
			float* ref = target + c * n + idx;
			float v = psrc[c * ps + p];
			ref[Z] += wz * v; ref[X + Z] += wxz * v; ref[Y + Z] += wyz * v; ref[X + Y + Z] += wxyz * v;
		}
		sum[0] += w0; sum[X] += wx; sum[Y] += wy; sum[X + Y] += wxy;
		for (int c = 0; c < ncomp; c++) {
			float* ref = target + c * n + idx;
			float v = psrc[c * ps + p];
			ref[0] += w0 * v; ref[X] += wx * v; ref[Y] += wy * v; ref[X + Y] += wxy * v;
		}
	}
#pragma omp parallel for
	for (int64_t idx = 0; idx < n; idx++) {
		if (wtmp[idx] < 1e-6f) {
			for (int c = 0; c < ncomp; c++) target[c * n + idx] = 0.;
		} else {
			float dv = wtmp[idx];
			for (int c = 0; c < ncomp; c++) target[c * n + idx] = (dv) ? (target[c * n + idx] / dv) : target[c * n + idx];
		}
	}
	return 0;
}
/* mapGridToParts(+Vec3), flip.cpp:693-704 */
int mf_map_grid_to_parts(int sx, int sy, int sz, int ncomp, const float* source, int64_t np, int64_t ps,
                         const float* pos, const int32_t* pflag, float* ptarget, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
#pragma omp parallel for
	for (int64_t p = 0; p < np; p++) {
		if (pflag[p] & MF_PDELETE) continue;
		for (int c = 0; c < ncomp; c++) ptarget[c * ps + p] = interpol1(&d, source + c * d.n, pos[p], pos[ps + p], pos[2 * ps + p]);
	}
	return 0;
}

/* GridBase::isInBounds(Vec3, bnd) -> toVec3i truncation, grid.h:65, 430-438 */
static inline int in_bounds_pos(const Dim* d, float x, float y, float z, int bnd) {
	int i = (int)x, j = (int)y, k = (int)z;
	int r = i >= bnd && j >= bnd && i < d->sx - bnd && j < d->sy - bnd;
	if (d->is3d)
		r &= (k >= bnd && k < d->gsz - bnd); /* positions are global coordinates: the z extent is the whole domain's */
	else
		r &= (k == 0);
	return r;
}
static inline int flag_at(const Dim* d, const int32_t* flags, float x, float y, float z) { /* FlagGrid::getAt, grid.h:324 */
	int k = (int)z - d->zoff; /* plane inside the slab window (identity without a window) */
	k = k < 0 ? 0 : (k > d->sz - 1 ? d->sz - 1 : k);
	return flags[IDX(*d, (int)x, (int)y, k)];
}
/* GridAdvectKernel, particle.h:458-481 */
static void grid_advect_kernel(const Dim* d, const int32_t* flags, const float* vel, int64_t np, int64_t ps,
                               const float* pos, int32_t* pflag, float dt, int deleteInObstacle, int stopInObstacle,
                               int skipNew, const int32_t* ptype, int exclude, float* u) {
#pragma omp parallel for
	for (int64_t p = 0; p < np; p++) {
		if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude)) || (skipNew && (pflag[p] & MF_PNEW))) {
			u[p] = u[ps + p] = u[2 * ps + p] = 0.;
			continue;
		}
		float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
		if (deleteInObstacle || stopInObstacle) {
			if (!in_bounds_pos(d, x, y, z, 1) || (flag_at(d, flags, x, y, z) & MF_OBSTACLE)) {
				if (stopInObstacle) u[p] = u[ps + p] = u[2 * ps + p] = 0.;
				if (deleteInObstacle) pflag[p] |= MF_PDELETE;
				continue;
			}
		}
		float v[3];
		interpol_mac(d, vel, x, y, z, v);
		u[p] = v[0] * dt;
		u[ps + p] = v[1] * dt;
		u[2 * ps + p] = v[2] * dt;
	}
}
/* ParticleSystem::advectInGrid, particle.h:526-550 + integratePointSet, util/integrator.h:26-78 */
int mf_advect_in_grid(int sx, int sy, int sz, const int32_t* flags, const float* vel, int64_t np, int64_t ps,
                      float* pos, int32_t* pflag, float dt, int mode, int deleteInObstacle, int stopInObstacle,
                      int skipNew, const int32_t* ptype, int exclude, float* scratch, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	float *x0 = scratch, *u = scratch + 3 * ps, *ut = scratch + 6 * ps;
	for (int c = 0; c < 3; c++) memcpy(x0 + c * ps, pos + c * ps, sizeof(float) * np); /* posOld / PosType x0(x) */
	for (int c = 0; c < 3; c++) memset(u + c * ps, 0, sizeof(float) * np);             /* returns(vector<Vec3> u(size)) */
#define RUN() grid_advect_kernel(&d, flags, vel, np, ps, pos, pflag, dt, deleteInObstacle, stopInObstacle, skipNew, ptype, exclude, u)
	RUN();
	if (mode == MF_INT_EULER) {
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) pos[c * ps + p] += u[c * ps + p];
	} else if (mode == MF_INT_RK2) {
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) pos[c * ps + p] = x0[c * ps + p] + (float)(0.5 * u[c * ps + p]);
		RUN();
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) pos[c * ps + p] = x0[c * ps + p] + u[c * ps + p];
	} else if (mode == MF_INT_RK4) {
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) {
				int64_t q = c * ps + p;
				ut[q] = u[q];
				pos[q] = x0[q] + (float)(0.5 * u[q]);
				ut[q] += u[q]; /* the fork's extra accumulation, integrator.h:55 */
			}
		RUN();
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) {
				int64_t q = c * ps + p;
				pos[q] = x0[q] + (float)(0.5 * u[q]);
				ut[q] += 2 * u[q];
			}
		RUN();
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) {
				int64_t q = c * ps + p;
				pos[q] = x0[q] + u[q];
				ut[q] += 2 * u[q];
			}
		RUN();
		for (int c = 0; c < 3; c++)
			for (int64_t p = 0; p < np; p++) {
				int64_t q = c * ps + p;
				pos[q] = x0[q] + (float)(1. / 6.) * (ut[q] + u[q]);
			}
	} else
		return fail("unknown integration type");
#undef RUN
	if (!deleteInObstacle) {
		/* KnClampPositions, particle.h:507-523 */
		const float hi[3] = {(float)sx - 1.f, (float)sy - 1.f, (float)d.gsz - 1.f};
#pragma omp parallel for
		for (int64_t p = 0; p < np; p++) {
			if (pflag[p] & MF_PDELETE) continue;
			if (ptype && (ptype[p] & exclude)) {
				for (int c = 0; c < 3; c++) pos[c * ps + p] = x0[c * ps + p];
				continue;
			}
			float q[3] = {pos[p], pos[ps + p], pos[2 * ps + p]};
			if (!in_bounds_pos(&d, q[0], q[1], q[2], 0))
				for (int c = 0; c < 3; c++) q[c] = q[c] < 0.f ? 0.f : (q[c] > hi[c] ? hi[c] : q[c]);
			if (stopInObstacle && (flag_at(&d, flags, q[0], q[1], q[2]) & MF_OBSTACLE)) {
				/* bisectBacktracePos, particle.h:494-504 */
				const float o[3] = {x0[p], x0[ps + p], x0[2 * ps + p]};
				float s = 0.;
				for (int i = 1; i < 5; ++i) {
					float ds = 1. / (float)(1 << i);
					float a = (float)(1. - (s + ds)), b = s + ds;
					float tx = (float)(o[0] * (1. - (s + ds))) + q[0] * b;
					float ty = (float)(o[1] * (1. - (s + ds))) + q[1] * b;
					float tz = (float)(o[2] * (1. - (s + ds))) + q[2] * b;
					(void)a;
					if (!(flag_at(&d, flags, tx, ty, tz) & MF_OBSTACLE)) s += ds;
				}
				for (int c = 0; c < 3; c++) q[c] = (float)(o[c] * (1. - (s))) + q[c] * (s);
			}
			pos[p] = q[0];
			pos[ps + p] = q[1];
			pos[2 * ps + p] = q[2];
		}
	} else {
		/* KnDeleteInObstacle, particle.h:485-491 */
#pragma omp parallel for
		for (int64_t p = 0; p < np; p++) {
			if (pflag[p] & MF_PDELETE) continue;
			float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
			if (!in_bounds_pos(&d, x, y, z, 1) || (flag_at(&d, flags, x, y, z) & MF_OBSTACLE)) pflag[p] |= MF_PDELETE;
		}
	}
	return 0;
}

/* ================================================================================================
 * glue: setWallBcs / addBuoyancy / addGravity kernels, plugin/extforces.cpp
 * ============================================================================================== */
/* KnSetWallBcs, extforces.cpp:187-237 (KERNEL(): all cells) */
int mf_set_wall_bcs(int sx, int sy, int sz, const int32_t* flags, float* vel, const float* obvel, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				int f = flags[idx];
				int curFluid = f & MF_FLUID, curObs = f & MF_OBSTACLE;
				float bx = 0.f, by = 0.f, bz = 0.f;
				if (!curFluid && !curObs) continue;
				if (obvel) {
					bx = obvel[idx];
					by = obvel[n + idx];
					if (d.is3d) bz = obvel[2 * n + idx];
				}
				if (i > 0 && (flags[idx - X] & MF_OBSTACLE)) vx[idx] = bx;
				if (i > 0 && curObs && (flags[idx - X] & MF_FLUID)) vx[idx] = bx;
				if (j > 0 && (flags[idx - Y] & MF_OBSTACLE)) vy[idx] = by;
				if (j > 0 && curObs && (flags[idx - Y] & MF_FLUID)) vy[idx] = by;
				if (!d.is3d) {
					vz[idx] = 0;
				} else {
					if (k > 0 && (flags[idx - Z] & MF_OBSTACLE)) vz[idx] = bz;
					if (k > 0 && curObs && (flags[idx - Z] & MF_FLUID)) vz[idx] = bz;
				}
				if (curFluid) {
					if ((i > 0 && (flags[idx - X] & MF_STICK)) || (i < sx - 1 && (flags[idx + X] & MF_STICK))) vy[idx] = vz[idx] = 0;
					if ((j > 0 && (flags[idx - Y] & MF_STICK)) || (j < sy - 1 && (flags[idx + Y] & MF_STICK))) vx[idx] = vz[idx] = 0;
					if (d.is3d && ((k > 0 && (flags[idx - Z] & MF_STICK)) || (k < sz - 1 && (flags[idx + Z] & MF_STICK))))
						vx[idx] = vy[idx] = 0;
				}
			}
	return 0;
}
/* KnAddBuoyancy, extforces.cpp:73-81: vel += (0.5*strength) * (f + f) evaluated in fp64 */
int mf_add_buoyancy(int sx, int sy, int sz, const int32_t* flags, const float* fac, float* vel, float fx, float fy,
                    float fz, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				if (flags[idx - X] & MF_FLUID) vx[idx] += (0.5 * fx) * (fac[idx] + fac[idx - X]);
				if (flags[idx - Y] & MF_FLUID) vy[idx] += (0.5 * fy) * (fac[idx] + fac[idx - Y]);
				if (d.is3d && (flags[idx - Z] & MF_FLUID)) vz[idx] += (0.5 * fz) * (fac[idx] + fac[idx - Z]);
			}
	return 0;
}
/* KnApplyForce, extforces.cpp:46-60 */
int mf_apply_force(int sx, int sy, int sz, const int32_t* flags, float* vel, float fx, float fy, float fz,
                   const float* exclude, int additive, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, X = d.X, Y = d.Y, Z = d.Z;
	float *vx = vel, *vy = vel + n, *vz = vel + 2 * n;
#pragma omp parallel for
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				int64_t idx = IDX(d, i, j, k);
				int curFluid = flags[idx] & MF_FLUID, curEmpty = flags[idx] & MF_EMPTY;
				if (!curFluid && !curEmpty) continue;
				if (exclude && (exclude[idx] < 0.)) continue;
				if ((flags[idx - X] & MF_FLUID) || (curFluid && (flags[idx - X] & MF_EMPTY))) vx[idx] = additive ? vx[idx] + fx : fx;
				if ((flags[idx - Y] & MF_FLUID) || (curFluid && (flags[idx - Y] & MF_EMPTY))) vy[idx] = additive ? vy[idx] + fy : fy;
				if (d.is3d && ((flags[idx - Z] & MF_FLUID) || (curFluid && (flags[idx - Z] & MF_EMPTY))))
					vz[idx] = additive ? vz[idx] + fz : fz;
			}
	return 0;
}

/* ================================================================================================
 * FLIP glue
 * ============================================================================================== */
/* extrapolateMACSimple (phiObs == NULL), fastmarch.cpp:231-376 */
int mf_extrapolate_mac_simple(int sx, int sy, int sz, const int32_t* flags, float* vel, int distance, int intoObs,
                              int32_t* tmp, float* velTmp, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	const int dim = d.is3d ? 3 : 2;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	for (int c = 0; c < dim; c++) {
		const int64_t o = c == 0 ? 1 : (c == 1 ? d.Y : d.Z);
		float* vc = vel + c * n;
		memset(tmp, 0, sizeof(int32_t) * n);
		for (int k = K0(d, 1); k < K1(d, 1); k++)
			for (int j = 1; j < sy - 1; j++)
				for (int i = 1; i < sx - 1; i++) {
					int64_t idx = IDX(d, i, j, k);
					int mark = (flags[idx] & MF_FLUID) || (flags[idx - o] & MF_FLUID);
					if (intoObs) mark = mark && !(flags[idx] & MF_OBSTACLE) && !(flags[idx - o] & MF_OBSTACLE);
					if (mark) tmp[idx] = 1;
				}
		for (int dd = 1; dd < 1 + distance; dd++) {
			/* knExtrapolateMACSimple: in place; a pass reads markers == dd only and writes dd+1 into zeros */
#pragma omp parallel for
			for (int k = K0(d, 1); k < K1(d, 1); k++)
				for (int j = 1; j < sy - 1; j++)
					for (int i = 1; i < sx - 1; i++) {
						int64_t idx = IDX(d, i, j, k);
						if (tmp[idx] != 0) continue;
						int nbs = 0;
						float avg = 0.;
						for (int q = 0; q < 2 * dim; q++)
							if (tmp[idx + nb[q]] == dd) {
								avg += vc[idx + nb[q]];
								nbs++;
							}
						if (nbs > 0) {
							tmp[idx] = dd + 1;
							vc[idx] = avg / nbs;
						}
					}
		}
	}
	memcpy(velTmp, vel, sizeof(float) * 3 * n);
	/* knExtrapolateIntoBnd, fastmarch.cpp:261-300 */
#pragma omp parallel for
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int64_t idx = IDX(d, i, j, k);
				int c = 0;
				float v[3] = {0, 0, 0};
				const int isObs = flags[idx] & MF_OBSTACLE;
#define TAKE(q, comp, neg)                                                       \
	{                                                                            \
		v[0] = velTmp[q]; v[1] = velTmp[n + (q)]; v[2] = velTmp[2 * n + (q)];   \
		if (isObs && ((neg) ? v[comp] < 0. : v[comp] > 0.)) v[comp] = 0.;        \
		c++;                                                                     \
	}
				if (i == 0) TAKE(idx + 1, 0, 1)
				else if (i == sx - 1) TAKE(idx - 1, 0, 0)
				if (j == 0) TAKE(idx + d.Y, 1, 1)
				else if (j == sy - 1) TAKE(idx - d.Y, 1, 0)
				if (d.is3d) {
					if (k == 0) TAKE(idx + d.Z, 2, 1)
					else if (k == sz - 1) TAKE(idx - d.Z, 2, 0)
				}
#undef TAKE
				if (c > 0) {
					vel[idx] = v[0] / (float)c;
					vel[n + idx] = v[1] / (float)c;
					vel[2 * n + idx] = v[2] / (float)c;
				}
			}
	return 0;
}
/* extrapolateMACFromWeight, fastmarch.cpp:378-430 */
int mf_extrapolate_mac_from_weight(int sx, int sy, int sz, float* vel, float* weight, int distance, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	const int dim = d.is3d ? 3 : 2;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	for (int c = 0; c < dim; c++) {
		float *vc = vel + c * n, *wc = weight + c * n;
		for (int k = K0(d, 1); k < K1(d, 1); k++)
			for (int j = 1; j < sy - 1; j++)
				for (int i = 1; i < sx - 1; i++) {
					int64_t idx = IDX(d, i, j, k);
					if (wc[idx] > 0.) wc[idx] = 1.0;
				}
		for (int dd = 1; dd < 1 + distance; dd++) {
#pragma omp parallel for
			for (int k = K0(d, 1); k < K1(d, 1); k++)
				for (int j = 1; j < sy - 1; j++)
					for (int i = 1; i < sx - 1; i++) {
						int64_t idx = IDX(d, i, j, k);
						if (wc[idx] != 0) continue;
						int nbs = 0;
						float avg = 0.;
						for (int q = 0; q < 2 * dim; q++)
							if (wc[idx + nb[q]] == dd) {
								avg += vc[idx + nb[q]];
								nbs++;
							}
						if (nbs > 0) {
							wc[idx] = dd + 1;
							vc[idx] = avg / nbs;
						}
					}
		}
	}
	return 0;
}
/* markFluidCells, plugin/flip.cpp:142-188 */
int mf_mark_fluid_cells(int sx, int sy, int sz, int32_t* flags, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                        const int32_t* ptype, int exclude, const float* phiObs, int32_t* ftmp, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t idx = 0; idx < d.n; idx++)
		if (flags[idx] & MF_FLUID) flags[idx] = (flags[idx] | MF_EMPTY) & ~MF_FLUID;
	for (int64_t p = 0; p < np; p++) {
		if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) continue;
		int i = (int)pos[p], j = (int)pos[ps + p], k = (int)pos[2 * ps + p] - d.zoff; /* plane inside the slab window */
		if (!in_bounds(&d, i, j, k)) continue;
		int64_t idx = IDX(d, i, j, k);
		if (flags[idx] & MF_EMPTY) flags[idx] = (flags[idx] | MF_FLUID) & ~MF_EMPTY;
	}
	if (phiObs) {
		memcpy(ftmp, flags, sizeof(int32_t) * d.n);
		for (int k = K0(d, 1); k < K1(d, 1); k++)
			for (int j = 1; j < sy - 1; j++)
				for (int i = 1; i < sx - 1; i++) {
					int64_t idx = IDX(d, i, j, k);
					if (phiObs[idx] > 0.) continue;
					if (!(flags[idx] & MF_EMPTY)) continue;
					int set = 0;
					if ((flags[idx - 1] & MF_FLUID) && (phiObs[idx + 1] <= 0.)) set = 1;
					if ((flags[idx + 1] & MF_FLUID) && (phiObs[idx - 1] <= 0.)) set = 1;
					if ((flags[idx - d.Y] & MF_FLUID) && (phiObs[idx + d.Y] <= 0.)) set = 1;
					if ((flags[idx + d.Y] & MF_FLUID) && (phiObs[idx - d.Y] <= 0.)) set = 1;
					if (d.is3d) {
						if ((flags[idx - d.Z] & MF_FLUID) && (phiObs[idx + d.Z] <= 0.)) set = 1;
						if ((flags[idx + d.Z] & MF_FLUID) && (phiObs[idx - d.Z] <= 0.)) set = 1;
					}
					if (set) ftmp[idx] = (flags[idx] | MF_FLUID) & ~MF_EMPTY;
				}
		memcpy(flags, ftmp, sizeof(int32_t) * d.n);
	}
	return 0;
}

/* KnProjectOutOfBnd, particle.h:579-590 */
int mf_project_out_of_bnd(int sx, int sy, int sz, int64_t np, int64_t ps, float* pos, const int32_t* pflag, float bnd,
                          int axis, const int32_t* ptype, int exclude, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t p = 0; p < np; p++) {
		if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) continue;
		float* x = pos + p;
		float* y = pos + ps + p;
		float* z = pos + 2 * ps + p;
		if (axis & 1) *x = *x > bnd ? *x : bnd;                           /* std::max(pos.x, bnd) */
		if (axis & 2) { float hi = (float)sx - bnd; *x = hi < *x ? hi : *x; } /* std::min(pos.x, X - bnd) */
		if (axis & 4) *y = *y > bnd ? *y : bnd;
		if (axis & 8) { float hi = (float)sy - bnd; *y = hi < *y ? hi : *y; }
		if (d.is3d) {
			if (axis & 16) *z = *z > bnd ? *z : bnd;
			if (axis & 32) { float hi = (float)d.gsz - bnd; *z = hi < *z ? hi : *z; }
		}
	}
	return 0;
}

/* getGradient, grid.h:556-573 */
static void get_gradient(const Dim* d, const float* data, int i, int j, int k, float g[3]) {
	if (i > d->sx - 2) i = d->sx - 2;
	if (j > d->sy - 2) j = d->sy - 2;
	if (i < 1) i = 1;
	if (j < 1) j = 1;
	g[0] = data[IDX(*d, i + 1, j, k)] - data[IDX(*d, i - 1, j, k)];
	g[1] = data[IDX(*d, i, j + 1, k)] - data[IDX(*d, i, j - 1, k)];
	g[2] = 0.f;
	if (d->is3d) {
		int kg = k + d->zoff; /* clamp to [1, size - 2] of the whole domain; inside a slab window stay addressable */
		if (kg > d->gsz - 2) kg = d->gsz - 2;
		if (kg < 1) kg = 1;
		k = kg - d->zoff;
		if (k > d->sz - 2) k = d->sz - 2;
		if (k < 1) k = 1;
		g[2] = data[IDX(*d, i, j, k + 1)] - data[IDX(*d, i, j, k - 1)];
	}
}
/* normalize(Vector3D<float>&), vectorbase.h:421-434: returns the norm */
static float normalize3(float v[3]) {
	float l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	const float eps2 = 1e-6f * 1e-6f;
	float nrm;
	if (fabs((double)l - 1.) < eps2) {
		nrm = 1.f;
	} else if (l > eps2) {
		nrm = sqrtf(l);
		float fac = (float)(1. / (double)nrm);
		v[0] *= fac;
		v[1] *= fac;
		v[2] *= fac;
	} else {
		v[0] = v[1] = v[2] = 0.f;
		nrm = 0.f;
	}
	return nrm;
}
/* knPushOutofObs, plugin/flip.cpp:584-596 */
int mf_push_out_of_obs(int sx, int sy, int sz, int64_t np, int64_t ps, float* pos, const int32_t* pflag, const float* phiObs,
                       float shift, float thresh, const int32_t* ptype, int exclude, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t p = 0; p < np; p++) {
		if ((pflag[p] & MF_PDELETE) || (ptype && (ptype[p] & exclude))) continue;
		float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
		int i, j, k;
		if (!cell_of(&d, x, y, z, &i, &j, &k)) continue;
		float v = interpol1(&d, phiObs, x, y, z);
		if (v < thresh) {
			float g[3];
			get_gradient(&d, phiObs, i, j, k, g);
			if (normalize3(g) < 1e-6f) continue;
			float f = thresh - v + shift;
			pos[p] = x + g[0] * f;
			pos[ps + p] = y + g[1] * f;
			pos[2 * ps + p] = z + g[2] * f;
		}
	}
	return 0;
}

/* gridParticleIndex, plugin/flip.cpp:273-320 */
int mf_grid_particle_index(int sx, int sy, int sz, int64_t np, int64_t ps, const float* pos, const int32_t* pflag,
                           int32_t* indexSys, int32_t* index, int32_t* counter, int32_t* keys, int32_t* vals,
                           int64_t* n_indexed_host, void* st) {
	(void)st;
	(void)keys;
	(void)vals;
	Dim d = mkdim(sx, sy, sz);
	memset(counter, 0, sizeof(int32_t) * d.n);
	memset(index, 0, sizeof(int32_t) * d.n);
	int64_t inactive = 0;
	for (int64_t p = 0; p < np; p++) {
		if (pflag[p] & MF_PDELETE) { inactive++; continue; }
		int i, j, k;
		if (!cell_of(&d, pos[p], pos[ps + p], pos[2 * ps + p], &i, &j, &k)) { inactive++; continue; }
		index[IDX(d, i, j, k)]++;
	}
	int64_t run = 0;
	for (int64_t c = 0; c < d.n; c++) {
		int num = index[c];
		index[c] = (int32_t)run;
		run += num;
	}
	for (int64_t p = 0; p < np; p++) {
		if (pflag[p] & MF_PDELETE) continue;
		int i, j, k;
		if (!cell_of(&d, pos[p], pos[ps + p], pos[2 * ps + p], &i, &j, &k)) continue;
		int64_t c = IDX(d, i, j, k);
		indexSys[index[c] + counter[c]] = (int32_t)p;
		counter[c]++;
	}
	if (n_indexed_host) *n_indexed_host = np - inactive;
	return 0;
}

/* ComputeUnionLevelsetPindex + setBound(0.5, 0), plugin/flip.cpp:322-363 */
int mf_union_particle_levelset(int sx, int sy, int sz, int64_t np, int64_t ps, const float* pos, const int32_t* indexSys,
                               int64_t n_indexed, const int32_t* index, float* phi, float radiusFactor, const int32_t* ptype,
                               int exclude, void* st) {
	(void)st;
	(void)np;
	Dim d = mkdim(sx, sy, sz);
	/* calculateRadiusFactor, flip.cpp:198-200 (double arithmetic, returned as Real), then 0.5 * it */
	const float rf = (float)((d.is3d ? sqrt(3.) : sqrt(2.)) * ((double)radiusFactor + .01));
	const float radius = (float)(0.5 * (double)rf);
	const int r = (int)radius + 1, rZ = d.is3d ? r : 0;
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				const float gx = (float)i + 0.5f, gy = (float)j + 0.5f, gz = (float)(k + d.zoff) + 0.5f;
				float phiv = (float)((double)radius * 1.0);
				for (int zj = k - rZ; zj <= k + rZ; zj++)
					for (int yj = j - r; yj <= j + r; yj++)
						for (int xj = i - r; xj <= i + r; xj++) {
							if (!in_bounds(&d, xj, yj, zj)) continue;
							const int64_t c = IDX(d, xj, yj, zj);
							const int64_t pStart = index[c];
							const int64_t pEnd = (c + 1 < d.n) ? index[c + 1] : n_indexed;
							for (int64_t q = pStart; q < pEnd; q++) {
								const int psrc = indexSys[q];
								if (ptype && (ptype[psrc] & exclude)) continue;
								const float dx = gx - pos[psrc], dy = gy - pos[ps + psrc], dz = gz - pos[2 * ps + psrc];
								/* norm(), vectorbase.h:385-389 */
								const float l = dx * dx + dy * dy + dz * dz;
								const float eps2 = 1e-6f * 1e-6f;
								float nr;
								if (l <= eps2) nr = 0.f;
								else nr = (fabs((double)l - 1.) < eps2) ? 1.f : sqrtf(l);
								const float cand = fabsf(nr) - radius;
								phiv = cand < phiv ? cand : phiv;   /* std::min(phiv, cand) */
							}
						}
				phi[IDX(d, i, j, k)] = phiv;
			}
	return mf_grid_set_bound(sx, sy, sz, phi, 0.5f, 0, st);
}

/* ---- shape level sets, shapes.cpp:178-229 (Box), 303-307 (Sphere), 367-385 (Cylinder); fp32, left to right ---- */
#define HD inline
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
#define HD inline
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
int mf_shape_apply_to_grid(int sx, int sy, int sz, int kind, const float* q, int gridkind, void* grid, const float* value,
                           const int32_t* respectFlags, void* st) {
	(void)st;
	if (kind < 0 || kind > 2 || gridkind < 0 || gridkind > 3) return fail("mf_shape_apply_to_grid: unknown shape or grid kind");
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n;
	float* gf = (float*)grid;
	int32_t* gi = (int32_t*)grid;
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				const int64_t idx = IDX(d, i, j, k);
				if (respectFlags && (respectFlags[idx] & MF_OBSTACLE)) continue;
				const float x = (float)i, y = (float)j, z = (float)(k + d.zoff);
				if (gridkind == 2) {
					if (shape_inside(kind, q, x, y + 0.5f, z + 0.5f)) gf[idx] = value[0];
					if (shape_inside(kind, q, x + 0.5f, y, z + 0.5f)) gf[n + idx] = value[1];
					if (shape_inside(kind, q, x + 0.5f, y + 0.5f, z)) gf[2 * n + idx] = value[2];
				} else if (shape_inside(kind, q, x + 0.5f, y + 0.5f, z + 0.5f)) {
					if (gridkind == 0) gf[idx] = value[0];
					else if (gridkind == 3) gi[idx] = (int32_t)value[0];
					else {
						gf[idx] = value[0];
						gf[n + idx] = value[1];
						gf[2 * n + idx] = value[2];
					}
				}
			}
	return 0;
}
int mf_shape_levelset(int sx, int sy, int sz, int kind, const float* q, float* phi, void* st) {
	(void)st;
	if (kind < 0 || kind > 2) return fail("mf_shape_levelset: unknown shape kind");
	Dim d = mkdim(sx, sy, sz);
#pragma omp parallel for
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				const float x = (float)i + 0.5f, y = (float)j + 0.5f, z = (float)(k + d.zoff) + 0.5f;
				phi[IDX(d, i, j, k)] = kind == 0 ? shape_sdf_box(d.is3d, q, x, y, z) : (kind == 1 ? shape_sdf_sphere(q, x, y, z) : shape_sdf_cylinder(q, x, y, z));
			}
	return 0;
}

/* resetOutflow, extforces.cpp:134-161 (particles: flagged, not compacted) */
int mf_reset_outflow(int sx, int sy, int sz, int32_t* flags, float* phi, float* real, int64_t np, int64_t ps, const float* pos,
                     int32_t* pflag, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	if (pos && pflag)
		for (int64_t p = 0; p < np; p++) {
			if (pflag[p] & MF_PDELETE) continue;
			const int i = (int)pos[p], j = (int)pos[ps + p], k = (int)pos[2 * ps + p];
			if (i < 0 || j < 0 || k < 0 || i >= sx || j >= sy || k >= sz) continue;
			if (flags[IDX(d, i, j, k)] & MF_OUTFLOW) pflag[p] |= MF_PDELETE;
		}
	for (int64_t idx = 0; idx < d.n; idx++)
		if (flags[idx] & MF_OUTFLOW) {
			flags[idx] = (flags[idx] | MF_EMPTY) & ~MF_FLUID;
			if (phi) phi[idx] = 0.5f;
			if (real) real[idx] = 0.f;
		}
	return 0;
}

/* knSetBoundary, grid.cpp:629-637 */
int mf_grid_set_bound(int sx, int sy, int sz, float* grid, float value, int w, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				int bnd = (i <= w || i >= sx - 1 - w || j <= w || j >= sy - 1 - w || z_wall(&d, k, w));
				if (bnd) grid[IDX(d, i, j, k)] = value;
			}
	return 0;
}

/* extrapolateLsSimple, fastmarch.cpp:432-522 */
int mf_extrapolate_ls_simple(int sx, int sy, int sz, float* phi, int distance, int inside, int include_walls, int32_t* tmp,
                             void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int dim = d.is3d ? 3 : 2;
	const int64_t nb[6] = {1, -1, d.Y, -d.Y, d.Z, -d.Z};
	memset(tmp, 0, sizeof(int32_t) * d.n);
	float direction = 1.f;
	if (!inside) {
		for (int k = 0; k < sz; k++) {
			if (d.is3d && (k + d.zoff < 1 || k + d.zoff >= d.gsz - 1)) continue;
			for (int j = 1; j < sy - 1; j++)
				for (int i = 1; i < sx - 1; i++)
					if (phi[IDX(d, i, j, k)] < 0.) tmp[IDX(d, i, j, k)] = 1;
		}
	} else {
		direction = -1.f;
		const int b = include_walls ? 0 : 1;
		for (int k = 0; k < sz; k++) {
			if (d.is3d && (k + d.zoff < b || k + d.zoff >= d.gsz - b)) continue; /* the b-cell border of the DOMAIN (global planes) */
			for (int j = b; j < sy - b; j++)
				for (int i = b; i < sx - b; i++)
					if (phi[IDX(d, i, j, k)] > 0.) tmp[IDX(d, i, j, k)] = 1;
		}
	}
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				const int64_t idx = IDX(d, i, j, k);
				if (tmp[idx]) continue;
				for (int n = 0; n < 2 * dim; n++)
					if (tmp[idx + nb[n]] == 1) { tmp[idx] = 2; break; }
			}
	for (int dd = 2; dd < 1 + distance; dd++) {
		/* knExtrapolateLsSimple: cells written in this pass get dd+1, which no cell of this pass reads */
		for (int k = K0(d, 1); k < K1(d, 1); k++)
			for (int j = 1; j < sy - 1; j++)
				for (int i = 1; i < sx - 1; i++) {
					const int64_t idx = IDX(d, i, j, k);
					if (tmp[idx] != 0) continue;
					int nbs = 0;
					float avg = 0.f;
					for (int n = 0; n < 2 * dim; n++)
						if (tmp[idx + nb[n]] == dd) {
							avg += phi[idx + nb[n]];
							nbs++;
						}
					if (nbs > 0) {
						tmp[idx] = dd + 1;
						phi[idx] = avg / nbs + direction;
					}
				}
	}
	const float rest = (float)(direction * (distance + 2));
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++)
				if (tmp[IDX(d, i, j, k)] == 0) phi[IDX(d, i, j, k)] = rest;
	return 0;
}

/* KnSetPartType, plugin/ptsplugins.cpp:56-59 */
int mf_set_part_type(int sx, int sy, int sz, const int32_t* flags, int64_t np, int64_t ps, const float* pos, int32_t* ptype,
                     int mark, int stype, int cflag, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t p = 0; p < np; p++) {
		const float x = pos[p], y = pos[ps + p], z = pos[2 * ps + p];
		if (in_bounds_pos(&d, x, y, z, 0) && (flag_at(&d, flags, x, y, z) & cflag) && (ptype[p] & stype)) ptype[p] = mark;
	}
	return 0;
}

/* knMarkIsolatedFluidCell, grid.cpp:987-1005 */
int mf_mark_isolated_fluid_cell(int sx, int sy, int sz, int32_t* flags, int mark, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t idx = 0; idx < d.n; idx++) {
		if (!(flags[idx] & MF_FLUID)) continue;
		/* the reference indexes idx+-stride without a bounds check; border cells are never fluid in practice, the
		 * restatement simply treats out-of-array neighbours as non-fluid */
#define FL(o) ((idx + (o) >= 0 && idx + (o) < d.n) ? (flags[idx + (o)] & MF_FLUID) : 0)
		if (FL(-1) || FL(1) || FL(-d.Y) || FL(d.Y)) continue;
		if (d.is3d && (FL(-d.Z) || FL(d.Z))) continue;
#undef FL
		flags[idx] = mark;
	}
	return 0;
}

/* KnAddForcePvel, ptsplugins.cpp:20-29: da = a*dt (fp32 Vec3 * Real) */
int mf_add_force_pvel(int64_t np, int64_t ps, float* pvel, float ax, float ay, float az, float dt, const int32_t* ptype,
                      int exclude, void* st) {
	(void)st;
	const float dx = ax * dt, dy = ay * dt, dz = az * dt;
	for (int64_t p = 0; p < np; p++) {
		if (ptype && (ptype[p] & exclude)) continue;
		pvel[p] += dx;
		pvel[ps + p] += dy;
		pvel[2 * ps + p] += dz;
	}
	return 0;
}
/* KnUpdateVelocityFromDeltaPos, ptsplugins.cpp:31-41: over_dt = 1.0/dt evaluated in double */
int mf_update_velocity_from_delta_pos(int64_t np, int64_t ps, const float* pos, float* pvel, const float* xprev, float dt,
                                      const int32_t* ptype, int exclude, void* st) {
	(void)st;
	const float over_dt = (float)(1.0 / (double)dt);
	for (int64_t p = 0; p < np; p++) {
		if (ptype && (ptype[p] & exclude)) continue;
		for (int c = 0; c < 3; c++) pvel[c * ps + p] = (pos[c * ps + p] - xprev[c * ps + p]) * over_dt;
	}
	return 0;
}
/* KnStepEuler, ptsplugins.cpp:43-53 */
int mf_euler_step(int64_t np, int64_t ps, float* pos, const float* pvel, float dt, const int32_t* ptype, int exclude, void* st) {
	(void)st;
	for (int64_t p = 0; p < np; p++) {
		if (ptype && (ptype[p] & exclude)) continue;
		for (int c = 0; c < 3; c++) pos[c * ps + p] += pvel[c * ps + p] * dt;
	}
	return 0;
}
/* KnJoin / KnSubtract, levelset.cpp:107-118 */
int mf_levelset_join(int64_t n, float* phi, const float* other, void* st) {
	(void)st;
	for (int64_t i = 0; i < n; i++) phi[i] = other[i] < phi[i] ? other[i] : phi[i];   /* min(a, b) = (b < a) ? b : a */
	return 0;
}
int mf_levelset_subtract(int64_t n, float* phi, const float* other, const int32_t* flags, int subtractType, void* st) {
	(void)st;
	for (int64_t i = 0; i < n; i++) {
		if (flags && (flags[i] & subtractType) == 0) continue;
		if (other[i] < 0.) phi[i] = other[i] * -1.f;
	}
	return 0;
}

/* knInterpolateGridTempl, grid.h:576-581 */
int mf_interpolate_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source, int ncomp,
                        float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* st) {
	(void)st;
	Dim t = mkdim(tsx, tsy, tsz), s = mkdim_src(ssx, ssy, ssz); /* each grid under its own slab window */
	for (int k = 0; k < tsz; k++)
		for (int j = 0; j < tsy; j++)
			for (int i = 0; i < tsx; i++) {
				float px = (float)i * sfx + ox, py = (float)j * sfy + oy, pz = (float)(k + t.zoff) * sfz + oz; /* global plane */
				if (!s.is3d) pz = 0.f;
				for (int c = 0; c < ncomp; c++)
					target[c * t.n + IDX(t, i, j, k)] = orderSpace == 1 ? interpol1(&s, source + c * s.n, px, py, pz)
					                                                    : interpol_cubic(&s, source + c * s.n, px, py, pz, ncomp == 3);
			}
	return 0;
}
/* KnInterpolateMACGrid, plugin/waveletturbulence.cpp:59-71 */
int mf_interpolate_mac_grid(int tsx, int tsy, int tsz, float* target, int ssx, int ssy, int ssz, const float* source,
                            float sfx, float sfy, float sfz, float ox, float oy, float oz, int orderSpace, void* st) {
	(void)st;
	Dim t = mkdim(tsx, tsy, tsz), s = mkdim_src(ssx, ssy, ssz); /* each grid under its own slab window */
	for (int k = 0; k < tsz; k++)
		for (int j = 0; j < tsy; j++)
			for (int i = 0; i < tsx; i++) {
				const float px = (float)i * sfx + ox, py = (float)j * sfy + oy, pz = (float)(k + t.zoff) * sfz + oz;
				const int64_t idx = IDX(t, i, j, k);
				/* MACGrid::getInterpolatedHi -> interpolMAC (grid.h:269-275), one component of each evaluation is kept */
				float v[3];
				if (orderSpace == 2) {
					/* getInterpolatedHi(pos - 0.5 e_c, 2)[c] = interpolCubicMAC(...)[c] = interpolCubic<Vec3>((pos - 0.5 e_c) + 0.5 e_c)[c] */
					target[idx] = interpol_cubic_mac(&s, source, 0, px - 0.5f, py, pz);
					target[t.n + idx] = interpol_cubic_mac(&s, source, 1, px, py - 0.5f, pz);
					target[2 * t.n + idx] = s.is3d ? interpol_cubic_mac(&s, source, 2, px, py, pz - 0.5f) : 0.f;
					continue;
				}
				interpol_mac(&s, source, px - 0.5f, py, pz, v);
				target[idx] = v[0];
				interpol_mac(&s, source, px, py - 0.5f, pz, v);
				target[t.n + idx] = v[1];
				if (s.is3d) {
					interpol_mac(&s, source, px, py, pz - 0.5f, v);
					target[2 * t.n + idx] = v[2];
				} else {
					target[2 * t.n + idx] = 0.f;
				}
			}
	return 0;
}

/* ================================================================================================
 * wavelet noise, noisefield.{h,cpp}; MTRand, util/randomstream.h
 * ============================================================================================== */
typedef struct {
	uint32_t state[624];
	uint32_t* next;
	int left;
} MTRand;
static uint32_t mt_twist(uint32_t m, uint32_t s0, uint32_t s1) {
	return m ^ (((s0 & 0x80000000u) | (s1 & 0x7fffffffu)) >> 1) ^ ((uint32_t)(-(int32_t)(s1 & 1u)) & 0x9908b0dfu);
}
static void mt_reload(MTRand* m) { /* randomstream.h:261-276 */
	uint32_t* p = m->state;
	int i;
	for (i = 624 - 397; i--; ++p) *p = mt_twist(p[397], p[0], p[1]);
	for (i = 397; --i; ++p) *p = mt_twist(p[397 - 624], p[0], p[1]);
	*p = mt_twist(p[397 - 624], p[0], m->state[0]);
	m->left = 624;
	m->next = m->state;
}
static void mt_seed(MTRand* m, uint32_t seed) { /* initialize + reload, randomstream.h:174-179, 243-259 */
	m->state[0] = seed;
	for (int i = 1; i < 624; i++) m->state[i] = 1812433253u * (m->state[i - 1] ^ (m->state[i - 1] >> 30)) + (uint32_t)i;
	mt_reload(m);
}
static uint32_t mt_int(MTRand* m) { /* randInt, randomstream.h:138-152 */
	if (m->left == 0) mt_reload(m);
	--m->left;
	uint32_t s1 = *m->next++;
	s1 ^= (s1 >> 11);
	s1 ^= (s1 << 7) & 0x9d2c5680u;
	s1 ^= (s1 << 15) & 0xefc60000u;
	return s1 ^ (s1 >> 18);
}
static double mt_rand(MTRand* m) { return (double)mt_int(m) * (1.0 / 4294967295.0); }
static double mt_rand_norm(MTRand* m, double mean, double variance) { /* randNorm, randomstream.h:129-136 */
	double r = sqrt(-2.0 * log(1.0 - ((double)mt_int(m) + 0.5) * (1.0 / 4294967296.0))) * variance;
	double phi = 2.0 * 3.14159265358979323846264338328 * ((double)mt_int(m) * (1.0 / 4294967296.0));
	return mean + r * cos(phi);
}

static const float noise_aCoeffs[32] = {
	0.000334, -0.001528, 0.000410, 0.003545, -0.000938, -0.008233, 0.002172, 0.019120,
	-0.005040, -0.044412, 0.011655, 0.103311, -0.025936, -0.243780, 0.033979, 0.655340,
	0.655340, 0.033979, -0.243780, -0.025936, 0.103311, 0.011655, -0.044412, -0.005040,
	0.019120, 0.002172, -0.008233, -0.000938, 0.003546, 0.000410, -0.001528, 0.000334};
static const float noise_pCoeffs[4] = {0.25, 0.75, 0.75, 0.25};
static void noise_downsample(const float* from, float* to, int n, int stride) { /* noisefield.cpp:42-50 */
	const float* a = &noise_aCoeffs[16];
	for (int i = 0; i < n / 2; i++) {
		to[i * stride] = 0;
		for (int k = 2 * i - 16; k < 2 * i + 16; k++) to[i * stride] += a[k - 2 * i] * from[(k & 127) * stride];
	}
}
static int mod_slow(int x, int n) {
	int m = x % n;
	return (m < 0) ? m + n : m;
}
static void noise_upsample(const float* from, float* to, int n, int stride) { /* noisefield.cpp:54-63 */
	const float* pp = &noise_pCoeffs[1];
	for (int i = 0; i < n; i++) {
		to[i * stride] = 0;
		for (int k = i / 2 - 1; k < i / 2 + 3; k++)
			to[i * stride] = (float)((double)to[i * stride] + 0.5 * (double)pp[k - i / 2] * (double)from[mod_slow(k, n / 2) * stride]);
	}
}
/* WaveletNoiseField::generateTile, noisefield.cpp:95-186 */
int mf_noise_generate_tile(float* tile, int seed, void* st) {
	(void)st;
	const int n = 128;
	const int64_t n3 = (int64_t)n * n * n, n3d = n3 * 3;
	float* noise3 = tile;
	float* temp13 = (float*)calloc(n3d, sizeof(float));
	float* temp23 = (float*)calloc(n3d, sizeof(float));
	if (!temp13 || !temp23) return fail("out of memory");
	MTRand mt;
	mt_seed(&mt, (uint32_t)seed);
	for (int64_t i = 0; i < n3d; i++) noise3[i] = (float)mt_rand_norm(&mt, (double)0.f, (double)1.f);
	for (int t = 0; t < 3; t++) {
		for (int iy = 0; iy < n; iy++)
			for (int iz = 0; iz < n; iz++) {
				const int64_t i = iy * n + (int64_t)iz * n * n + t * n3;
				noise_downsample(&noise3[i], &temp13[i], n, 1);
				noise_upsample(&temp13[i], &temp23[i], n, 1);
			}
		for (int ix = 0; ix < n; ix++)
			for (int iz = 0; iz < n; iz++) {
				const int64_t i = ix + (int64_t)iz * n * n + t * n3;
				noise_downsample(&temp23[i], &temp13[i], n, n);
				noise_upsample(&temp13[i], &temp23[i], n, n);
			}
		for (int ix = 0; ix < n; ix++)
			for (int iy = 0; iy < n; iy++) {
				const int64_t i = ix + iy * n + t * n3;
				noise_downsample(&temp23[i], &temp13[i], n, n * n);
				noise_upsample(&temp13[i], &temp23[i], n, n * n);
			}
	}
	for (int64_t i = 0; i < n3d; i++) noise3[i] -= temp23[i];
	int offset = n / 2;
	if (offset % 2 == 0) offset++;
	int64_t icnt = 0;
	for (int t = 0; t < 3; t++)
		for (int ix = 0; ix < n; ix++)
			for (int iy = 0; iy < n; iy++)
				for (int iz = 0; iz < n; iz++) {
					temp13[icnt] = noise3[((ix + offset) & 127) + ((iy + offset) & 127) * n + (int64_t)((iz + offset) & 127) * n * n + t * n3];
					icnt++;
				}
	for (int64_t i = 0; i < n3d; i++) noise3[i] += temp13[i];
	free(temp13);
	free(temp23);
	return 0;
}
/* mSeedOffset, noisefield.cpp:64-70 */
int mf_noise_seed_offset(int fixedSeed, float* out) {
	if (fixedSeed == -1) fixedSeed = 13322223 + 123;
	MTRand mt;
	mt_seed(&mt, (uint32_t)fixedSeed);
	float v[3];
	for (int c = 0; c < 3; c++) v[c] = (float)mt_rand(&mt);
	normalize3(v);
	out[0] = v[0];
	out[1] = v[1];
	out[2] = v[2];
	return 0;
}
/* WNoise, noisefield.h:163-196 */
static float wnoise(float p0, float p1, float p2, const float* data) {
	float w[3][3], t, result = 0;
	const float p[3] = {p0, p1, p2};
	int mid[3];
	for (int c = 0; c < 3; c++) {
		mid[c] = (int)ceilf(p[c] - 0.5f);
		t = (float)mid[c] - (p[c] - 0.5f);
		w[c][0] = t * t * 0.5f;
		w[c][2] = (1.f - t) * (1.f - t) * 0.5f;
		w[c][1] = 1.f - w[c][0] - w[c][2];
	}
	for (int z = -1; z <= 1; z++)
		for (int y = -1; y <= 1; y++)
			for (int x = -1; x <= 1; x++) {
				float weight = 1.0f;
				const int xC = (mid[0] + x) & 127;
				weight *= w[0][x + 1];
				const int yC = (mid[1] + y) & 127;
				weight *= w[1][y + 1];
				const int zC = (mid[2] + z) & 127;
				weight *= w[2][z + 1];
				result += weight * data[(zC * 128 + yC) * 128 + xC];
			}
	return result;
}
/* WaveletNoiseField::evaluate, noisefield.h:313-336 */
static float noise_evaluate(const float* P, const float* tile, float x, float y, float z) {
	float pos[3] = {x, y, z};
	for (int c = 0; c < 3; c++) pos[c] *= P[c];
	for (int c = 0; c < 3; c++) pos[c] += P[3 + c];
	for (int c = 0; c < 3; c++) pos[c] += P[6];
	for (int c = 0; c < 3; c++) pos[c] *= P[7 + c];
	for (int c = 0; c < 3; c++) pos[c] += P[10 + c];
	float v = wnoise(pos[0], pos[1], pos[2], tile);
	v += P[13];
	v *= P[14];
	if (P[15] != 0.f) {
		if (v < P[16]) v = P[16];
		if (v > P[17]) v = P[17];
	}
	return v;
}
/* KnApplyNoiseInfl, plugin/initplugins.cpp:27-36 */
int mf_density_inflow(int sx, int sy, int sz, const int32_t* flags, float* density, const float* sdf, const float* tile,
                      const float* P, float scale, float sigma, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				const int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID) || sdf[idx] > sigma) continue;
				double f = 1.0 - 0.5 / (double)sigma * (double)(sdf[idx] + sigma);
				if (f < 0.0) f = 0.0;
				else if (f > 1.0) f = 1.0;
				const float factor = (float)f;
				const float target = noise_evaluate(P, tile, (float)i, (float)j, (float)(k + d.zoff)) * scale * factor; /* global plane */
				if (density[idx] < target) density[idx] = target;
			}
	return 0;
}

/* ================================================================================================
 * wavelet turbulence pieces: plugin/waveletturbulence.cpp, plugin/extforces.cpp:409-428, noisefield.cpp:191-297
 * ============================================================================================== */
/* KnApplyComputeEnergy, waveletturbulence.cpp:180-189 */
int mf_compute_energy(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* energy, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	for (int64_t idx = 0; idx < d.n; idx++) {
		float e = 0.f;
		if (flags[idx] & MF_FLUID) {
			float v[3];
			get_centered(&d, vel, idx, v);
			e = (float)(0.5 * (double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]));
		}
		energy[idx] = e;
	}
	return 0;
}
/* norm(), vectorbase.h:385-389 */
static float norm3(float x, float y, float z) {
	const float l = x * x + y * y + z * z;
	const float eps2 = 1e-6f * 1e-6f;
	if (l <= eps2) return 0.f;
	return (fabs((double)l - 1.) < eps2) ? 1.f : sqrtf(l);
}
int mf_vorticity_confinement(int sx, int sy, int sz, float* vel, const int32_t* flags, float strength, const float* strengthCell,
                             float* vc, float* curl, float* nrm, float* force, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	memset(vc, 0, sizeof(float) * 3 * n);
	memset(curl, 0, sizeof(float) * 3 * n);
	memset(force, 0, sizeof(float) * 3 * n);
#define IN_LOOP                                    \
	for (int k = K0(d, 1); k < K1(d, 1); k++)      \
		for (int j = 1; j < sy - 1; j++)           \
			for (int i = 1; i < sx - 1; i++)
	/* GetCentered, commonkernels.h:126-131 */
	IN_LOOP {
		const int64_t idx = IDX(d, i, j, k);
		vc[idx] = (float)(0.5 * (double)(vel[idx] + vel[idx + 1]));
		vc[n + idx] = (float)(0.5 * (double)(vel[n + idx] + vel[n + idx + Y]));
		float vz = (float)(0.5 * (double)(vel[2 * n + idx] + 0.f));
		if (d.is3d) vz = (float)((double)vz + 0.5 * (double)vel[2 * n + idx + Z]);
		else vz = 0.f;
		vc[2 * n + idx] = vz;
	}
	/* CurlOp, commonkernels.h:38-47 */
	IN_LOOP {
		const int64_t idx = IDX(d, i, j, k);
		const float *gx = vc, *gy = vc + n, *gz = vc + 2 * n;
		float v0 = 0.f, v1 = 0.f;
		const float v2 = (float)(0.5 * (double)((gy[idx + 1] - gy[idx - 1]) - (gx[idx + Y] - gx[idx - Y])));
		if (d.is3d) {
			v0 = (float)(0.5 * (double)((gz[idx + Y] - gz[idx - Y]) - (gy[idx + Z] - gy[idx - Z])));
			v1 = (float)(0.5 * (double)((gx[idx + Z] - gx[idx - Z]) - (gz[idx + 1] - gz[idx - 1])));
		}
		curl[idx] = v0;
		curl[n + idx] = v1;
		curl[2 * n + idx] = v2;
	}
	/* GridNorm, commonkernels.h:116-118 */
	for (int64_t idx = 0; idx < n; idx++) nrm[idx] = norm3(curl[idx], curl[n + idx], curl[2 * n + idx]);
	/* KnConfForce, extforces.cpp:410-417 */
	IN_LOOP {
		const int64_t idx = IDX(d, i, j, k);
		float g[3];
		g[0] = (float)(0.5 * (double)(nrm[idx + 1] - nrm[idx - 1]));
		g[1] = (float)(0.5 * (double)(nrm[idx + Y] - nrm[idx - Y]));
		g[2] = (float)(0.5 * 0.);
		if (d.is3d) g[2] = (float)(0.5 * (double)(nrm[idx + Z] - nrm[idx - Z]));
		normalize3(g);
		float str = strength;
		if (strengthCell) str += strengthCell[idx];
		const float cx = curl[idx], cy = curl[n + idx], cz = curl[2 * n + idx];
		force[idx] = str * ((g[1] * cz) - (g[2] * cy));
		force[n + idx] = str * ((g[2] * cx) - (g[0] * cz));
		force[2 * n + idx] = str * ((g[0] * cy) - (g[1] * cx));
	}
	/* KnApplyForceField(additive, !isMAC), extforces.cpp:24-43 */
	IN_LOOP {
		const int64_t idx = IDX(d, i, j, k);
		const int curFluid = flags[idx] & MF_FLUID, curEmpty = flags[idx] & MF_EMPTY;
		if (!curFluid && !curEmpty) continue;
		const float fx = (float)(0.5 * (double)(force[idx - 1] + force[idx]));
		const float fy = (float)(0.5 * (double)(force[n + idx - Y] + force[n + idx]));
		float fz = 0.f;
		if (d.is3d) fz = (float)(0.5 * (double)(force[2 * n + idx - Z] + force[2 * n + idx]));
		if ((flags[idx - 1] & MF_FLUID) || (curFluid && (flags[idx - 1] & MF_EMPTY))) vel[idx] = vel[idx] + fx;
		if ((flags[idx - Y] & MF_FLUID) || (curFluid && (flags[idx - Y] & MF_EMPTY))) vel[n + idx] = vel[n + idx] + fy;
		if (d.is3d && ((flags[idx - Z] & MF_FLUID) || (curFluid && (flags[idx - Z] & MF_EMPTY)))) vel[2 * n + idx] = vel[2 * n + idx] + fz;
	}
#undef IN_LOOP
	return 0;
}
/* downsampleNeumann / upsampleNeumann, noisefield.cpp:191-227 */
static void noise_downsample_neumann(const float* from, float* to, int n, int64_t stride) {
	const float* a = &noise_aCoeffs[16];
	for (int i = 0; i < n / 2; i++) {
		to[i * stride] = 0;
		for (int k = 2 * i - 16; k < 2 * i + 16; k++) {
			float fv;
			if (k < 0) fv = from[0];
			else if (k > n - 1) fv = from[(n - 1) * stride];
			else fv = from[k * stride];
			to[i * stride] += a[k - 2 * i] * fv;
		}
	}
}
static void noise_upsample_neumann(const float* from, float* to, int n, int64_t stride) {
	const float* pp = &noise_pCoeffs[1];
	for (int i = 0; i < n; i++) {
		to[i * stride] = 0;
		for (int k = i / 2 - 1; k < i / 2 + 3; k++) {
			float fv;
			if (k > n / 2 - 1) fv = from[(n / 2 - 1) * stride];
			else if (k < 0) fv = from[0];
			else fv = from[k * stride];
			to[i * stride] = (float)((double)to[i * stride] + 0.5 * (double)pp[k - i / 2] * (double)fv);
		}
	}
}
/* WaveletNoiseField::computeCoefficients, noisefield.cpp:229-297 */
int mf_compute_wavelet_coeffs(int sx, int sy, int sz, float* input, float* temp13, float* temp23, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	const int64_t n3 = d.n, sxy = (int64_t)sx * sy;
	float* noise3 = input;
	for (int64_t i = 0; i < n3; i++) temp13[i] = temp23[i] = 0.f;
	for (int iz = 0; iz < sz; iz++)
		for (int iy = 0; iy < sy; iy++) {
			const int64_t i = iz * sxy + (int64_t)iy * sx;
			noise_downsample_neumann(&noise3[i], &temp13[i], sx, 1);
			noise_upsample_neumann(&temp13[i], &temp23[i], sx, 1);
		}
	for (int iz = 0; iz < sz; iz++)
		for (int ix = 0; ix < sx; ix++) {
			const int64_t i = iz * sxy + ix;
			noise_downsample_neumann(&temp23[i], &temp13[i], sy, sx);
			noise_upsample_neumann(&temp13[i], &temp23[i], sy, sx);
		}
	if (d.is3d)
		for (int iy = 0; iy < sy; iy++)
			for (int ix = 0; ix < sx; ix++) {
				const int64_t i = (int64_t)iy * sx + ix;
				noise_downsample_neumann(&temp23[i], &temp13[i], sz, sxy);
				noise_upsample_neumann(&temp13[i], &temp23[i], sz, sxy);
			}
	for (int64_t i = 0; i < n3; i++) {
		const float residual = noise3[i] - temp23[i];
		temp13[i] = sqrtf(fabsf(residual));
	}
	float smoothingFactor = (float)(1. / 6.);
	if (!d.is3d) smoothingFactor = (float)(1. / 4.);
	for (int k = K0(d, 1); k < K1(d, 1); k++)
		for (int j = 1; j < sy - 1; j++)
			for (int i = 1; i < sx - 1; i++) {
				const int64_t c = k * sxy + (int64_t)j * sx + i;
				float res = temp13[c - 1] + temp13[c + 1];
				res += temp13[c - sx] + temp13[c + sx];
				if (d.is3d) res += temp13[c - sxy] + temp13[c + sxy];
				input[c] = res * smoothingFactor;
			}
	return 0;
}
/* WNoiseVec, noisefield.h:210-310 */
static void wnoise_vec(float p0, float p1, float p2, const float* data, float out[3]) {
	const float p[3] = {p0, p1, p2};
	int mid[3];
	float t[3], w[3][3], dw[3][3], nb[3][3][3];
	for (int c = 0; c < 3; c++) {
		mid[c] = (int)ceil((double)(p[c] - 0.5f));
		t[c] = (float)mid[c] - (p[c] - 0.5f);
	}
	for (int z = -1; z <= 1; z++)
		for (int y = -1; y <= 1; y++)
			for (int x = -1; x <= 1; x++) {
				const int xC = (mid[0] + x) & 127, yC = (mid[1] + y) & 127, zC = (mid[2] + z) & 127;
				nb[x + 1][y + 1][z + 1] = data[zC * 128 * 128 + yC * 128 + xC];
			}
	for (int c = 0; c < 3; c++) {
		dw[c][0] = -t[c];
		dw[c][2] = (1.f - t[c]);
		dw[c][1] = 2.0f * t[c] - 1.0f;
		w[c][0] = t[c] * t[c] * 0.5f;
		w[c][2] = (1.f - t[c]) * (1.f - t[c]) * 0.5f;
		w[c][1] = 1.f - w[c][0] - w[c][2];
	}
	for (int comp = 0; comp < 3; comp++) {
		float result = 0.0f;
		for (int z = -1; z <= 1; z++)
			for (int y = -1; y <= 1; y++)
				for (int x = -1; x <= 1; x++) {
					const float a = (comp == 0 ? dw[0] : w[0])[x + 1], b = (comp == 1 ? dw[1] : w[1])[y + 1], c = (comp == 2 ? dw[2] : w[2])[z + 1];
					const float weight = a * b * c;
					result += weight * nb[x + 1][y + 1][z + 1];
				}
		out[comp] = result;
	}
}
/* WaveletNoiseField::evaluateVec, noisefield.h:338-364 */
static void noise_evaluate_vec(const float* P, const float* tile, float x, float y, float z, int t, float v[3]) {
	float pos[3] = {x, y, z};
	for (int c = 0; c < 3; c++) pos[c] *= P[c];
	for (int c = 0; c < 3; c++) pos[c] += P[3 + c];
	for (int c = 0; c < 3; c++) pos[c] += P[6];
	for (int c = 0; c < 3; c++) pos[c] *= P[7 + c];
	for (int c = 0; c < 3; c++) pos[c] += P[10 + c];
	wnoise_vec(pos[0], pos[1], pos[2], tile + (int64_t)t * 128 * 128 * 128, v);
	for (int c = 0; c < 3; c++) v[c] += P[13];
	for (int c = 0; c < 3; c++) v[c] *= P[14];
	if (P[15] != 0.f)
		for (int c = 0; c < 3; c++) {
			if (v[c] < P[16]) v[c] = P[16];
			if (v[c] > P[17]) v[c] = P[17];
		}
}
/* knApplyNoiseVec3, waveletturbulence.cpp:120-154 (uv == NULL) */
int mf_apply_noise_vec3(int sx, int sy, int sz, const int32_t* flags, float* target, const float* tile, const float* P,
                        float scale, float scaleSpatial, const float* weight, int wsx, int wsy, int wsz, const float* uv, int usx, int usy,
                        int usz, void* st) {
	(void)st;
	Dim d = mkdim(sx, sy, sz);
	if (uv && weight && (usx != wsx || usy != wsy || usz != wsz)) return fail("UV and weight grid have to match!");
	if (uv && !weight) { /* the size of the uv grid decides, waveletturbulence.cpp:161-166 */
		wsx = usx;
		wsy = usy;
		wsz = usz;
	}
	const int two = weight || uv;
	Dim wd = two ? mkdim_src(wsx, wsy, wsz) : d; /* a weight / uv grid of another size lives under its own (source) slab window */
	const int interp = two && (wd.gsz != d.gsz || wsx != sx || wsy != sy);
	if (two && !interp && (wsz != sz || wd.zoff != d.zoff)) return fail("applyNoiseVec3: weight grid of the same resolution must share the target's slab window");
	const float sf[3] = {(float)wsx / sx, (float)wsy / sy, (float)wd.gsz / d.gsz};   /* calcGridSizeFactor, grid.h:391-393 (whole-domain sizes) */
	for (int k = 0; k < sz; k++)
		for (int j = 0; j < sy; j++)
			for (int i = 0; i < sx; i++) {
				const int64_t idx = IDX(d, i, j, k);
				if (!(flags[idx] & MF_FLUID)) continue;
				float w = 1;
				if (weight) {
					if (!interp) w = weight[idx];
					else w = interpol1(&wd, weight, (float)i * sf[0], (float)j * sf[1], (float)(k + d.zoff) * sf[2]);
				}
				float pos[3] = {(float)i + 0.5f, (float)j + 0.5f, (float)(k + d.zoff) + 0.5f}; /* global cell centre */
				if (uv) { /* waveletturbulence.cpp:139-147 */
					if (!interp) {
						for (int c = 0; c < 3; c++) pos[c] = uv[c * d.n + idx];
					} else {
						for (int c = 0; c < 3; c++)
							pos[c] = interpol1(&wd, uv + c * wd.n, (float)i * sf[0], (float)j * sf[1], (float)(k + d.zoff) * sf[2]) / sf[c];
					}
				}
				for (int c = 0; c < 3; c++) pos[c] *= scaleSpatial;
				float d0[3], d1[3], d2[3];
				noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 0, d0);
				noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 1, d1);
				noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 2, d2);
				const float cu[3] = {d0[1] - d1[2], d2[2] - d0[0], d1[0] - d2[1]};
				for (int c = 0; c < 3; c++) target[c * d.n + idx] += cu[c] * scale * w;
			}
	return 0;
}

/* device-scalar variants (here: host pointers) */
int mf_grid_dot_dev(int64_t n, const float* a, const float* b, double* out, void* s) {
	(void)s;
	*out = dot64(n, a, b);
	return 0;
}
int mf_grid_max_abs_dev(int64_t n, const float* a, float* out, void* s) {
	(void)s;
	*out = maxabs(n, a);
	return 0;
}
int mf_grid_max_abs_dev_f64(int64_t n, const float* a, double* out, void* s) {
	(void)s;
	*out = (double)maxabs(n, a);
	return 0;
}
/* cgSolveDiffusion matrix set-up, conjugategrad.cpp:364-375 */
int mf_diffusion_matrix(int sx, int sy, int sz, const int32_t* flags, float* A0, float* Ai, float* Aj, float* Ak, float alpha,
                        void* st) {
	(void)st;
	const int64_t n = (int64_t)sx * sy * sz;
	for (int64_t idx = 0; idx < n; idx++) {
		if (flags[idx] & MF_OBSTACLE) {
			Ai[idx] = Aj[idx] = Ak[idx] = 0.f;
			A0[idx] = 1.f;
		} else {
			Ai[idx] *= alpha;
			Aj[idx] *= alpha;
			Ak[idx] *= alpha;
			A0[idx] *= alpha;
			A0[idx] = (float)((double)A0[idx] + 1.);
		}
	}
	return 0;
}
int mf_cg_slab_alpha(const double* g, int world, const float* sigma, float* alpha, const int32_t* state, void* s) {
	(void)s;
	if (state && state[0]) {
		alpha[0] = 0.f;
		alpha[1] = -0.f;
		return 0;
	}
	double acc = 0.0;
	for (int r = 0; r < world; r++) acc += g[2 * r + 1];
	const float dp = (float)acc;
	alpha[0] = (fabs((double)dp) > 0.) ? sigma[0] / dp : 0.f;
	alpha[1] = -alpha[0]; /* nalpha of the scalar block (mf_cg_slab_axpy2) */
	return 0;
}
int mf_apply_matrix_dot_dev(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                            const float* Ai, const float* Aj, const float* Ak, int k0, int k1, const void* scalars, double* dot,
                            void* st) {
	(void)scalars;
	if (k0 < 0 || k1 > sz || k0 > k1) return fail("mf_apply_matrix_dot_dev: invalid plane range");
	int rc = mf_apply_matrix(sx, sy, sz, flags, dst, src, A0, Ai, Aj, Ak, st);
	if (rc) return rc;
	const int64_t XY = (int64_t)sx * sy;
	double acc = 0.0;
	for (int64_t i = k0 * XY; i < k1 * XY; i++) acc += (double)(dst[i] * src[i]);
	dot[0] = acc;
	return 0;
}
int mf_cg_slab_axpy2(int64_t n, const void* scalars, float* x, const float* search, float* residual, const float* tmp,
                     double* maxabs, void* st) {
	(void)st;
	const float alpha = ((const float*)scalars)[1], nalpha = ((const float*)scalars)[2];
	float lo = 3.402823466e38f, hi = -3.402823466e38f;
	for (int64_t i = 0; i < n; i++) {
		x[i] = x[i] + alpha * search[i];
		const float r = residual[i] + nalpha * tmp[i];
		residual[i] = r;
		lo = r < lo ? r : lo;
		hi = r > hi ? r : hi;
	}
	lo = fabsf(lo);
	hi = fabsf(hi);
	maxabs[0] = n > 0 ? (double)(lo > hi ? lo : hi) : 0.0;
	return 0;
}
int mf_cg_slab_beta(const double* g, int world, float* sigma, float* beta, float* res, float accuracy, int iter, int32_t* state,
                    void* s) {
	(void)s;
	if (state && state[0]) return 0;
	double acc = 0.0, mx = 0.0;
	for (int r = 0; r < world; r++) {
		acc += g[2 * r + 1];
		mx = g[2 * r] > mx ? g[2 * r] : mx;
	}
	const float sigmaNew = (float)acc;
	res[0] = (float)mx;
	beta[0] = sigmaNew / sigma[0];
	sigma[0] = sigmaNew;
	if (state) {
		if (res[0] < accuracy) {
			state[0] = 1;
			state[1] = iter;
		} else if (!(res[0] < 1e35f)) {
			state[0] = 2;
			state[1] = iter;
		}
	}
	return 0;
}
int mf_grid_scaled_add_dev(int64_t n, float* me, const float* other, const float* factor, float sign, void* s) {
	return mf_grid_scaled_add(n, me, other, sign * factor[0], s);
}
int mf_update_search_vec_dev(int64_t n, float* dst, const float* src, const float* factor, void* s) {
	return mf_update_search_vec(n, dst, src, factor[0], s);
}

int mf_time_apply_matrix(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                         const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* st) {
	(void)sx; (void)sy; (void)sz; (void)flags; (void)dst; (void)src; (void)A0; (void)Ai; (void)Aj; (void)Ak; (void)reps;
	(void)avg_us; (void)st;
	return fail("mf_time_apply_matrix: HIP only");
}
int mf_time_apply_matrix_packed(int sx, int sy, int sz, const int32_t* flags, float* dst, const float* src, const float* A0,
                                const float* Ai, const float* Aj, const float* Ak, int reps, double* avg_us, void* st) {
	(void)sx; (void)sy; (void)sz; (void)flags; (void)dst; (void)src; (void)A0; (void)Ai; (void)Aj; (void)Ak; (void)reps;
	(void)avg_us; (void)st;
	return fail("mf_time_apply_matrix_packed: HIP only");
}

/* the same three calls the plugin layer makes for a plain system (pressure.cpp:32-84, conjugategrad.h:154-187, conjugategrad.cpp:210-307);
 * the coefficient grids are scratch of this call */
int mf_solve_pressure_fused(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* pressure, float* rhs,
                            float* residual, float* search, float* tmp, float* Aprecond, float accuracy, int maxIter,
                            int useL2Norm, float* out_host, void* st) {
	const int64_t n = (int64_t)sx * sy * sz;
	if (sz <= 1) return fail("mf_solve_pressure_fused: 3D only");
	float* A = (float*)calloc((size_t)(4 * n), sizeof(float));
	if (!A) return fail("mf_solve_pressure_fused: out of memory");
	memset(rhs, 0, sizeof(float) * (size_t)n);
	memset(residual, 0, sizeof(float) * (size_t)n);
	memset(search, 0, sizeof(float) * (size_t)n);
	memset(tmp, 0, sizeof(float) * (size_t)n);
	memset(Aprecond, 0, sizeof(float) * (size_t)n);
	int rc = mf_make_rhs(sx, sy, sz, flags, rhs, vel, NULL, NULL, NULL, NULL, NULL, 0.f, 1e-4f, NULL, NULL, st);
	if (!rc) rc = mf_make_laplace_matrix(sx, sy, sz, flags, A, A + n, A + 2 * n, A + 3 * n, NULL, st);
	if (!rc) rc = mf_cg_solve(sx, sy, sz, flags, pressure, rhs, residual, search, tmp, A, A + n, A + 2 * n, A + 3 * n, Aprecond, MF_PC_MICP, accuracy,
	                          maxIter, useL2Norm, out_host, st);
	free(A);
	return rc;
}

/* the two composite stretches of the z-slab PCG iteration (same arithmetic in the same order as the HIP library): as in
 * GridCg::iterate (conjugategrad.cpp:250-291), with `x += alpha * search` (:254) deferred to the search update (:283) -- nothing in
 * between reads x.  scalars = {sigma, alpha, nalpha, beta, resNorm, ..., word 12: xpending} */
int mf_cg_slab_after_dp(const double* gathered, int world, void* scalars, const int32_t* state_dev, int64_t own_off, int64_t n_own,
                        float* residual, float* tmp, double* maxabs_dev, int sx, int sy, int sz,
                        const int32_t* flags, const float* Aprecond, const float* Ai, const float* Aj, const float* Ak, double* dot_dev,
                        void* st) {
	float* sc = (float*)scalars;
	int rc = mf_cg_slab_alpha(gathered, world, sc + 0, sc + 1, state_dev, st);
	if (rc) return rc;
	const int stopped = state_dev && state_dev[0];
	((int32_t*)scalars)[12] = stopped ? 0 : 1;
	if (!stopped) {
		const float nalpha = sc[2];
		float* r = residual + own_off;
		const float* t = tmp + own_off;
		float lo = 3.402823466e38f, hi = -3.402823466e38f;
		for (int64_t i = 0; i < n_own; i++) {
			const float v = r[i] + nalpha * t[i];
			r[i] = v;
			lo = v < lo ? v : lo;
			hi = v > hi ? v : hi;
		}
		lo = fabsf(lo);
		hi = fabsf(hi);
		maxabs_dev[0] = n_own > 0 ? (double)(lo > hi ? lo : hi) : 0.0;
	}
	return mf_mic_apply_dot_dev(sx, sy, sz, flags, tmp, residual, Aprecond, Ai, Aj, Ak, dot_dev, st);
}
int mf_cg_slab_after_zr(const double* gathered, int world, void* scalars, float accuracy, int iter, int32_t* state_dev, int64_t own_off,
                        int64_t n_own, float* x, float* search, const float* tmp, void* st) {
	float* sc = (float*)scalars;
	int rc = mf_cg_slab_beta(gathered, world, sc + 0, sc + 3, sc + 4, accuracy, iter, state_dev, st);
	if (rc) return rc;
	if (!((int32_t*)scalars)[12]) return 0;
	const int stopped = state_dev && state_dev[0];
	const float alpha = sc[1], beta = sc[3];
	float* xs = x + own_off;
	float* s = search + own_off;
	const float* t = tmp + own_off;
	for (int64_t i = 0; i < n_own; i++) {
		const float sv = s[i];
		xs[i] = xs[i] + alpha * sv;
		if (!stopped) s[i] = t[i] + beta * sv;
	}
	return 0;
}
