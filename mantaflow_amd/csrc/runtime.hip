// runtime.hip -- error reporting, workspace, element-wise grid ops and reductions (gfx950).
#include "common.h"
#include <stdarg.h>
#include <float.h>

namespace mf {
thread_local char g_err[512];
thread_local int g_slab_zoff = 0, g_slab_gsz = 0;
thread_local int g_slab_src_zoff = 0, g_slab_src_gsz = 0;
int fail(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return 1;
}

static Workspace g_ws[16];
static bool g_ws_ok[16];
int get_workspace(Workspace** out) {
	int dev = 0;
	MF_HIP(hipGetDevice(&dev));
	if (dev < 0 || dev >= 16) return fail("device index %d out of range", dev);
	if (!g_ws_ok[dev]) {
		Workspace& w = g_ws[dev];
		MF_HIP(hipMalloc((void**)&w.partials, sizeof(double) * MAX_BLOCKS * 4));
		MF_HIP(hipMalloc((void**)&w.fpartials, sizeof(float) * MAX_BLOCKS * 4));
		MF_HIP(hipMalloc(&w.scalars, 4096));
		MF_HIP(hipMemset(w.scalars, 0, 4096));
		MF_HIP(hipHostMalloc(&w.host, 4096, hipHostMallocDefault));
		MF_HIP(hipMalloc((void**)&w.tilework, 4096));
		MF_HIP(hipMemset(w.tilework, 0, 4096));
		g_ws_ok[dev] = true;
	}
	*out = &g_ws[dev];
	return 0;
}
}  // namespace mf

using namespace mf;

// ---------------------------------------------------------------------------------------------------------
// element-wise kernels: grid-stride, float4 where the pointers allow it (all grids are 16-B aligned torch
// allocations; the scalar tail handles n % 4 and misaligned views)
// ---------------------------------------------------------------------------------------------------------
enum Op1 { OP_FILL, OP_ADDC, OP_MULC, OP_CLAMP, OP_STOMP };
template <int OP>
__device__ __forceinline__ float op1(float a, float p, float q) {
	if (OP == OP_FILL) return p;
	if (OP == OP_ADDC) return a + p;
	if (OP == OP_MULC) return a * p;
	if (OP == OP_CLAMP) return a < p ? p : (a > q ? q : a);  // general.h:137-141
	if (OP == OP_STOMP) return a < p ? 0.f : a;               // grid.cpp:247
	return a;
}
template <int OP>
__global__ void __launch_bounds__(BLOCK) k_unary(int64_t n, float* __restrict__ a, float p, float q) {
	const int64_t n4 = n >> 2;
	float4* a4 = (float4*)a;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
		float4 v = (OP == OP_FILL) ? make_float4(0, 0, 0, 0) : a4[i];
		v.x = op1<OP>(v.x, p, q);
		v.y = op1<OP>(v.y, p, q);
		v.z = op1<OP>(v.z, p, q);
		v.w = op1<OP>(v.w, p, q);
		a4[i] = v;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		a[i] = op1<OP>(OP == OP_FILL ? 0.f : a[i], p, q);
	}
}
enum Op2 { OP_COPY, OP_AXPY, OP_XPAY, OP_ADD, OP_SUB, OP_MUL, OP_SAFEDIV };
template <int OP>
__device__ __forceinline__ float op2(float a, float b, float f) {
	if (OP == OP_COPY) return b;
	if (OP == OP_AXPY) return a + f * b;               // gridScaledAdd, grid.h:514
	if (OP == OP_XPAY) return b + f * a;               // UpdateSearchVec, conjugategrad.cpp:195
	if (OP == OP_ADD) return a + b;
	if (OP == OP_SUB) return a - b;
	if (OP == OP_MUL) return a * b;
	if (OP == OP_SAFEDIV) return (b != 0.f) ? (a / b) : a;  // general.h:150
	return a;
}
template <int OP>
__global__ void __launch_bounds__(BLOCK) k_binary(int64_t n, float* __restrict__ a, const float* __restrict__ b, float f) {
	const int64_t n4 = n >> 2;
	float4* a4 = (float4*)a;
	const float4* b4 = (const float4*)b;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
		float4 v = (OP == OP_COPY) ? make_float4(0, 0, 0, 0) : a4[i];
		const float4 w = b4[i];
		v.x = op2<OP>(v.x, w.x, f);
		v.y = op2<OP>(v.y, w.y, f);
		v.z = op2<OP>(v.z, w.z, f);
		v.w = op2<OP>(v.w, w.w, f);
		a4[i] = v;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
		const int64_t i = (n4 << 2) + threadIdx.x;
		a[i] = op2<OP>(OP == OP_COPY ? 0.f : a[i], b[i], f);
	}
}
template <int OP>
__global__ void __launch_bounds__(BLOCK) k_unary_scalar(int64_t n, float* a, float p, float q) {
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		a[i] = op1<OP>(OP == OP_FILL ? 0.f : a[i], p, q);
}
template <int OP>
__global__ void __launch_bounds__(BLOCK) k_binary_scalar(int64_t n, float* a, const float* b, float f) {
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		a[i] = op2<OP>(OP == OP_COPY ? 0.f : a[i], b[i], f);
}
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int OP>
static int run_unary(int64_t n, float* a, float p, float q, void* stream) {
	if (n <= 0) return 0;
	if (al16(a))
		hipLaunchKernelGGL(k_unary<OP>, dim3(blocks_for(n >> 2, BLOCK, 2048)), dim3(BLOCK), 0, (hipStream_t)stream, n, a, p, q);
	else
		hipLaunchKernelGGL(k_unary_scalar<OP>, dim3(blocks_for(n, BLOCK, 2048)), dim3(BLOCK), 0, (hipStream_t)stream, n, a, p, q);
	MF_LAUNCH_CHECK();
	return 0;
}
template <int OP>
static int run_binary(int64_t n, float* a, const float* b, float f, void* stream) {
	if (n <= 0) return 0;
	if (al16(a) && al16(b))
		hipLaunchKernelGGL(k_binary<OP>, dim3(blocks_for(n >> 2, BLOCK, 2048)), dim3(BLOCK), 0, (hipStream_t)stream, n, a, b, f);
	else
		hipLaunchKernelGGL(k_binary_scalar<OP>, dim3(blocks_for(n, BLOCK, 2048)), dim3(BLOCK), 0, (hipStream_t)stream, n, a, b, f);
	MF_LAUNCH_CHECK();
	return 0;
}

// ---------------------------------------------------------------------------------------------------------
// reductions: per-block partials in fixed order -> one finishing block; deterministic run to run
// ---------------------------------------------------------------------------------------------------------
// mode 0: sum of fp32 products a*b accumulated in fp64 (GridDotProduct, conjugategrad.cpp:175-178)
// mode 1: sum of squares of the value converted to fp64 (GridSumSqr, commonkernels.h:32-35)
template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_dot_partials(int64_t n, const float* __restrict__ a, const float* __restrict__ b,
                                                        double* __restrict__ partials) {
	double acc = 0.0;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		if (MODE == 0) {
			const float pr = a[i] * b[i];
			acc += (double)pr;
		} else {
			const double v = (double)a[i];
			acc += v * v;
		}
	}
	acc = block_sum(acc);
	if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(BLOCK) k_sum_finish(int nb, const double* __restrict__ partials, double* __restrict__ out) {
	double acc = strided_sum(partials, nb);
	acc = block_sum(acc);
	if (threadIdx.x == 0) *out = acc;
}
// CompMinReal / CompMaxReal, grid.cpp:185-196
__global__ void __launch_bounds__(BLOCK) k_minmax_partials(int64_t n, const float* __restrict__ a, float* __restrict__ partials) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const float v = a[i];
		lo = fminf(lo, v);
		hi = fmaxf(hi, v);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		partials[2 * blockIdx.x] = lo;
		partials[2 * blockIdx.x + 1] = hi;
	}
}
__global__ void __launch_bounds__(BLOCK) k_minmax_finish(int nb, const float* __restrict__ partials, float* __restrict__ out) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int i = threadIdx.x; i < nb; i += blockDim.x) {
		lo = fminf(lo, partials[2 * i]);
		hi = fmaxf(hi, partials[2 * i + 1]);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		out[0] = lo;
		out[1] = hi;
	}
}
// CompMaxVec, grid.cpp:218-224 : max of normSquare (x*x + y*y + z*z, vectorbase.h:392-395)
__global__ void __launch_bounds__(BLOCK) k_maxnormsq_partials(int64_t n, const float* __restrict__ a, float* __restrict__ partials) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const float x = a[i], y = a[n + i], z = a[2 * n + i];
		const float q = x * x + y * y + z * z;
		hi = fmaxf(hi, q);
		lo = fminf(lo, q);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		partials[2 * blockIdx.x] = lo;
		partials[2 * blockIdx.x + 1] = hi;
	}
}
__global__ void __launch_bounds__(BLOCK) k_count_flag_partials(int64_t n, const int32_t* __restrict__ f, int mask, double* __restrict__ partials) {
	double acc = 0.0;
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		if (f[i] & mask) acc += 1.0;
	acc = block_sum(acc);
	if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// device-scalar variants (multi-GPU PCG: no host round trip between a reduction and the vector update that uses it)
template <int OP>
__global__ void __launch_bounds__(BLOCK) k_binary_devf(int64_t n, float* a, const float* b, const float* __restrict__ fdev, float sign) {
	const float f = sign * fdev[0];   // sign is +-1: exact
	for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
		a[i] = op2<OP>(a[i], b[i], f);
}
__global__ void __launch_bounds__(BLOCK) k_maxabs_finish(int nb, const float* __restrict__ partials, float* __restrict__ out) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int i = threadIdx.x; i < nb; i += blockDim.x) {
		lo = fminf(lo, partials[2 * i]);
		hi = fmaxf(hi, partials[2 * i + 1]);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		lo = fabsf(lo);
		hi = fabsf(hi);
		out[0] = lo > hi ? lo : hi;   // max(fabs(amin), fabs(amax)), grid.cpp:359
	}
}

// z-slab PCG scalar steps on gathered per-rank reductions g[world][2] = {max|residual|, dot}: rows are summed / maxed in
// rank order, so every rank computes the same bits (conjugategrad.cpp:250-291 for the formulas)
__global__ void k_slab_alpha(const double* __restrict__ g, int world, const float* __restrict__ sigma, float* __restrict__ alpha,
                             const int32_t* __restrict__ state) {
	if (state && state[0]) {
		alpha[0] = 0.f;
		alpha[1] = -0.f;
		return;
	}
	double acc = 0.0;
	for (int r = 0; r < world; r++) acc += g[2 * r + 1];
	const float dp = (float)acc;
	const float a = (fabs((double)dp) > 0.) ? sigma[0] / dp : 0.f;
	alpha[0] = a;
	alpha[1] = -a;      // nalpha of the CgScalars layout (mf_cg_slab_axpy2)
}
__global__ void k_slab_beta(const double* __restrict__ g, int world, float* __restrict__ sigma, float* __restrict__ beta, float* __restrict__ res,
                            float accuracy, int iter, int32_t* __restrict__ state) {
	if (state && state[0]) return;
	double acc = 0.0, mx = 0.0;
	for (int r = 0; r < world; r++) {
		acc += g[2 * r + 1];
		mx = g[2 * r] > mx ? g[2 * r] : mx;
	}
	const float sigmaNew = (float)acc;
	const float rn = (float)mx;
	res[0] = rn;
	beta[0] = sigmaNew / sigma[0];
	sigma[0] = sigmaNew;
	if (state) {
		if (rn < accuracy) {
			state[0] = 1;
			state[1] = iter;
		} else if (!(rn < 1e35f)) {
			state[0] = 2;
			state[1] = iter;
		}
	}
}
__global__ void __launch_bounds__(BLOCK) k_maxabs_finish64(int nb, const float* __restrict__ partials, double* __restrict__ out) {
	float lo = FLT_MAX, hi = -FLT_MAX;
	for (int i = threadIdx.x; i < nb; i += blockDim.x) {
		lo = fminf(lo, partials[2 * i]);
		hi = fmaxf(hi, partials[2 * i + 1]);
	}
	block_minmax(lo, hi);
	if (threadIdx.x == 0) {
		lo = fabsf(lo);
		hi = fabsf(hi);
		out[0] = (double)(lo > hi ? lo : hi);
	}
}

static int read_back(Workspace* ws, const void* dev, size_t bytes, void* host_out, hipStream_t s) {
	MF_HIP(hipMemcpyAsync(ws->host, dev, bytes, hipMemcpyDeviceToHost, s));
	MF_HIP(hipStreamSynchronize(s));
	memcpy(host_out, ws->host, bytes);
	return 0;
}

extern "C" {

const char* mf_last_error(void) { return mf::g_err; }
const char* mf_backend(void) { return "hip"; }
int mf_abi_version(void) { return MF_ABI_VERSION; }
int mf_set_slab_window(int zoff, int gsz) {
	if (gsz < 0 || zoff < 0 || (gsz > 0 && zoff >= gsz)) return fail("invalid slab window %d / %d", zoff, gsz);
	mf::g_slab_zoff = zoff;
	mf::g_slab_gsz = gsz;
	return 0;
}
int mf_set_slab_window_source(int zoff, int gsz) {
	if (gsz < 0 || zoff < 0 || (gsz > 0 && zoff >= gsz)) return fail("invalid source slab window %d / %d", zoff, gsz);
	mf::g_slab_src_zoff = zoff;
	mf::g_slab_src_gsz = gsz;
	return 0;
}

int mf_fill_f32(int64_t n, float* a, float v, void* s) { return run_unary<OP_FILL>(n, a, v, 0.f, s); }
int mf_fill_i32(int64_t n, int32_t* a, int32_t v, void* s) {
	float f;
	memcpy(&f, &v, 4);
	return run_unary<OP_FILL>(n, (float*)a, f, 0.f, s);
}
int mf_copy_f32(int64_t n, float* dst, const float* src, void* s) {
	if (n <= 0) return 0;
	MF_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, (hipStream_t)s));
	return 0;
}
int mf_grid_scaled_add(int64_t n, float* me, const float* o, float f, void* s) { return run_binary<OP_AXPY>(n, me, o, f, s); }
int mf_update_search_vec(int64_t n, float* dst, const float* src, float f, void* s) { return run_binary<OP_XPAY>(n, dst, src, f, s); }
int mf_grid_add(int64_t n, float* me, const float* o, void* s) { return run_binary<OP_ADD>(n, me, o, 0.f, s); }
int mf_grid_sub(int64_t n, float* me, const float* o, void* s) { return run_binary<OP_SUB>(n, me, o, 0.f, s); }
int mf_grid_mult(int64_t n, float* me, const float* o, void* s) { return run_binary<OP_MUL>(n, me, o, 0.f, s); }
int mf_grid_safe_divide(int64_t n, float* me, const float* o, void* s) { return run_binary<OP_SAFEDIV>(n, me, o, 0.f, s); }
int mf_grid_stomp(int64_t n, float* a, float th, void* s) { return run_unary<OP_STOMP>(n, a, th, 0.f, s); }
int mf_grid_add_const(int64_t n, float* a, float v, void* s) { return run_unary<OP_ADDC>(n, a, v, 0.f, s); }
int mf_grid_mult_const(int64_t n, float* a, float v, void* s) { return run_unary<OP_MULC>(n, a, v, 0.f, s); }
int mf_grid_clamp(int64_t n, float* a, float lo, float hi, void* s) { return run_unary<OP_CLAMP>(n, a, lo, hi, s); }

int mf_grid_dot(int64_t n, const float* a, const float* b, double* r, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_dot_partials<0>, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, b, ws->partials);
	hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->partials, (double*)ws->scalars);
	MF_LAUNCH_CHECK();
	return read_back(ws, ws->scalars, sizeof(double), r, (hipStream_t)s);
}
int mf_grid_dot_dev(int64_t n, const float* a, const float* b, double* out_dev, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_dot_partials<0>, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, b, ws->partials);
	hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->partials, out_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_grid_max_abs_dev(int64_t n, const float* a, float* out_dev, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_minmax_partials, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, ws->fpartials);
	hipLaunchKernelGGL(k_maxabs_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->fpartials, out_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_grid_max_abs_dev_f64(int64_t n, const float* a, double* out_dev, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_minmax_partials, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, ws->fpartials);
	hipLaunchKernelGGL(k_maxabs_finish64, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->fpartials, out_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_cg_slab_alpha(const double* gathered, int world, const float* sigma_dev, float* alpha_dev, const int32_t* state_dev, void* s) {
	hipLaunchKernelGGL(k_slab_alpha, dim3(1), dim3(1), 0, (hipStream_t)s, gathered, world, sigma_dev, alpha_dev, state_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_cg_slab_beta(const double* gathered, int world, float* sigma_dev, float* beta_dev, float* res_dev, float accuracy, int iter,
                    int32_t* state_dev, void* s) {
	hipLaunchKernelGGL(k_slab_beta, dim3(1), dim3(1), 0, (hipStream_t)s, gathered, world, sigma_dev, beta_dev, res_dev, accuracy, iter, state_dev);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_grid_scaled_add_dev(int64_t n, float* me, const float* o, const float* factor_dev, float sign, void* s) {
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_binary_devf<OP_AXPY>, dim3(blocks_for(n, BLOCK * 4, 4096)), dim3(BLOCK), 0, (hipStream_t)s, n, me, o, factor_dev, sign);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_update_search_vec_dev(int64_t n, float* dst, const float* src, const float* factor_dev, void* s) {
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k_binary_devf<OP_XPAY>, dim3(blocks_for(n, BLOCK * 4, 4096)), dim3(BLOCK), 0, (hipStream_t)s, n, dst, src, factor_dev, 1.f);
	MF_LAUNCH_CHECK();
	return 0;
}
int mf_grid_sum_sqr(int64_t n, const float* a, double* r, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_dot_partials<1>, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, a, ws->partials);
	hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->partials, (double*)ws->scalars);
	MF_LAUNCH_CHECK();
	return read_back(ws, ws->scalars, sizeof(double), r, (hipStream_t)s);
}
int mf_grid_min_max(int64_t n, const float* a, float* mn, float* mx, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_minmax_partials, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, ws->fpartials);
	hipLaunchKernelGGL(k_minmax_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->fpartials, (float*)ws->scalars);
	MF_LAUNCH_CHECK();
	float r[2];
	MF_TRY(read_back(ws, ws->scalars, sizeof(r), r, (hipStream_t)s));
	*mn = r[0];
	*mx = r[1];
	return 0;
}
int mf_grid_max_abs(int64_t n, const float* a, float* r, void* s) {
	float lo, hi;
	MF_TRY(mf_grid_min_max(n, a, &lo, &hi, s));
	lo = fabsf(lo);
	hi = fabsf(hi);
	*r = lo > hi ? lo : hi;  // max(fabs(amin), fabs(amax)), grid.cpp:359
	return 0;
}
int mf_grid_max_abs_vec3(int64_t n, const float* a, float* r, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_maxnormsq_partials, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, a, ws->fpartials);
	hipLaunchKernelGGL(k_minmax_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->fpartials, (float*)ws->scalars);
	MF_LAUNCH_CHECK();
	float v[2];
	MF_TRY(read_back(ws, ws->scalars, sizeof(v), v, (hipStream_t)s));
	*r = sqrtf(v[1]);
	return 0;
}
int mf_count_empty_cells(int64_t n, const int32_t* flags, int32_t* r, void* s) {
	Workspace* ws;
	MF_TRY(get_workspace(&ws));
	const int nb = blocks_for(n, BLOCK * 8, 1024);
	hipLaunchKernelGGL(k_count_flag_partials, dim3(nb), dim3(BLOCK), 0, (hipStream_t)s, n, flags, (int)MF_EMPTY, ws->partials);
	hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(BLOCK), 0, (hipStream_t)s, nb, ws->partials, (double*)ws->scalars);
	MF_LAUNCH_CHECK();
	double d;
	MF_TRY(read_back(ws, ws->scalars, sizeof(double), &d, (hipStream_t)s));
	*r = (int32_t)d;
	return 0;
}

}  // extern "C"
