// turbulence.hip -- the wavelet-turbulence pieces of scenes/waveletTurbulence.py (BASELINE config 5's scene):
// computeEnergy, vorticityConfinement, computeWaveletCoeffs, applyNoiseVec3.
// Reference: source/plugin/waveletturbulence.cpp, plugin/extforces.cpp:24-43, 409-428, commonkernels.h, noisefield.{h,cpp}.
#include "common.h"
#include <math.h>

using namespace mf;

namespace {

inline unsigned nblk_n(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK > 0 ? (n + BLOCK - 1) / BLOCK : 1); }
#define CELL_IJK(d)                                                               \
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;                \
	if (idx >= (d).n) return;                                                     \
	const int i = (int)(idx % (d).sx);                                            \
	const int j = (int)((idx / (d).sx) % (d).sy);                                 \
	const int k = (int)(idx / ((int64_t)(d).sx * (d).sy));                        \
	(void)i; (void)j; (void)k;
#define INTERIOR(d) (i >= 1 && i < (d).sx - 1 && j >= 1 && j < (d).sy - 1 && (!(d).is3d || (k >= 1 && k < (d).sz - 1)))

// `0.5 * x` with a double literal: exact halving, written as the reference writes it
__device__ __forceinline__ float half_of(float x) { return (float)(0.5 * (double)x); }

// KnApplyComputeEnergy, waveletturbulence.cpp:180-189 (MACGrid::getCentered, grid.h:460-471)
__global__ void __launch_bounds__(BLOCK) k_compute_energy(Dim d, const int32_t* __restrict__ flags, const float* __restrict__ vel, float* __restrict__ energy) {
	CELL_IJK(d)
	float e = 0.f;
	if (flags[idx] & MF_FLUID) {
		const int64_t n = d.n;
		const float v0 = half_of(vel[idx] + vel[idx + 1]), v1 = half_of(vel[n + idx] + vel[n + idx + d.Y]);
		const float v2 = d.is3d ? half_of(vel[2 * n + idx] + vel[2 * n + idx + d.Z]) : 0.f;
		e = half_of(v0 * v0 + v1 * v1 + v2 * v2);
	}
	energy[idx] = e;
}

// GetCentered, commonkernels.h:126-131
__global__ void __launch_bounds__(BLOCK) k_get_centered(Dim d, const float* __restrict__ vel, float* __restrict__ vc) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t n = d.n;
	vc[idx] = half_of(vel[idx] + vel[idx + 1]);
	vc[n + idx] = half_of(vel[n + idx] + vel[n + idx + d.Y]);
	float vz = half_of(vel[2 * n + idx] + 0.f);
	if (d.is3d) vz = (float)((double)vz + 0.5 * (double)vel[2 * n + idx + d.Z]);
	else vz = 0.f;
	vc[2 * n + idx] = vz;
}
// CurlOp, commonkernels.h:38-47
__global__ void __launch_bounds__(BLOCK) k_curl(Dim d, const float* __restrict__ g, float* __restrict__ curl) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	const float *gx = g, *gy = g + n, *gz = g + 2 * n;
	float v0 = 0.f, v1 = 0.f;
	const float v2 = half_of((gy[idx + 1] - gy[idx - 1]) - (gx[idx + Y] - gx[idx - Y]));
	if (d.is3d) {
		v0 = half_of((gz[idx + Y] - gz[idx - Y]) - (gy[idx + Z] - gy[idx - Z]));
		v1 = half_of((gx[idx + Z] - gx[idx - Z]) - (gz[idx + 1] - gz[idx - 1]));
	}
	curl[idx] = v0;
	curl[n + idx] = v1;
	curl[2 * n + idx] = v2;
}
// GridNorm, commonkernels.h:116-118 (norm(), vectorbase.h:385-389)
__global__ void __launch_bounds__(BLOCK) k_grid_norm(Dim d, const float* __restrict__ v, float* __restrict__ nrm) {
	CELL_IJK(d)
	const int64_t n = d.n;
	const float x = v[idx], y = v[n + idx], z = v[2 * n + idx];
	const float l = x * x + y * y + z * z;
	const float eps2 = 1e-6f * 1e-6f;
	float r;
	if (l <= eps2) r = 0.f;
	else r = (fabs((double)l - 1.) < (double)eps2) ? 1.f : sqrtf(l);
	nrm[idx] = r;
}
// KnConfForce, extforces.cpp:410-417 (normalize(), vectorbase.h:421-434; cross(), :362-368)
__global__ void __launch_bounds__(BLOCK)
k_conf_force(Dim d, float* __restrict__ force, const float* __restrict__ nrm, const float* __restrict__ curl, float strength,
             const float* __restrict__ strengthCell) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	float gx = half_of(nrm[idx + 1] - nrm[idx - 1]), gy = half_of(nrm[idx + Y] - nrm[idx - Y]), gz = 0.f;
	if (d.is3d) gz = half_of(nrm[idx + Z] - nrm[idx - Z]);
	const float l = gx * gx + gy * gy + gz * gz;
	const float eps2 = 1e-6f * 1e-6f;
	if (fabs((double)l - 1.) < (double)eps2) {
	} else if (l > eps2) {
		const float fac = (float)(1. / (double)sqrtf(l));
		gx *= fac;
		gy *= fac;
		gz *= fac;
	} else {
		gx = gy = gz = 0.f;
	}
	float str = strength;
	if (strengthCell) str += strengthCell[idx];
	const float cx = curl[idx], cy = curl[n + idx], cz = curl[2 * n + idx];
	force[idx] = str * ((gy * cz) - (gz * cy));
	force[n + idx] = str * ((gz * cx) - (gx * cz));
	force[2 * n + idx] = str * ((gx * cy) - (gy * cx));
}
// KnApplyForceField(additive, !isMAC), extforces.cpp:24-43
__global__ void __launch_bounds__(BLOCK) k_apply_force_field(Dim d, const int32_t* __restrict__ flags, float* __restrict__ vel, const float* __restrict__ force) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t n = d.n, Y = d.Y, Z = d.Z;
	const bool curFluid = flags[idx] & MF_FLUID, curEmpty = flags[idx] & MF_EMPTY;
	if (!curFluid && !curEmpty) return;
	const float fx = half_of(force[idx - 1] + force[idx]), fy = half_of(force[n + idx - Y] + force[n + idx]);
	float fz = 0.f;
	if (d.is3d) fz = half_of(force[2 * n + idx - Z] + force[2 * n + idx]);
	if ((flags[idx - 1] & MF_FLUID) || (curFluid && (flags[idx - 1] & MF_EMPTY))) vel[idx] = vel[idx] + fx;
	if ((flags[idx - Y] & MF_FLUID) || (curFluid && (flags[idx - Y] & MF_EMPTY))) vel[n + idx] = vel[n + idx] + fy;
	if (d.is3d && ((flags[idx - Z] & MF_FLUID) || (curFluid && (flags[idx - Z] & MF_EMPTY)))) vel[2 * n + idx] = vel[2 * n + idx] + fz;
}

// ---- WaveletNoiseField::computeCoefficients, noisefield.cpp:191-297: one thread per grid line, the down- and up-sampling
// filters run serially along the line exactly as in the reference ----
__constant__ float c_aCoeffs[32] = {0.000334, -0.001528, 0.000410, 0.003545, -0.000938, -0.008233, 0.002172, 0.019120,
                                    -0.005040, -0.044412, 0.011655, 0.103311, -0.025936, -0.243780, 0.033979, 0.655340,
                                    0.655340, 0.033979, -0.243780, -0.025936, 0.103311, 0.011655, -0.044412, -0.005040,
                                    0.019120, 0.002172, -0.008233, -0.000938, 0.003546, 0.000410, -0.001528, 0.000334};
__constant__ float c_pCoeffs[4] = {0.25, 0.75, 0.75, 0.25};
__device__ void downsample_neumann(const float* from, float* to, int n, int64_t stride) {
	for (int i = 0; i < n / 2; i++) {
		float acc = 0;
		for (int k = 2 * i - 16; k < 2 * i + 16; k++) {
			const int kk = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
			acc += c_aCoeffs[16 + k - 2 * i] * from[kk * stride];
		}
		to[i * stride] = acc;
	}
}
__device__ void upsample_neumann(const float* from, float* to, int n, int64_t stride) {
	for (int i = 0; i < n; i++) {
		float acc = 0;
		for (int k = i / 2 - 1; k < i / 2 + 3; k++) {
			const int kk = k > n / 2 - 1 ? n / 2 - 1 : (k < 0 ? 0 : k);
			acc = (float)((double)acc + 0.5 * (double)c_pCoeffs[1 + k - i / 2] * (double)from[kk * stride]);
		}
		to[i * stride] = acc;
	}
}
// axis 0: lines along x (one per (y,z)); 1: along y; 2: along z
__global__ void __launch_bounds__(64)
k_wavelet_lines(Dim d, int axis, const float* from, float* t13, float* t23) {   // from may alias t23 (passes 2 and 3)
	const int64_t line = blockIdx.x * (int64_t)64 + threadIdx.x;
	const int64_t sxy = (int64_t)d.sx * d.sy;
	int64_t base, stride, nlines;
	int n;
	if (axis == 0) {
		nlines = (int64_t)d.sy * d.sz;
		base = line * d.sx;
		stride = 1;
		n = d.sx;
	} else if (axis == 1) {
		nlines = (int64_t)d.sx * d.sz;
		base = (line / d.sx) * sxy + (line % d.sx);
		stride = d.sx;
		n = d.sy;
	} else {
		nlines = sxy;
		base = line;
		stride = sxy;
		n = d.sz;
	}
	if (line >= nlines) return;
	downsample_neumann(from + base, t13 + base, n, stride);
	upsample_neumann(t13 + base, t23 + base, n, stride);
}
__global__ void __launch_bounds__(BLOCK) k_wavelet_residual(int64_t n, const float* __restrict__ in, const float* __restrict__ t23, float* __restrict__ t13) {
	const int64_t i = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (i < n) t13[i] = sqrtf(fabsf(in[i] - t23[i]));
}
__global__ void __launch_bounds__(BLOCK) k_wavelet_smooth(Dim d, float* __restrict__ input, const float* __restrict__ t13, float smoothing) {
	CELL_IJK(d)
	if (!INTERIOR(d)) return;
	const int64_t sx = d.sx, sxy = (int64_t)d.sx * d.sy;
	float res = t13[idx - 1] + t13[idx + 1];
	res += t13[idx - sx] + t13[idx + sx];
	if (d.is3d) res += t13[idx - sxy] + t13[idx + sxy];
	input[idx] = res * smoothing;
}

// ---- applyNoiseVec3 ----
struct NoiseParams {
	float gsInv[3], seedOff[3], time, posScale[3], posOffset[3], valOffset, valScale, clamp, clampNeg, clampPos;
};
// WNoiseVec, noisefield.h:210-310
__device__ void wnoise_vec(float p0, float p1, float p2, const float* __restrict__ data, float out[3]) {
	const float p[3] = {p0, p1, p2};
	int mid[3];
	float t[3], w[3][3], dw[3][3], nb[3][3][3];
#pragma unroll
	for (int c = 0; c < 3; c++) {
		mid[c] = (int)ceil((double)(p[c] - 0.5f));
		t[c] = (float)mid[c] - (p[c] - 0.5f);
	}
#pragma unroll
	for (int z = -1; z <= 1; z++)
#pragma unroll
		for (int y = -1; y <= 1; y++)
#pragma unroll
			for (int x = -1; x <= 1; x++) {
				const int xC = (mid[0] + x) & 127, yC = (mid[1] + y) & 127, zC = (mid[2] + z) & 127;
				nb[x + 1][y + 1][z + 1] = data[zC * 128 * 128 + yC * 128 + xC];
			}
#pragma unroll
	for (int c = 0; c < 3; c++) {
		dw[c][0] = -t[c];
		dw[c][2] = (1.f - t[c]);
		dw[c][1] = 2.0f * t[c] - 1.0f;
		w[c][0] = t[c] * t[c] * 0.5f;
		w[c][2] = (1.f - t[c]) * (1.f - t[c]) * 0.5f;
		w[c][1] = 1.f - w[c][0] - w[c][2];
	}
#pragma unroll
	for (int comp = 0; comp < 3; comp++) {
		float result = 0.0f;
#pragma unroll
		for (int z = -1; z <= 1; z++)
#pragma unroll
			for (int y = -1; y <= 1; y++)
#pragma unroll
				for (int x = -1; x <= 1; x++) {
					const float a = (comp == 0) ? dw[0][x + 1] : w[0][x + 1];
					const float b = (comp == 1) ? dw[1][y + 1] : w[1][y + 1];
					const float c = (comp == 2) ? dw[2][z + 1] : w[2][z + 1];
					const float weight = a * b * c;
					result += weight * nb[x + 1][y + 1][z + 1];
				}
		out[comp] = result;
	}
}
// WaveletNoiseField::evaluateVec, noisefield.h:338-364
__device__ void noise_evaluate_vec(const NoiseParams& P, const float* __restrict__ tile, float x, float y, float z, int t, float v[3]) {
	float pos[3] = {x, y, z};
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] *= P.gsInv[c];
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] += P.seedOff[c];
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] += P.time;
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] *= P.posScale[c];
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] += P.posOffset[c];
	wnoise_vec(pos[0], pos[1], pos[2], tile + (int64_t)t * 128 * 128 * 128, v);
#pragma unroll
	for (int c = 0; c < 3; c++) v[c] += P.valOffset;
#pragma unroll
	for (int c = 0; c < 3; c++) v[c] *= P.valScale;
	if (P.clamp != 0.f) {
#pragma unroll
		for (int c = 0; c < 3; c++) {
			if (v[c] < P.clampNeg) v[c] = P.clampNeg;
			if (v[c] > P.clampPos) v[c] = P.clampPos;
		}
	}
}
// (Measured, not kept: one fused evaluation of the six derivatives the curl needs, each tile's pair accumulated as a float2 with
// v_pk_mul_f32 / v_pk_add_f32 and the index / weight set-up shared -- bit-exact, 1026 instead of 1187 VALU instructions in the kernel,
// and 4.88 instead of 3.93 ms per 512^3 call: the packed fp32 operations bought no issue slots here and the three long accumulation
// chains left less to overlap with the 81 gathers.)
// knApplyNoiseVec3, waveletturbulence.cpp:120-154 (uv == NULL)
__global__ void __launch_bounds__(BLOCK)
k_apply_noise_vec3(Dim d, const int32_t* __restrict__ flags, float* __restrict__ target, const float* __restrict__ tile, NoiseParams P,
                   float scale, float scaleSpatial, const float* __restrict__ weight, Dim wd, int interp, float sf0, float sf1, float sf2,
                   const float* __restrict__ uv) {
	CELL_IJK(d)
	if (!(flags[idx] & MF_FLUID)) return;
	float w = 1;
	if (weight) {
		if (!interp) w = weight[idx];
		else w = interpol1(wd, weight, (float)i * sf0, (float)j * sf1, (float)(k + d.zoff) * sf2);
	}
	float pos[3] = {(float)i + 0.5f, (float)j + 0.5f, (float)(k + d.zoff) + 0.5f};   // global cell centre (z-slab window)
	if (uv) {
		if (!interp) {
#pragma unroll
			for (int c = 0; c < 3; c++) pos[c] = uv[c * d.n + idx];
		} else {
			// uv->getInterpolated(Vec3(i,j,k) * sourceFactor), then pos /= sourceFactor (uv coordinates are in the uv grid's space)
			const float sfv[3] = {sf0, sf1, sf2};
#pragma unroll
			for (int c = 0; c < 3; c++)
				pos[c] = interpol1(wd, uv + c * wd.n, (float)i * sf0, (float)j * sf1, (float)(k + d.zoff) * sf2) / sfv[c];
		}
	}
#pragma unroll
	for (int c = 0; c < 3; c++) pos[c] *= scaleSpatial;
	float d0[3], d1[3], d2[3];
	noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 0, d0);
	noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 1, d1);
	noise_evaluate_vec(P, tile, pos[0], pos[1], pos[2], 2, d2);
	const float cu[3] = {d0[1] - d1[2], d2[2] - d0[0], d1[0] - d2[1]};
#pragma unroll
	for (int c = 0; c < 3; c++) target[c * d.n + idx] += cu[c] * scale * w;
}

}  // namespace

extern "C" {

int mf_compute_energy(int sx, int sy, int sz, const int32_t* flags, const float* vel, float* energy, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipLaunchKernelGGL(k_compute_energy, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, vel, energy);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_vorticity_confinement(int sx, int sy, int sz, float* vel, const int32_t* flags, float strength, const float* strengthCell,
                             float* velCenter, float* curl, float* norm, float* force, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	const unsigned nb = nblk_n(d.n);
	MF_HIP(hipMemsetAsync(velCenter, 0, sizeof(float) * 3 * d.n, st));
	MF_HIP(hipMemsetAsync(curl, 0, sizeof(float) * 3 * d.n, st));
	MF_HIP(hipMemsetAsync(force, 0, sizeof(float) * 3 * d.n, st));
	hipLaunchKernelGGL(k_get_centered, dim3(nb), dim3(BLOCK), 0, st, d, vel, velCenter);
	hipLaunchKernelGGL(k_curl, dim3(nb), dim3(BLOCK), 0, st, d, velCenter, curl);
	hipLaunchKernelGGL(k_grid_norm, dim3(nb), dim3(BLOCK), 0, st, d, curl, norm);
	hipLaunchKernelGGL(k_conf_force, dim3(nb), dim3(BLOCK), 0, st, d, force, norm, curl, strength, strengthCell);
	hipLaunchKernelGGL(k_apply_force_field, dim3(nb), dim3(BLOCK), 0, st, d, flags, vel, force);
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_compute_wavelet_coeffs(int sx, int sy, int sz, float* input, float* temp1, float* temp2, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	hipStream_t st = (hipStream_t)stream;
	MF_HIP(hipMemsetAsync(temp1, 0, sizeof(float) * d.n, st));
	MF_HIP(hipMemsetAsync(temp2, 0, sizeof(float) * d.n, st));
	const int64_t lx = (int64_t)sy * sz, ly = (int64_t)sx * sz, lz = (int64_t)sx * sy;
	hipLaunchKernelGGL(k_wavelet_lines, dim3((unsigned)((lx + 63) / 64)), dim3(64), 0, st, d, 0, input, temp1, temp2);
	hipLaunchKernelGGL(k_wavelet_lines, dim3((unsigned)((ly + 63) / 64)), dim3(64), 0, st, d, 1, temp2, temp1, temp2);
	if (d.is3d) hipLaunchKernelGGL(k_wavelet_lines, dim3((unsigned)((lz + 63) / 64)), dim3(64), 0, st, d, 2, temp2, temp1, temp2);
	hipLaunchKernelGGL(k_wavelet_residual, dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d.n, input, temp2, temp1);
	hipLaunchKernelGGL(k_wavelet_smooth, dim3(nblk_n(d.n)), dim3(BLOCK), 0, st, d, input, temp1, d.is3d ? (float)(1. / 6.) : (float)(1. / 4.));
	MF_LAUNCH_CHECK();
	return 0;
}

int mf_apply_noise_vec3(int sx, int sy, int sz, const int32_t* flags, float* target, const float* tile, const float* params,
                        float scale, float scaleSpatial, const float* weight, int wsx, int wsy, int wsz, const float* uv, int usx, int usy,
                        int usz, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
	Dim wd = d;
	int interp = 0;
	float sf[3] = {1.f, 1.f, 1.f};
	if (uv && weight && (usx != wsx || usy != wsy || usz != wsz)) return fail("UV and weight grid have to match!");
	if (uv && !weight) {      // the size of the uv grid decides (waveletturbulence.cpp:161-166)
		wsx = usx;
		wsy = usy;
		wsz = usz;
	}
	if (weight || uv) {
		MF_TRY(check_dim(wsx, wsy, wsz));
		wd = mkdim_src(wsx, wsy, wsz);       // a weight grid of another size lives under its own (source) slab window
		interp = (wd.gsz != d.gsz || wsx != sx || wsy != sy);
		if (!interp && (wsz != sz || wd.zoff != d.zoff)) return fail("applyNoiseVec3: weight grid of the same resolution must share the target's slab window");
		sf[0] = (float)wsx / sx;   // calcGridSizeFactor, grid.h:391-393 (whole-domain sizes)
		sf[1] = (float)wsy / sy;
		sf[2] = (float)wd.gsz / d.gsz;
	}
	NoiseParams P;
	for (int c = 0; c < 3; c++) {
		P.gsInv[c] = params[c];
		P.seedOff[c] = params[3 + c];
		P.posScale[c] = params[7 + c];
		P.posOffset[c] = params[10 + c];
	}
	P.time = params[6];
	P.valOffset = params[13];
	P.valScale = params[14];
	P.clamp = params[15];
	P.clampNeg = params[16];
	P.clampPos = params[17];
	hipLaunchKernelGGL(k_apply_noise_vec3, dim3(nblk_n(d.n)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, target, tile, P, scale, scaleSpatial, weight, wd, interp, sf[0], sf[1], sf[2], uv);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
