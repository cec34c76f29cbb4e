// noise.hip -- wavelet noise (source/noisefield.{h,cpp}) for the noise-modulated smoke source of scenes/simpleplume.py:
// the 3 x 128^3 noise tile is generated on the host exactly as the reference does (it is set-up work, done once per
// process: MT19937 + Box-Muller in double precision, fp32 filter passes), uploaded, and evaluated on the device by
// densityInflow (KnApplyNoiseInfl, plugin/initplugins.cpp:27-43).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

using namespace mf;

namespace {

// MTRand, util/randomstream.h:40-276 (Mersenne twister MT19937, Wagner's implementation)
struct MTRand {
	uint32_t state[624];
	uint32_t* next;
	int left;
	static uint32_t twist(uint32_t m, uint32_t s0, uint32_t s1) {
		return m ^ (((s0 & 0x80000000u) | (s1 & 0x7fffffffu)) >> 1) ^ ((uint32_t)(-(int32_t)(s1 & 1u)) & 0x9908b0dfu);
	}
	void reload() {
		uint32_t* p = state;
		int i;
		for (i = 624 - 397; i--; ++p) *p = twist(p[397], p[0], p[1]);
		for (i = 397; --i; ++p) *p = twist(p[397 - 624], p[0], p[1]);
		*p = twist(p[397 - 624], p[0], state[0]);
		left = 624;
		next = state;
	}
	explicit MTRand(uint32_t seed) {
		state[0] = seed;
		for (int i = 1; i < 624; i++) state[i] = 1812433253u * (state[i - 1] ^ (state[i - 1] >> 30)) + (uint32_t)i;
		reload();
	}
	uint32_t randInt() {
		if (left == 0) reload();
		--left;
		uint32_t s1 = *next++;
		s1 ^= (s1 >> 11);
		s1 ^= (s1 << 7) & 0x9d2c5680u;
		s1 ^= (s1 << 15) & 0xefc60000u;
		return s1 ^ (s1 >> 18);
	}
	double rand() { return double(randInt()) * (1.0 / 4294967295.0); }
	double randNorm(double mean, double variance) {   // Box-Muller, randomstream.h:129-136
		const double r = sqrt(-2.0 * log(1.0 - (double(randInt()) + 0.5) * (1.0 / 4294967296.0))) * variance;
		const double phi = 2.0 * 3.14159265358979323846264338328 * (double(randInt()) * (1.0 / 4294967296.0));
		return mean + r * cos(phi);
	}
};

const float kA[32] = {0.000334, -0.001528, 0.000410, 0.003545, -0.000938, -0.008233, 0.002172, 0.019120,
                      -0.005040, -0.044412, 0.011655, 0.103311, -0.025936, -0.243780, 0.033979, 0.655340,
                      0.655340, 0.033979, -0.243780, -0.025936, 0.103311, 0.011655, -0.044412, -0.005040,
                      0.019120, 0.002172, -0.008233, -0.000938, 0.003546, 0.000410, -0.001528, 0.000334};
const float kP[4] = {0.25, 0.75, 0.75, 0.25};

// WaveletNoiseField::downsample / upsample, noisefield.cpp:42-63 (the up-sampling sum is formed in double and rounded
// back to float after every term, exactly as `to[i] += 0.5 * pp[..] * from[..]` does with Real = float)
void downsample(const float* from, float* to, int n, int stride) {
	const float* a = &kA[16];
	for (int i = 0; i < n / 2; i++) {
		float acc = 0;
		for (int k = 2 * i - 16; k < 2 * i + 16; k++) acc += a[k - 2 * i] * from[(k & 127) * stride];
		to[i * stride] = acc;
	}
}
void upsample(const float* from, float* to, int n, int stride) {
	const float* pp = &kP[1];
	const int h = n / 2;
	for (int i = 0; i < n; i++) {
		float acc = 0;
		for (int k = i / 2 - 1; k < i / 2 + 3; k++) {
			int m = k % h;
			if (m < 0) m += h;
			acc = (float)((double)acc + 0.5 * (double)pp[k - i / 2] * (double)from[m * stride]);
		}
		to[i * stride] = acc;
	}
}
// WaveletNoiseField::generateTile, noisefield.cpp:95-186
void generate_tile(float* noise3, int seed) {
	const int n = 128;
	const int64_t n3 = (int64_t)n * n * n, n3d = 3 * n3;
	std::vector<float> t1(n3d, 0.f), t2(n3d, 0.f);
	float* temp13 = t1.data();
	float* temp23 = t2.data();
	MTRand mt((uint32_t)seed);
	for (int64_t i = 0; i < n3d; i++) noise3[i] = (float)mt.randNorm(0.0, 1.0);
	for (int t = 0; t < 3; t++) {
		for (int iy = 0; iy < n; iy++)
			for (int iz = 0; iz < n; iz++) {
				const int64_t i = iy * n + (int64_t)iz * n * n + t * n3;
				downsample(&noise3[i], &temp13[i], n, 1);
				upsample(&temp13[i], &temp23[i], n, 1);
			}
		for (int ix = 0; ix < n; ix++)
			for (int iz = 0; iz < n; iz++) {
				const int64_t i = ix + (int64_t)iz * n * n + t * n3;
				downsample(&temp23[i], &temp13[i], n, n);
				upsample(&temp13[i], &temp23[i], n, n);
			}
		for (int ix = 0; ix < n; ix++)
			for (int iy = 0; iy < n; iy++) {
				const int64_t i = ix + iy * n + t * n3;
				downsample(&temp23[i], &temp13[i], n, n * n);
				upsample(&temp13[i], &temp23[i], n, n * n);
			}
	}
	for (int64_t i = 0; i < n3d; i++) noise3[i] -= temp23[i];
	const int offset = 65;   // n/2, made odd (noisefield.cpp:161-162)
	int64_t icnt = 0;
	for (int t = 0; t < 3; t++)
		for (int ix = 0; ix < n; ix++)
			for (int iy = 0; iy < n; iy++)
				for (int iz = 0; iz < n; iz++)
					temp13[icnt++] = noise3[((ix + offset) & 127) + ((iy + offset) & 127) * n + (int64_t)((iz + offset) & 127) * n * n + t * n3];
	for (int64_t i = 0; i < n3d; i++) noise3[i] += temp13[i];
}

struct NoiseParams {
	float gsInv[3], seedOff[3], time, posScale[3], posOffset[3], valOffset, valScale, clamp, clampNeg, clampPos;
};

// WNoise, noisefield.h:163-196: quadratic B-spline over the 27 neighbouring tile entries, x fastest
__device__ __forceinline__ float wnoise(float p0, float p1, float p2, const float* __restrict__ data) {
	float w[3][3];
	int mid[3];
	const float p[3] = {p0, p1, p2};
#pragma unroll
	for (int c = 0; c < 3; c++) {
		mid[c] = (int)ceilf(p[c] - 0.5f);
		const float t = (float)mid[c] - (p[c] - 0.5f);
		w[c][0] = t * t * 0.5f;
		w[c][2] = (1.f - t) * (1.f - t) * 0.5f;
		w[c][1] = 1.f - w[c][0] - w[c][2];
	}
	float result = 0.f;
#pragma unroll
	for (int z = -1; z <= 1; z++)
#pragma unroll
		for (int y = -1; y <= 1; y++)
#pragma unroll
			for (int x = -1; x <= 1; x++) {
				float weight = 1.0f;
				weight *= w[0][x + 1];
				weight *= w[1][y + 1];
				weight *= w[2][z + 1];
				const int xC = (mid[0] + x) & 127, yC = (mid[1] + y) & 127, zC = (mid[2] + z) & 127;
				result += weight * data[(zC * 128 + yC) * 128 + xC];
			}
	return result;
}
// WaveletNoiseField::evaluate, noisefield.h:313-336
__device__ __forceinline__ float noise_evaluate(const NoiseParams& P, const float* __restrict__ tile, float x, float y, float z) {
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
	float v = wnoise(pos[0], pos[1], pos[2], tile);
	v += P.valOffset;
	v *= P.valScale;
	if (P.clamp != 0.f) {
		if (v < P.clampNeg) v = P.clampNeg;
		if (v > P.clampPos) v = P.clampPos;
	}
	return v;
}
// KnApplyNoiseInfl, plugin/initplugins.cpp:27-36
__global__ void __launch_bounds__(BLOCK)
k_density_inflow(Dim d, const int32_t* __restrict__ flags, float* __restrict__ density, const float* __restrict__ sdf,
                 const float* __restrict__ tile, NoiseParams P, float scale, float sigma) {
	const int64_t idx = blockIdx.x * (int64_t)BLOCK + threadIdx.x;
	if (idx >= d.n) return;
	if (!(flags[idx] & MF_FLUID) || sdf[idx] > sigma) return;
	const int i = (int)(idx % d.sx), j = (int)((idx / d.sx) % d.sy), k = (int)(idx / ((int64_t)d.sx * d.sy));
	double f = 1.0 - 0.5 / (double)sigma * (double)(sdf[idx] + sigma);
	if (f < 0.0) f = 0.0;
	else if (f > 1.0) f = 1.0;
	const float factor = (float)f;
	const float target = noise_evaluate(P, tile, (float)i, (float)j, (float)(k + d.zoff)) * scale * factor;   // global plane
	if (density[idx] < target) density[idx] = target;
}

}  // namespace

extern "C" {

int mf_noise_generate_tile(float* tile, int seed, void* stream) {
	const size_t n3d = (size_t)3 * 128 * 128 * 128;
	std::vector<float> host(n3d);
	generate_tile(host.data(), seed);
	MF_HIP(hipMemcpyAsync(tile, host.data(), n3d * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
	MF_HIP(hipStreamSynchronize((hipStream_t)stream));
	return 0;
}

int mf_noise_seed_offset(int fixedSeed, float* out) {
	if (fixedSeed == -1) fixedSeed = 13322223 + 123;   // randomSeed + 123, noisefield.cpp:65-67
	MTRand mt((uint32_t)fixedSeed);
	float v[3];
	for (int c = 0; c < 3; c++) v[c] = (float)mt.rand();   // getVec3: three getReal() in order
	// normalize(), vectorbase.h:421-434
	const float l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
	const float eps2 = 1e-6f * 1e-6f;
	if (fabs((double)l - 1.) < (double)eps2) {
	} else if (l > eps2) {
		const float fac = (float)(1. / (double)sqrtf(l));
		v[0] *= fac;
		v[1] *= fac;
		v[2] *= fac;
	} else {
		v[0] = v[1] = v[2] = 0.f;
	}
	out[0] = v[0];
	out[1] = v[1];
	out[2] = v[2];
	return 0;
}

int mf_density_inflow(int sx, int sy, int sz, const int32_t* flags, float* density, const float* sdf, const float* tile,
                      const float* params, float scale, float sigma, void* stream) {
	MF_TRY(check_dim(sx, sy, sz));
	const Dim d = mkdim(sx, sy, sz);
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
	hipLaunchKernelGGL(k_density_inflow, dim3((unsigned)((d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, d, flags, density, sdf, tile, P, scale, sigma);
	MF_LAUNCH_CHECK();
	return 0;
}

}  // extern "C"
