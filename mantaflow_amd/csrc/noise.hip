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

// The wavelet noise tile (WaveletNoiseField::generateTile, noisefield.cpp:95-186; Cook & DeRose 2005): Gaussian noise minus its
// coarse part, i.e. minus what survives "analyse + decimate, then synthesise" along x, then y, then z -- plus a shifted copy of the
// result to decorrelate it.  Host set-up code; the bits matter (the tile is scene input), so every sum keeps the reference's order
// and precision: the analysis sum is an fp32 accumulation over the 32 taps in ascending order (noisefield.cpp:42-50), the synthesis sum
// adds `0.5 * p * sample` in double and rounds to fp32 after every term (:52-63).
constexpr int TILE_N = 128, TILE_H = TILE_N / 2;

// coarse part of one periodic line of TILE_N samples, in place (`line[q * stride]`)
static void line_keep_coarse(float* line, int stride) {
	float half[TILE_H];
	for (int q = 0; q < TILE_H; q++) {                      // analysis filter centred between samples 2q - 1 and 2q, decimated
		float sum = 0;
		for (int tap = -16; tap < 16; tap++) sum += kA[16 + tap] * line[((2 * q + tap) & (TILE_N - 1)) * stride];
		half[q] = sum;
	}
	float full[TILE_N];
	for (int q = 0; q < TILE_N; q++) {                      // synthesis: four taps {0.25, 0.75, 0.75, 0.25} around q / 2
		float sum = 0;
		for (int tap = -1; tap < 3; tap++) {
			const int src = ((q / 2 + tap) % TILE_H + TILE_H) % TILE_H;
			sum = (float)((double)sum + 0.5 * (double)kP[1 + tap - 0] * (double)half[src]);
		}
		full[q] = sum;
	}
	for (int q = 0; q < TILE_N; q++) line[q * stride] = full[q];
}
void generate_tile(float* tile3, int seed) {
	const int64_t plane = (int64_t)TILE_N * TILE_N, vol = plane * TILE_N, total = 3 * vol;
	MTRand mt((uint32_t)seed);
	for (int64_t q = 0; q < total; q++) tile3[q] = (float)mt.randNorm(0.0, 1.0);
	std::vector<float> coarse(tile3, tile3 + total);
	// along x, then y, then z: every line of the three component volumes; `first` = index of a line's first sample
	const int64_t strides[3] = {1, TILE_N, plane};
	for (int comp = 0; comp < 3; comp++)
		for (int axis = 0; axis < 3; axis++) {
			const int64_t su = strides[(axis + 1) % 3], sv = strides[(axis + 2) % 3];
			for (int u = 0; u < TILE_N; u++)
				for (int v = 0; v < TILE_N; v++) line_keep_coarse(coarse.data() + comp * vol + u * su + v * sv, (int)strides[axis]);
		}
	std::vector<float> detail(total);
	for (int64_t q = 0; q < total; q++) {
		tile3[q] -= coarse[q];
		detail[q] = tile3[q];
	}
	// the decorrelating copy: the sample stored at (slow, mid, fast) = (a, b, c) receives the detail at x = a + 65, y = b + 65,
	// z = c + 65 (an odd offset; the reference fills its temporary in (x, y, z) loop order and adds it linearly -- a transposed
	// shifted copy, noisefield.cpp:161-177)
	const int shift = TILE_H + 1;
	for (int comp = 0; comp < 3; comp++)
		for (int a = 0; a < TILE_N; a++)
			for (int b = 0; b < TILE_N; b++)
				for (int c = 0; c < TILE_N; c++) {
					const int x = (a + shift) & (TILE_N - 1), y = (b + shift) & (TILE_N - 1), z = (c + shift) & (TILE_N - 1);
					tile3[comp * vol + a * plane + b * TILE_N + c] += detail[comp * vol + x + y * TILE_N + z * plane];
				}
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
