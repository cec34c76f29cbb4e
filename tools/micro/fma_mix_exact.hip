#include <hip/hip_runtime.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
__device__ __forceinline__ float mulh(float v, unsigned packed, int hi, float negzero) {
	float r;
	if (hi) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(v), "v"(packed), "v"(negzero));
	else asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(v), "v"(packed), "v"(negzero));
	return r;
}
__global__ void k(const float4* in, float* out, float nz) {
	float4 s = in[threadIdx.x];
	const float pa = __builtin_fabsf(s.y);
	float ti = mulh(s.x, __float_as_uint(s.z), 0, nz);
	float tj = mulh(s.x, __float_as_uint(s.z), 1, nz);
	float tk = mulh(s.x, __float_as_uint(s.w), 0, nz);
	out[3 * threadIdx.x] = ti * pa;
	out[3 * threadIdx.x + 1] = tj * pa;
	out[3 * threadIdx.x + 2] = tk * pa;
}
int main() {
	float4 h[64]; float o[192], *dout; float4* din;
	for (int i = 0; i < 64; i++) {
		unsigned z = ((i & 1) ? 0xBC00u : 0u) | (((i & 2) ? 0xBC00u : 0u) << 16), w = (i & 4) ? 0xBC00u : 0u;
		float v = (i & 8) ? -1.5f * (i + 1) : 1e-40f * (i + 1);   // incl. fp32 denormals
		if (i == 20) v = 0.f; if (i == 21) v = -0.f;
		h[i] = make_float4(v, (i & 16) ? -0.7f : 1.3f, 0.f, 0.f);
		memcpy(&h[i].z, &z, 4); memcpy(&h[i].w, &w, 4);
	}
	hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o); hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
	union { unsigned u; float f; } nz; nz.u = 0x80000000u;
	hipLaunchKernelGGL(k, 1, 64, 0, 0, din, dout, nz.f);
	hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 64; i++) {
		float A[3] = {(i & 1) ? -1.f : 0.f, (i & 2) ? -1.f : 0.f, (i & 4) ? -1.f : 0.f};
		for (int c = 0; c < 3; c++) {
			volatile float t = h[i].x * A[c];
			volatile float r = t * fabsf(h[i].y);
			unsigned a, b; float rr = r; memcpy(&a, &rr, 4); memcpy(&b, &o[3 * i + c], 4);
			if (a != b) { bad++; printf("mismatch i=%d c=%d %08x %08x\n", i, c, a, b); }
		}
	}
	printf("bad=%d\n", bad);
	return bad != 0;
}
