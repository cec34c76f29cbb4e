// does an 8-byte global load from a 4-byte aligned address return the two floats?  (gfx950, unaligned access mode)
#include <hip/hip_runtime.h>
#include <stdio.h>
struct __attribute__((packed, aligned(4))) F2u { float a, b; };
__global__ void k(const float* in, float* out, int n) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n - 1) return;
	F2u v = *(const F2u*)(in + i);
	out[2 * i] = v.a;
	out[2 * i + 1] = v.b;
}
int main() {
	const int n = 4096;
	float h[n], *din, *dout, o[2 * n];
	for (int i = 0; i < n; i++) h[i] = (float)i * 0.5f + 1.f;
	hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
	hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, n / 256, 256, 0, 0, din, dout, n);
	hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < n - 1; i++) if (o[2 * i] != h[i] || o[2 * i + 1] != h[i + 1]) bad++;
	printf("unaligned dwordx2: %d mismatches of %d\n", bad, n - 1);
	return bad != 0;
}
