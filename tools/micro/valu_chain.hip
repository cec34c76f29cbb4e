// Microbenchmark: what one lone wave per SIMD can issue on gfx950 -- cycles per dependent VALU op, per DPP move, per ds_bpermute
// round trip, per independent op; shader clock (s_memtime) against the 100 MHz wall clock.  hipcc --offload-arch=gfx950 -O3 valu_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int KIND>
__global__ void __launch_bounds__(64) k_chain(int n, float a, float* out, long long* cyc, long long* wall) {
	float x = a + threadIdx.x, y = a * 2.f, z = a * 3.f, w = a * 5.f;
	__shared__ float sh[64 * 8];
	sh[threadIdx.x] = x;
	__syncthreads();
	const long long w0 = wall_clock64();
	const long long c0 = clock64();
#pragma unroll 1
	for (int i = 0; i < n; i++) {
#pragma unroll
		for (int u = 0; u < 16; u++) {
			if (KIND == 0) x = x * a;                                    // dependent v_mul
			if (KIND == 1) { x = x * a; y = y * a; z = z * a; w = w * a; }   // 4 independent chains
			if (KIND == 2) x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x * a), 0x111, 0xf, 0xf, true));   // mul + dpp
			if (KIND == 3) x = __shfl_up(x * a, 8, 64);                   // mul + ds_bpermute
			if (KIND == 4) { sh[threadIdx.x] = x * a; x = sh[threadIdx.x ^ 1]; }   // mul + LDS write + read
			if (KIND == 5) x = (x - y) * a;                               // sub + mul dependent
		}
	}
	const long long c1 = clock64();
	const long long w1 = wall_clock64();
	out[blockIdx.x * 64 + threadIdx.x] = x + y + z + w;
	if (threadIdx.x == 0) {
		cyc[blockIdx.x] = c1 - c0;
		wall[blockIdx.x] = w1 - w0;
	}
}

int main() {
	const int nb = 256, n = 2000;
	float* out;
	long long *cyc, *wall;
	hipMalloc(&out, nb * 64 * 4);
	hipMalloc(&cyc, nb * 8);
	hipMalloc(&wall, nb * 8);
	long long hc[256], hw[256];
	const char* names[] = {"dependent v_mul", "4 independent v_mul chains (per 4 ops)", "v_mul + dpp row_shr:1", "v_mul + ds_bpermute", "v_mul + ds_write + ds_read", "v_sub + v_mul"};
	for (int kind = 0; kind < 6; kind++) {
		for (int rep = 0; rep < 2; rep++) {
			switch (kind) {
				case 0: hipLaunchKernelGGL(k_chain<0>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
				case 1: hipLaunchKernelGGL(k_chain<1>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
				case 2: hipLaunchKernelGGL(k_chain<2>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
				case 3: hipLaunchKernelGGL(k_chain<3>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
				case 4: hipLaunchKernelGGL(k_chain<4>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
				case 5: hipLaunchKernelGGL(k_chain<5>, dim3(nb), dim3(64), 0, 0, n, 1.0001f, out, cyc, wall); break;
			}
			hipDeviceSynchronize();
		}
		hipMemcpy(hc, cyc, nb * 8, hipMemcpyDeviceToHost);
		hipMemcpy(hw, wall, nb * 8, hipMemcpyDeviceToHost);
		const double iters = (double)n * 16;
		printf("%-40s: %.1f s_memtime ticks, %.1f ns per iteration (block 0; wall clock 100 MHz)\n", names[kind], hc[0] / iters, hw[0] * 10.0 / iters);
	}
	return 0;
}
