// Microbenchmark / feasibility: global_load_lds_dwordx4 on gfx950 (memory -> LDS without a VGPR destination) as a polling primitive: a
// "pump" wave keeps firing agent-scope loads of a 1 KiB window into LDS, a second wave of the same workgroup watches the tags in LDS.
// One-way hand-off time between two workgroups, to compare with pingpong.hip.  hipcc --offload-arch=gfx950 -O3 lds_dma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// block `a` writes round i into its window (64 x 16 B), block `b` answers; each side = 2 waves: wave 0 watches LDS + stores, wave 1 pumps
__global__ void __launch_bounds__(128) k_dma_pingpong(int a, int b, int rounds, int nap, unsigned long long* buf, long long* out) {
	const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
	if (me < 0) return;
	__shared__ __attribute__((aligned(16))) unsigned long long sG[128];     // 64 lanes x 16 B
	__shared__ int s_stop;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	unsigned long long* mine = buf + (me == 0 ? 0 : 1024);
	unsigned long long* theirs = buf + (me == 0 ? 1024 : 0);
	if (threadIdx.x == 0) s_stop = 0;
	if (threadIdx.x < 128) sG[threadIdx.x] = 0;
	__syncthreads();
	if (wave == 1) {
		// pump: lane i loads 16 B at theirs + 2 i, lands at sG + 2 i (the LDS address of a lane is base + 16 * lane)
		int n = 0;
		while (*(volatile int*)&s_stop == 0 && n < (1 << 22)) {
			__builtin_amdgcn_global_load_lds((glb_ptr)(theirs + 2 * lane), (lds_ptr)sG, 16, 0, 2 /* sc1 */);
			__builtin_amdgcn_s_waitcnt(0x0f70 | 4);     // vmcnt(4): at most four samples in flight (gfx9 encoding: vmcnt low bits 3:0)
			__builtin_amdgcn_s_sleep(2);
			if (nap > 2) __builtin_amdgcn_s_sleep(8);
			n++;
		}
		__builtin_amdgcn_s_waitcnt(0x0f70);
		return;
	}
	const long long t0 = wall_clock64();
	for (int i = 1; i <= rounds; i++) {
		if (me == 0) __hip_atomic_store(mine + 2 * lane, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0;
		for (;;) {
			const unsigned long long v = *(volatile unsigned long long*)&sG[2 * lane];
			if (__all(v == (unsigned long long)i) || ++spins > (1 << 22)) break;
		}
		if (me == 1) __hip_atomic_store(mine + 2 * lane, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	const long long t1 = wall_clock64();
	if (lane == 0) {
		*(volatile int*)&s_stop = 1;
		if (me == 0) out[0] = t1 - t0;
	}
}

int main() {
	unsigned long long* buf;
	long long* out;
	hipMalloc(&buf, 4096 * 8);
	hipMalloc(&out, 8);
	const int rounds = 2000;
	for (int nap = 2; nap <= 3; nap++)
		for (int pair = 0; pair < 3; pair++) {
			const int a = 0, b = pair == 0 ? 8 : (pair == 1 ? 1 : 4);
			long long h = 0;
			for (int rep = 0; rep < 2; rep++) {
				hipMemset(buf, 0, 4096 * 8);
				hipLaunchKernelGGL(k_dma_pingpong, dim3(256), dim3(128), 0, 0, a, b, rounds, nap, buf, out);
				hipDeviceSynchronize();
			}
			hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
			printf("LDS-DMA pump (%s), blocks %d <-> %d: %.0f ns one way\n", nap == 2 ? "sleep 2" : "sleep 2+8", a, b, h * 10.0 / rounds / 2);
		}
	return 0;
}
