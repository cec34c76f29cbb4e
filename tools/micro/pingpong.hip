// micro-benchmark: one-way latency of an 8-byte tagged granule hand-off between two workgroups (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>  // 0: sc1 (agent, relaxed) store + load   1: atomicExch store + sc1 load   2: system-scope
__global__ void pingpong(unsigned long long* A, unsigned long long* B, int n, int sleepv, long long* cyc) {
	const bool me0 = blockIdx.x == 0;
	if (threadIdx.x != 0) return;
	long long t0 = clock64();
	for (int i = 1; i <= n; i++) {
		unsigned long long v = ((unsigned long long)i << 32) | 1u;
		if (me0) {
			if (MODE == 1) atomicExch(A, v); else __hip_atomic_store(A, v, __ATOMIC_RELAXED, MODE == 2 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
			while ((unsigned)(__hip_atomic_load(B, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != (unsigned)i) { if (sleepv) __builtin_amdgcn_s_sleep(1); }
		} else {
			while ((unsigned)(__hip_atomic_load(A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != (unsigned)i) { if (sleepv) __builtin_amdgcn_s_sleep(1); }
			if (MODE == 1) atomicExch(B, v); else __hip_atomic_store(B, v, __ATOMIC_RELAXED, MODE == 2 ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (me0) *cyc = clock64() - t0;
}
int main(int argc, char** argv) {
	unsigned long long *A, *B; long long* cyc;
	hipMalloc(&A, 4096); hipMalloc(&B, 4096); hipMalloc(&cyc, 8);
	const int n = 2000;
	for (int mode = 0; mode < 3; mode++)
		for (int nblk = 2; nblk <= 64; nblk *= 8)   // extra idle blocks shift which CUs/XCDs host block 1
			for (int sl = 0; sl < 2; sl++) {
				hipMemset(A, 0, 4096); hipMemset(B, 0, 4096);
				hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
				hipEventRecord(e0);
				if (mode == 0) hipLaunchKernelGGL(pingpong<0>, dim3(2), dim3(64), 0, 0, A, B + 256, n, sl, cyc);
				if (mode == 1) hipLaunchKernelGGL(pingpong<1>, dim3(2), dim3(64), 0, 0, A, B + 256, n, sl, cyc);
				if (mode == 2) hipLaunchKernelGGL(pingpong<2>, dim3(2), dim3(64), 0, 0, A, B + 256, n, sl, cyc);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1);
				printf("mode %d sleep %d: %.3f us per round trip (%.3f us one way)\n", mode, sl, ms * 1e3 / n, ms * 1e3 / n / 2);
				if (nblk > 2) break;
			}
	return 0;
}
