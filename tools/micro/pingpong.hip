// micro-benchmark: one-way latency of an 8-byte tagged granule hand-off between two workgroups (gfx950), by placement
// (same XCD / different XCD, read from HW_REG_XCC_ID) and by cache-scope bits of the store and the polling load.
//   hipcc --offload-arch=gfx950 -O2 pingpong.hip -o pingpong && ./pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define SPIN_LIMIT 2000000
// SS / LS: scope of store / load: 0 = none (wavefront), 1 = sc0 (workgroup), 2 = sc1 (agent), 3 = sc0 sc1 (system)
template <int SS> __device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v) {
	if (SS == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
	if (SS == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
	if (SS == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
	if (SS == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LS> __device__ __forceinline__ unsigned long long ld64(const unsigned long long* p) {
	unsigned long long v;
	if (LS == 0) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	if (LS == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	if (LS == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	if (LS == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	return v;
}
template <int SS, int LS>
__global__ void pingpong(unsigned long long* A, unsigned long long* B, int n, int p1, int* info) {
	const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)p1 ? 1 : -1);
	if (me < 0 || threadIdx.x != 0) return;
	info[me] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // HW_REG_XCC_ID
	int fail = 0;
	for (int i = 1; i <= n && !fail; i++) {
		const unsigned long long v = ((unsigned long long)i << 32) | 1u;
		unsigned long long* mine = me == 0 ? A : B;
		const unsigned long long* theirs = me == 0 ? B : A;
		if (me == 0) st64<SS>(mine, v);
		int spins = 0;
		while ((unsigned)(ld64<LS>(theirs) >> 32) != (unsigned)i)
			if (++spins > SPIN_LIMIT) { fail = 1; break; }
		if (me == 1) st64<SS>(mine, v);
	}
	if (fail) {
		info[2 + me] = 1;
		// release the partner: publish the last tag with full scope
		st64<3>(me == 0 ? A : B, ((unsigned long long)n << 32) | 1u);
	}
}
template <int SS, int LS>
static void run(unsigned long long* A, unsigned long long* B, int* info, int p1) {
	const int n = 2000;
	hipMemset(A, 0, 4096); hipMemset(B, 0, 4096); hipMemset(info, 0, 64);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL((pingpong<SS, LS>), dim3(p1 + 1), dim3(64), 0, 0, A, B + 256, n, p1, info);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	int h[4]; hipMemcpy(h, info, sizeof h, hipMemcpyDeviceToHost);
	printf("store scope %d load scope %d  blocks 0/%d (xcc %d/%d): %s %.3f us one way\n", SS, LS, p1, h[0], h[1],
	       (h[2] || h[3]) ? "TIMED OUT (not coherent)" : "ok", ms * 1e3 / n / 2);
}
int main() {
	unsigned long long *A, *B; int* info;
	hipMalloc(&A, 4096); hipMalloc(&B, 4096); hipMalloc(&info, 64);
	const int partners[3] = {1, 8, 16};
	for (int q = 0; q < 3; q++) {
		const int p1 = partners[q];
		run<2, 2>(A, B, info, p1);
		run<3, 3>(A, B, info, p1);
		run<0, 2>(A, B, info, p1);
		run<1, 2>(A, B, info, p1);
		run<0, 1>(A, B, info, p1);
		run<1, 1>(A, B, info, p1);
	}
	return 0;
}
