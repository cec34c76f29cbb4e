// micro-benchmark (gfx950): hand-off of a 64-byte line of eight {value, tag} granules between two workgroups, the way the MIC row sweeps
// publish a face -- the consumer polls either with vector loads (global_load_dwordx2 sc1, eight lanes) or through the SCALAR cache path
// (one s_load_dwordx16 glc for the whole line).  Optionally three more waves of the consumer's workgroup stream HBM loads the whole
// time (the loader waves of the sweep): the question is whether a scalar poll queues behind them as the vector poll does.
//   hipcc --offload-arch=gfx950 -O2 pingpong_scalar.hip -o pingpong_scalar && ./pingpong_scalar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define SPIN_LIMIT 4000000
typedef unsigned u16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void st_line(unsigned long long* line, int lane, float val, unsigned tag) {
	const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(val);
	if (lane < 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(line + lane), "v"(g) : "memory");
}
// vector poll: lanes 0..7 load one granule each; returns true when all eight tags are `tag`
__device__ __forceinline__ bool poll_vector(const unsigned long long* line, int lane, unsigned tag) {
	unsigned long long v = (unsigned long long)tag << 32;
	if (lane < 8) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(line + lane) : "memory");
	return __all((unsigned)(v >> 32) == tag);
}
// scalar poll: the whole line with one s_load_dwordx16 (glc: miss in the scalar cache)
__device__ __forceinline__ bool poll_scalar(const unsigned long long* line, unsigned tag) {
	u16v v;
	asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(line) : "memory");
	bool ok = true;
#pragma unroll
	for (int q = 0; q < 8; q++) ok = ok && (v[2 * q + 1] == tag);
	return ok;
}
// scalar poll of FOUR consecutive lines in one round trip (what a poller wave of the sweep would do: 64 SGPRs); only line 0 is checked
__device__ __forceinline__ bool poll_scalar4(const unsigned long long* line, unsigned tag) {
	u16v v, w1, w2, w3;
	asm volatile("s_load_dwordx16 %0, %4, 0x0 glc\n\ts_load_dwordx16 %1, %4, 0x40 glc\n\ts_load_dwordx16 %2, %4, 0x80 glc\n\ts_load_dwordx16 %3, %4, 0xc0 glc\n\ts_waitcnt lgkmcnt(0)"
	             : "=&s"(v), "=&s"(w1), "=&s"(w2), "=&s"(w3) : "s"(line) : "memory");
	bool ok = true;
#pragma unroll
	for (int q = 0; q < 8; q++) ok = ok && (v[2 * q + 1] == tag);
	return ok && (w1[1] + w2[1] + w3[1] != 0x12345u);
}
template <int SCALAR>
__global__ void __launch_bounds__(256)
pingpong(unsigned long long* A, unsigned long long* B, int n, int p1, int* info, const float4* big, size_t nbig, int background, float* sink) {
	const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == (unsigned)p1 ? 1 : -1);
	if (me < 0) return;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	__shared__ int stop;
	if (threadIdx.x == 0) stop = 0;
	__syncthreads();
	if (wave > 0) {
		// background HBM stream in both workgroups (the loader waves of a sweep)
		if (!background) return;
		float acc = 0.f;
		size_t i = (size_t)(blockIdx.x * 3 + wave) * 1048576 + lane;
		while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const float4 v = big[(i + (size_t)q * 64) % nbig];
				acc += v.x + v.y + v.z + v.w;
			}
			i += 8 * 64 * 7;
		}
		if (acc == 12345.f) sink[0] = acc;
		return;
	}
	if (lane == 0) info[me] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;   // HW_REG_XCC_ID
	int fail = 0;
	unsigned long long* mine = me == 0 ? A : B;
	const unsigned long long* theirs = me == 0 ? B : A;
	for (int i = 1; i <= n && !fail; i++) {
		if (me == 0) st_line(mine, lane, (float)i, (unsigned)i);
		int spins = 0;
		for (;;) {
			const bool ok = SCALAR == 2 ? poll_scalar4(theirs, (unsigned)i) : (SCALAR ? poll_scalar(theirs, (unsigned)i) : poll_vector(theirs, lane, (unsigned)i));
			if (ok) break;
			if (++spins > SPIN_LIMIT) { fail = 1; break; }
		}
		if (me == 1) st_line(mine, lane, (float)i, (unsigned)i);
	}
	if (fail) {
		if (lane == 0) info[2 + me] = 1;
		st_line(mine, lane, 0.f, (unsigned)n);      // release the partner
	}
	if (lane == 0) __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int SCALAR>
static void run(unsigned long long* A, unsigned long long* B, int* info, int p1, const float4* big, size_t nbig, int background, float* sink, const char* mem) {
	const int n = 2000;
	hipMemset(A, 0, 4096); hipMemset(B, 0, 4096); hipMemset(info, 0, 64);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipEventRecord(e0);
	hipLaunchKernelGGL((pingpong<SCALAR>), dim3(p1 + 1), dim3(256), 0, 0, A, B, n, p1, info, big, nbig, background, sink);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	int h[4]; hipMemcpy(h, info, sizeof h, hipMemcpyDeviceToHost);
	printf("%-12s %-6s poll, %s, blocks 0/%d (xcc %d/%d): %s %.3f us one way\n", mem, SCALAR == 2 ? "scal x4" : (SCALAR ? "scalar" : "vector"), background ? "HBM stream running" : "idle CU        ",
	       p1, h[0], h[1], (h[2] || h[3]) ? "TIMED OUT (not coherent)" : "ok", ms * 1e3 / n / 2);
}
int main() {
	int* info; float* sink; float4* big;
	const size_t nbig = (size_t)1 << 26;      // 1 GiB of float4: far beyond L2 + MALL
	hipMalloc(&info, 64); hipMalloc(&sink, 64); hipMalloc(&big, nbig * sizeof(float4));
	hipMemset(big, 0, nbig * sizeof(float4));
	for (int kind = 0; kind < 2; kind++) {
		unsigned long long *A, *B;
		const char* mem = kind == 0 ? "hipMalloc" : "fine-grained";
		if (kind == 0) { hipMalloc(&A, 4096); hipMalloc(&B, 4096); }
		else {
			if (hipExtMallocWithFlags((void**)&A, 4096, hipDeviceMallocFinegrained) != hipSuccess || hipExtMallocWithFlags((void**)&B, 4096, hipDeviceMallocFinegrained) != hipSuccess) {
				printf("fine-grained allocation not available\n");
				break;
			}
		}
		const int partners[2] = {1, 8};
		for (int q = 0; q < 2; q++)
			for (int bg = 0; bg < 2; bg++) {
				run<0>(A, B, info, partners[q], big, nbig, bg, sink, mem);
				run<1>(A, B, info, partners[q], big, nbig, bg, sink, mem);
				run<2>(A, B, info, partners[q], big, nbig, bg, sink, mem);
			}
	}
	return 0;
}
