// Microbenchmark: one-way hand-off time between two workgroups through global memory on gfx950, as the MIC sweeps do it (tagged 8-byte
// granules, agent-scope relaxed atomic store / polled atomic load) -- same XCD (blocks b and b+8) vs another XCD (b and b+1), and with
// read-modify-write atomics instead.  hipcc --offload-arch=gfx950 -O3 handoff_pingpong.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__device__ __forceinline__ unsigned long long ld(unsigned long long* p) {
	if (KIND == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (KIND == 1) return __hip_atomic_fetch_add(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <int KIND>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
	if (KIND == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (KIND == 1) __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// blocks `a` and `b` play; every other block exits.  One lane per side (LANES = 1) or a 64-lane window of 8 lines (LANES = 64)
template <int KIND, int LANES>
__global__ void __launch_bounds__(64) k_pingpong(int a, int b, int rounds, unsigned long long* buf, long long* out) {
	const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
	if (me < 0) return;
	if (LANES == 1 && threadIdx.x != 0) return;
	unsigned long long* mine = buf + (me == 0 ? 0 : 1024) + threadIdx.x;
	unsigned long long* theirs = buf + (me == 0 ? 1024 : 0) + threadIdx.x;
	const long long t0 = wall_clock64();
	for (int i = 1; i <= rounds; i++) {
		if (me == 0) {
			st<KIND>(mine, (unsigned long long)i);
			for (;;) {
				const unsigned long long v = ld<KIND>(theirs);
				if (LANES == 1 ? v == (unsigned long long)i : __all(v == (unsigned long long)i)) break;
			}
		} else {
			for (;;) {
				const unsigned long long v = ld<KIND>(theirs);
				if (LANES == 1 ? v == (unsigned long long)i : __all(v == (unsigned long long)i)) break;
			}
			st<KIND>(mine, (unsigned long long)i);
		}
	}
	const long long t1 = wall_clock64();
	if (threadIdx.x == 0 && me == 0) out[0] = t1 - t0;
}

template <int KIND, int LANES>
static void run(const char* name, int a, int b, unsigned long long* buf, long long* out) {
	const int rounds = 2000;
	long long h = 0;
	for (int rep = 0; rep < 2; rep++) {
		hipMemset(buf, 0, 4096 * 8);
		hipLaunchKernelGGL((k_pingpong<KIND, LANES>), dim3(256), dim3(64), 0, 0, a, b, rounds, buf, out);
		hipDeviceSynchronize();
	}
	hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
	printf("%-58s blocks %3d <-> %3d: %.0f ns one way\n", name, a, b, h * 10.0 / rounds / 2);
}

int main() {
	unsigned long long* buf;
	long long* out;
	hipMalloc(&buf, 4096 * 8);
	hipMalloc(&out, 8);
	for (int pair = 0; pair < 4; pair++) {
		const int a = 0, b = pair == 0 ? 8 : (pair == 1 ? 1 : (pair == 2 ? 4 : 128));
		run<0, 1>("agent-scope atomic store / load, 1 lane", a, b, buf, out);
		run<0, 64>("agent-scope atomic store / load, 64 lanes (8 lines)", a, b, buf, out);
		run<1, 1>("atomic exchange / fetch_add(0), 1 lane", a, b, buf, out);
		run<2, 1>("system-scope atomic store / load, 1 lane", a, b, buf, out);
	}
	return 0;
}
