// micro-test (gfx950): lane patterns of v_permlane16_swap / v_permlane32_swap and DPP row_ror:8 -- the three "flip one bit of
// lane >> 3" moves the MIC row sweeps use instead of ds_bpermute for the k-neighbour (Gray-coded lane -> k mapping)
//   hipcc --offload-arch=gfx950 -O2 permlane_swap.hip -o permlane_swap.bin && ./permlane_swap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
	const unsigned v = threadIdx.x;
	auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
	auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
	const unsigned d = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
	o[threadIdx.x] = r[0];
	o[64 + threadIdx.x] = r[1];
	o[128 + threadIdx.x] = q[0];
	o[192 + threadIdx.x] = q[1];
	o[256 + threadIdx.x] = d;
}
int main() {
	unsigned* d;
	unsigned h[320];
	if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 1;
	const char* names[5] = {"permlane16_swap[0]", "permlane16_swap[1]", "permlane32_swap[0]", "permlane32_swap[1]", "dpp row_ror:8"};
	for (int a = 0; a < 5; a++) {
		printf("%-20s", names[a]);
		for (int l = 0; l < 64; l += 8) printf(" %2u", h[a * 64 + l]);
		printf("   (source lane of lanes 0, 8, 16, ... 56)\n");
	}
	return 0;
}
