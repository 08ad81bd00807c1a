// DPP control semantics on gfx950: row_shl:1 (0x101), row_shr:1 (0x111), row_newbcast:n (0x150 + n)
// build: hipcc --offload-arch=gfx950 -O2 -o build/dpp_probe tools/probe/dpp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int * out) {
	const int l = threadIdx.x;
	out[l] = __builtin_amdgcn_update_dpp(-1, l, 0x101, 0xf, 0xf, false);
	out[64 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x111, 0xf, 0xf, false);
	out[128 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x150 + 3, 0xf, 0xf, false);
	out[192 + l] = __builtin_amdgcn_update_dpp(-1, l, 0x150 + 9, 0xf, 0xf, false);
}
int main() {
	int h[256], * d;
	(void)hipMalloc(&d, sizeof(h));
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	(void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	const char * nm[4] = { "row_shl:1 (0x101)", "row_shr:1 (0x111)", "row_newbcast:3 (0x153)", "row_newbcast:9 (0x159)" };
	for (int t = 0; t < 4; t++) { printf("%-24s:", nm[t]); for (int l = 0; l < 34; l++) printf(" %d", h[64 * t + l]); printf("\n"); }
	return 0;
}
