// Where do the wavefronts of a 4-wavefront workgroup land when 380 such workgroups (the column solve's launch at ne30) share
// 256 CUs at two wavefronts per SIMD?  Every wavefront records its HW_ID / XCC_ID and spins long enough for all workgroups
// to be resident together.   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip && ./wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(unsigned * out, long long spin) {
	extern __shared__ double lds[];
	const int wv = threadIdx.y;
	lds[threadIdx.y * 64 + threadIdx.x] = 0.0;
	const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
	const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
	const long long t0 = __builtin_readcyclecounter();
	while (__builtin_readcyclecounter() - t0 < spin) { }
	if (threadIdx.x == 0) { out[(blockIdx.x * 4 + wv) * 2] = hw; out[(blockIdx.x * 4 + wv) * 2 + 1] = xcc; }
}

int main() {
	const int nwg = 380;
	unsigned * d; hipMalloc(&d, nwg * 4 * 2 * sizeof(unsigned));
	hipLaunchKernelGGL(probe, dim3(nwg), dim3(64, 4), 62000, 0, d, 200000000ll / 100);
	hipDeviceSynchronize();
	std::vector<unsigned> h(nwg * 4 * 2);
	hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
	// gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
	std::map<unsigned, std::vector<std::pair<int, int> > > cu;      // key (xcc, se, sh, cu) -> (wg * 4 + wave, simd)
	for (int w = 0; w < nwg * 4; w++) {
		const unsigned hw = h[w * 2], xcc = h[w * 2 + 1] & 0xf;
		const unsigned simd = (hw >> 4) & 3, cuid = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
		cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(std::make_pair(w, (int)simd));
	}
	int hist[16] = { 0 }, shown = 0;
	for (auto & kv : cu) {
		hist[std::min<size_t>(kv.second.size(), 15)]++;
		if (kv.second.size() > 4 && shown < 6) {
			shown++;
			printf("CU %05x:", kv.first);
			for (auto & p : kv.second) printf("  wg%d.w%d->simd%d", p.first / 4, p.first % 4, p.second);
			printf("\n");
		}
	}
	printf("distinct CUs %zu; CUs by resident wavefronts:", cu.size());
	for (int i = 0; i < 16; i++) if (hist[i]) printf("  %d waves: %d", i, hist[i]);
	printf("\n");
	// how often do the two workgroups of a CU put the same wave index on the same SIMD?
	int same = 0, diff = 0;
	for (auto & kv : cu) if (kv.second.size() == 8) {
		int simd_of[2][4]; int wgs[2] = { -1, -1 };
		for (auto & p : kv.second) { int wg = p.first / 4; int s = (wgs[0] == -1 || wgs[0] == wg) ? 0 : 1; wgs[s] = wg; simd_of[s][p.first % 4] = p.second; }
		for (int w = 0; w < 4; w++) (simd_of[0][w] == simd_of[1][w]) ? same++ : diff++;
	}
	printf("CUs with two workgroups: wave index k of both on the same SIMD %d times, on different SIMDs %d times\n", same, diff);
	return 0;
}
