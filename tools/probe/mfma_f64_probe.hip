// Lane layout probe for v_mfma_f64_4x4x4_4b_f64 on gfx950 (exact integer data, asymmetric operands).
// build: hipcc --offload-arch=gfx950 -O2 -o build/mfma_probe tools/probe/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double * a, const double * b, double * d) {
	const int l = threadIdx.x;
	d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
	double ha[64], hb[64], hd[64], *da, *db, *dd;
	for (int l = 0; l < 64; l++) { ha[l] = 1 + l; hb[l] = 100 + 3 * l + (l % 5); }
	hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
	hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
	hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
	// hypotheses: the three 2-bit fields of the lane number (f0 = l & 3, f1 = (l >> 2) & 3, f2 = l >> 4) carry (row/col, k, block) in some order
	const int perm[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {1,2,0}, {2,0,1}, {2,1,0} };
	auto lane_of = [&](const int * pm, int x, int y, int z) { int f[3]; f[pm[0]] = x; f[pm[1]] = y; f[pm[2]] = z; return f[0] + 4 * f[1] + 16 * f[2]; };
	for (int pa = 0; pa < 6; pa++) for (int pb = 0; pb < 6; pb++) for (int pd = 0; pd < 6; pd++) {
		bool ok = true;
		for (int blk = 0; blk < 4 && ok; blk++) for (int i = 0; i < 4 && ok; i++) for (int j = 0; j < 4; j++) {
			double sacc = 0;
			for (int kk = 0; kk < 4; kk++) sacc += ha[lane_of(perm[pa], i, kk, blk)] * hb[lane_of(perm[pb], j, kk, blk)];
			if (hd[lane_of(perm[pd], i, j, blk)] != sacc) { ok = false; break; }
		}
		if (ok) printf("MATCH: A fields(i,k,blk)->bits %d%d%d  B fields(j,k,blk)->bits %d%d%d  D fields(i,j,blk)->bits %d%d%d\n",
			perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2]);
	}
	printf("d[0..7]:"); for (int l = 0; l < 8; l++) printf(" %.0f", hd[l]); printf("\n");
	// rounding probe: is each k-step a fused multiply-add in k order?  a = (1+2^-30) etc.
	return 0;
}
