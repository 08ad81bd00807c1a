// What the memory system gives the access shape of the column solve's U-row stream (k_vi_pair / vi_back_substitute): G column
// groups, one wavefront each, every group owns a contiguous chunk of ROWS rows x 10 entries x 64 lanes doubles; the forward phase
// appends rows (stores), the back substitution reads them back in reverse with about five rows in flight.  No arithmetic
// here: the rate is what the chip delivers for the pattern itself.  Variants: 8 bytes per lane (one entry per instruction,
// [row][entry][lane], the current layout) against 16 bytes per lane ([row][entry pair][lane][2]); one or two wavefronts per
// workgroup; chunk stride padded by one row so that concurrent wavefronts do not sit on congruent addresses.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ustream_probe tools/probe/ustream_probe.hip && /tmp/ustream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define ROWS 93
#define NE 10

// Reverse reads with the loads written as inline assembly (the compiler otherwise sinks each load to its use and waits for
// it alone): DEPTH rows are in flight, a row is waited for with a counted vmcnt and consumed, then its refill is issued.
#define LD8(dst, ptr) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define LD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
template <int DEPTH>
__global__ __launch_bounds__(64) void k_read8(const double * __restrict__ s, double * __restrict__ out, size_t chunk) {
	const double * b = s + (size_t)blockIdx.x * chunk + threadIdx.x;
	double acc = 0.0;
	double v[DEPTH][NE];
#pragma unroll
	for (int q = 0; q < DEPTH; q++)
#pragma unroll
		for (int c = 0; c < NE; c++) LD8(v[q][c], b + ((size_t)(ROWS - 1 - q) * NE + c) * 64);
	for (int r = ROWS - 1; r >= DEPTH - 1; r -= DEPTH) {
#pragma unroll
		for (int q = 0; q < DEPTH; q++) {
			if (DEPTH == 5) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
			if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
			if (DEPTH == 6) asm volatile("s_waitcnt vmcnt(50)" ::: "memory");
#pragma unroll
			for (int c = 0; c < NE; c++) { asm volatile("" : "+v"(v[q][c])); acc += v[q][c]; }
			const int rn = r - q - DEPTH;
#pragma unroll
			for (int c = 0; c < NE; c++) LD8(v[q][c], b + ((size_t)(rn < 0 ? 0 : rn) * NE + c) * 64);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}
// reverse read, 16 B per lane: [row][5 pairs][64 lanes][2]
template <int DEPTH>
__global__ __launch_bounds__(64) void k_read16(const double2 * __restrict__ s, double * __restrict__ out, size_t chunk2) {
	const double2 * b = s + (size_t)blockIdx.x * chunk2 + threadIdx.x;
	double acc = 0.0;
	double2 v[DEPTH][NE / 2];
#pragma unroll
	for (int q = 0; q < DEPTH; q++)
#pragma unroll
		for (int c = 0; c < NE / 2; c++) LD16(v[q][c], b + ((size_t)(ROWS - 1 - q) * (NE / 2) + c) * 64);
	for (int r = ROWS - 1; r >= DEPTH - 1; r -= DEPTH) {
#pragma unroll
		for (int q = 0; q < DEPTH; q++) {
			if (DEPTH == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
			if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
			if (DEPTH == 10) asm volatile("s_waitcnt vmcnt(45)" ::: "memory");
#pragma unroll
			for (int c = 0; c < NE / 2; c++) { asm volatile("" : "+v"(v[q][c].x), "+v"(v[q][c].y)); acc += v[q][c].x + v[q][c].y; }
			const int rn = r - q - DEPTH;
#pragma unroll
			for (int c = 0; c < NE / 2; c++) LD16(v[q][c], b + ((size_t)(rn < 0 ? 0 : rn) * (NE / 2) + c) * 64);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}
// forward append, 8 B and 16 B per lane (a little arithmetic between rows so that the stores trickle as in the elimination)
__global__ __launch_bounds__(64) void k_write8(double * __restrict__ s, size_t chunk, int spin) {
	double * b = s + (size_t)blockIdx.x * chunk + threadIdx.x;
	double x = threadIdx.x;
	for (int r = 0; r < ROWS; r++) {
		for (int t = 0; t < spin; t++) x = x * 1.0000001 + 0.5;
#pragma unroll
		for (int c = 0; c < NE; c++) b[((size_t)r * NE + c) * 64] = x + c;
	}
}
__global__ __launch_bounds__(64) void k_write16(double2 * __restrict__ s, size_t chunk2, int spin) {
	double2 * b = s + (size_t)blockIdx.x * chunk2 + threadIdx.x;
	double x = threadIdx.x;
	for (int r = 0; r < ROWS; r++) {
		for (int t = 0; t < spin; t++) x = x * 1.0000001 + 0.5;
#pragma unroll
		for (int c = 0; c < NE / 2; c++) b[((size_t)r * (NE / 2) + c) * 64] = make_double2(x + c, x - c);
	}
}

int main() {
	const int G = 794;
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int pad = 0; pad <= 1; pad++) {
		const size_t chunk = (size_t)(ROWS + pad) * NE * 64;      // doubles per group
		double * s, * out;
		CHK(hipMalloc((void **)&s, chunk * G * sizeof(double) * 2));      // twice: alternate halves so that the Infinity Cache does not serve the reads
		CHK(hipMalloc((void **)&out, (size_t)G * 64 * sizeof(double)));
		CHK(hipMemset(s, 0, chunk * G * sizeof(double) * 2));
		const double mb = (double)ROWS * NE * 64 * 8 * G / 1e6;
		auto timeit = [&](const char * name, auto launch) {
			for (int w = 0; w < 3; w++) launch(w & 1);
			hipDeviceSynchronize();
			hipEventRecord(e0);
			const int N = 20;
			for (int i = 0; i < N; i++) launch(i & 1);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			printf("pad %d  %-40s %7.1f us  %6.2f TB/s (%.0f MB)\n", pad, name, 1e3 * ms / N, mb / (1e3 * ms / N), mb);
		};
		timeit("read  8 B/lane, reverse, 3 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read8<3>, dim3(G), dim3(64), 0, 0, s + h * chunk * G, out, chunk); });
		timeit("read  8 B/lane, reverse, 5 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read8<5>, dim3(G), dim3(64), 0, 0, s + h * chunk * G, out, chunk); });
		timeit("read  8 B/lane, reverse, 6 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read8<6>, dim3(G), dim3(64), 0, 0, s + h * chunk * G, out, chunk); });
		timeit("read 16 B/lane, reverse, 3 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read16<3>, dim3(G), dim3(64), 0, 0, (const double2 *)(s + h * chunk * G), out, chunk / 2); });
		timeit("read 16 B/lane, reverse, 5 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read16<5>, dim3(G), dim3(64), 0, 0, (const double2 *)(s + h * chunk * G), out, chunk / 2); });
		timeit("read 16 B/lane, reverse, 10 rows ahead", [&](int h) { hipLaunchKernelGGL(k_read16<10>, dim3(G), dim3(64), 0, 0, (const double2 *)(s + h * chunk * G), out, chunk / 2); });
		for (int spin : { 0, 200, 1000 }) {
			char nm[64];
			snprintf(nm, sizeof nm, "write  8 B/lane, spin %d", spin);
			timeit(nm, [&](int h) { hipLaunchKernelGGL(k_write8, dim3(G), dim3(64), 0, 0, s + h * chunk * G, chunk, spin); });
			snprintf(nm, sizeof nm, "write 16 B/lane, spin %d", spin);
			timeit(nm, [&](int h) { hipLaunchKernelGGL(k_write16, dim3(G), dim3(64), 0, 0, (double2 *)(s + h * chunk * G), chunk / 2, spin); });
		}
		// write then read back (the column solve's own sequence): does the reverse read find the rows in the Infinity Cache?
		timeit("write 8 then read 8 (same half)", [&](int h) { hipLaunchKernelGGL(k_write8, dim3(G), dim3(64), 0, 0, s, chunk, 0); hipLaunchKernelGGL(k_read8<5>, dim3(G), dim3(64), 0, 0, s, out, chunk); });
		timeit("write 16 then read 16 (same half)", [&](int h) { hipLaunchKernelGGL(k_write16, dim3(G), dim3(64), 0, 0, (double2 *)s, chunk / 2, 0); hipLaunchKernelGGL(k_read16<5>, dim3(G), dim3(64), 0, 0, (const double2 *)s, out, chunk / 2); });
		CHK(hipFree(s)); CHK(hipFree(out));
	}
	return 0;
}
