// Streaming rate of this chip for the access shapes the kernels here use: 8 bytes per lane (one fp64 column value per thread,
// 512 B per wavefront instruction) against 16 bytes per lane, out of place (y = a x), in place (x = a x) and a 5-stream
// combination (y = sum a_m x_m), on arrays far beyond the 256 MB Infinity Cache.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_width tools/probe/stream_width.hip && /tmp/stream_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_scale8(const double * __restrict__ x, double * __restrict__ y, size_t n) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) y[i] = 1.000001 * x[i];
}
__global__ __launch_bounds__(256) void k_scale16(const double2 * __restrict__ x, double2 * __restrict__ y, size_t n2) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n2) { double2 v = x[i]; v.x *= 1.000001; v.y *= 1.000001; y[i] = v; }
}
__global__ __launch_bounds__(256) void k_inplace8(double * x, size_t n) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) x[i] = 1.000001 * x[i];
}
__global__ __launch_bounds__(256) void k_inplace16(double2 * x, size_t n2) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n2) { double2 v = x[i]; v.x *= 1.000001; v.y *= 1.000001; x[i] = v; }
}
__global__ __launch_bounds__(256) void k_comb8(const double * __restrict__ x, double * __restrict__ y, size_t n, size_t stride) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) { double v = 0.0; for (int m = 0; m < 5; m++) v += 0.2 * x[i + m * stride]; y[i] = v; }
}
__global__ __launch_bounds__(256) void k_comb16(const double2 * __restrict__ x, double2 * __restrict__ y, size_t n2, size_t stride2) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n2) { double2 v = { 0.0, 0.0 }; for (int m = 0; m < 5; m++) { const double2 w = x[i + m * stride2]; v.x += 0.2 * w.x; v.y += 0.2 * w.y; } y[i] = v; }
}

// The access shape of the step's kernels: thread = (column, level); a field is L slabs of NS columns, an instance 5 fields; a
// workgroup covers 64 columns x 4 levels (SHAPE 0, the kernels' tiling) or 256 columns x 1 level (SHAPE 1); every thread reads
// NIN fields' values of its (column, level) and writes 5.
template <int NIN, int SHAPE>
__global__ __launch_bounds__(256) void k_slabs(const double * __restrict__ x, double * __restrict__ y, int NS, int L) {
	const int col = SHAPE == 0 ? blockIdx.x * 64 + (threadIdx.x & 63) : blockIdx.x * 256 + threadIdx.x;
	const int k = SHAPE == 0 ? blockIdx.y * 4 + (threadIdx.x >> 6) : blockIdx.y;
	if (k >= L || col >= NS) return;
	double v = 0.0;
#pragma unroll
	for (int f = 0; f < NIN; f++) v += 0.1 * x[((size_t)f * L + k) * NS + col];
#pragma unroll
	for (int g = 0; g < 5; g++) y[((size_t)g * L + k) * NS + col] = v + g;
}

// The same work on a tile-interleaved layout: [level][64-column tile][field][64] -- the fields of a tile and level are one
// contiguous run (NF x 512 B) instead of NF separate slab streams; instances stay separate arrays (NI of them read).
template <int NI>
__global__ __launch_bounds__(256) void k_tiles(const double * __restrict__ x, double * __restrict__ y, int NS, int L, size_t inst) {
	const int tile = blockIdx.x, lane = threadIdx.x & 63;
	const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
	if (k >= L) return;
	const size_t base = (((size_t)k * (NS / 64) + tile) * 5) * 64 + lane;
	double v = 0.0;
#pragma unroll
	for (int i = 0; i < NI; i++)
#pragma unroll
		for (int f = 0; f < 5; f++) v += 0.1 * x[i * inst + base + f * 64];
#pragma unroll
	for (int g = 0; g < 5; g++) y[base + g * 64] = v + g;
}

int main() {
	const size_t n = (size_t)96 << 20;            // 96 Mi doubles = 805 MB per array
	double * x = nullptr, * y = nullptr;
	CHK(hipMalloc((void **)&x, 5 * n * sizeof(double))); CHK(hipMalloc((void **)&y, n * sizeof(double)));      // 4 GB and 0.8 GB: also enough for the slab shapes below (35 fields of 83 MB in, 5 out)
	CHK(hipMemset(x, 0, 5 * n * sizeof(double))); CHK(hipMemset(y, 0, n * sizeof(double)));
	hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
	auto timeit = [&](const char * name, double bytes, auto launch) {
		for (int w = 0; w < 2; w++) launch();
		hipEventRecord(a, 0);
		for (int r = 0; r < 10; r++) launch();
		hipEventRecord(b, 0); hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		printf("%-34s %8.1f us  %6.2f TB/s\n", name, 1e3 * ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
		return 0;
	};
	const double B = (double)n * 8;
	timeit("y = a x      8 B per lane", 2 * B, [&] { hipLaunchKernelGGL(k_scale8, dim3((n + 255) / 256), dim3(256), 0, 0, x, y, n); });
	timeit("y = a x     16 B per lane", 2 * B, [&] { hipLaunchKernelGGL(k_scale16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const double2 *)x, (double2 *)y, n / 2); });
	timeit("x = a x      8 B per lane", 2 * B, [&] { hipLaunchKernelGGL(k_inplace8, dim3((n + 255) / 256), dim3(256), 0, 0, y, n); });
	timeit("x = a x     16 B per lane", 2 * B, [&] { hipLaunchKernelGGL(k_inplace16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (double2 *)y, n / 2); });
	timeit("y = sum5 a x 8 B per lane", 6 * B, [&] { hipLaunchKernelGGL(k_comb8, dim3((n + 255) / 256), dim3(256), 0, 0, x, y, n, n); });
	timeit("y = sum5 a x 16 B per lane", 6 * B, [&] { hipLaunchKernelGGL(k_comb16, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, (const double2 *)x, (double2 *)y, n / 2, n / 2); });
	timeit("hipMemcpyAsync D2D", 2 * B, [&] { hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, 0); });
	// the state's size: one 104 MB instance (fits the Infinity Cache together with its copy)
	const size_t ns = (size_t)13046400;
	timeit("y = a x      8 B, 104 MB arrays", 2.0 * ns * 8, [&] { hipLaunchKernelGGL(k_scale8, dim3((ns + 255) / 256), dim3(256), 0, 0, x, y, ns); });
	timeit("y = a x     16 B, 104 MB arrays", 2.0 * ns * 8, [&] { hipLaunchKernelGGL(k_scale16, dim3((ns / 2 + 255) / 256), dim3(256), 0, 0, (const double2 *)x, (double2 *)y, ns / 2); });
	// ne60-sized slabs so that the arrays are beyond the Infinity Cache: NS = 345 600 columns, L = 30: a field = 83 MB
	{
		const int NS = 345600, L = 30;
		const double fb = (double)NS * L * 8;
		timeit("slabs  5 in 5 out, 64 x 4 levels", 10 * fb, [&] { hipLaunchKernelGGL((k_slabs<5, 0>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L); });
		timeit("slabs  5 in 5 out, 256 x 1 level", 10 * fb, [&] { hipLaunchKernelGGL((k_slabs<5, 1>), dim3(NS / 256, L), dim3(256), 0, 0, x, y, NS, L); });
		timeit("slabs 10 in 5 out, 64 x 4 levels", 15 * fb, [&] { hipLaunchKernelGGL((k_slabs<10, 0>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L); });
		timeit("slabs 10 in 5 out, 256 x 1 level", 15 * fb, [&] { hipLaunchKernelGGL((k_slabs<10, 1>), dim3(NS / 256, L), dim3(256), 0, 0, x, y, NS, L); });
		timeit("slabs 35 in 5 out, 64 x 4 levels", 40 * fb, [&] { hipLaunchKernelGGL((k_slabs<35, 0>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L); });
		timeit("tiles  5 in 5 out (interleaved)", 10 * fb, [&] { hipLaunchKernelGGL((k_tiles<1>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L, (size_t)5 * L * NS); });
		timeit("tiles 10 in 5 out (interleaved)", 15 * fb, [&] { hipLaunchKernelGGL((k_tiles<2>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L, (size_t)5 * L * NS); });
		timeit("tiles 35 in 5 out (interleaved)", 40 * fb, [&] { hipLaunchKernelGGL((k_tiles<7>), dim3(NS / 64, (L + 3) / 4), dim3(256), 0, 0, x, y, NS, L, (size_t)5 * L * NS); });
		timeit("slabs 35 in 5 out, 256 x 1 level", 40 * fb, [&] { hipLaunchKernelGGL((k_slabs<35, 1>), dim3(NS / 256, L), dim3(256), 0, 0, x, y, NS, L); });
	}
	return 0;
}
