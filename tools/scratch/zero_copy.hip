// Scratch: the host-buffer entry (asif_hip_filter_batch_host) runs the explicit filter in place on page-locked host
// memory: 24 B per instance read and 20 B written across the link.  All 1 024 waves of the 65 536-instance launch load
// together and store together, so the link runs one way at a time.  Does a grid-stride loop on fewer waves (later
// chunks load while earlier ones store) get closer to duplex?  Move-only kernels (three doubles in, two doubles and an
// int out), whole launch timed from the host (launch + hipStreamSynchronize), 50 calls each.
//   hipcc --offload-arch=gfx950 -O2 zero_copy.hip -o zero_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void move_all(const double *x0, const double *x1, const double *ud, double *ua, double *rl, int32_t *rc, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double a = x0[i], b = x1[i], c = ud[i];
	ua[i] = a + c;
	rl[i] = b;
	rc[i] = a > b ? 1 : -1;
}
__global__ void move_loop(const double *x0, const double *x1, const double *ud, double *ua, double *rl, int32_t *rc, int64_t n)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const double a = x0[i], b = x1[i], c = ud[i];
		ua[i] = a + c;
		rl[i] = b;
		rc[i] = a > b ? 1 : -1;
	}
}
__global__ void read_only(const double *x0, const double *x1, const double *ud, double *ua, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double a = x0[i], b = x1[i], c = ud[i];
	if (a + b + c == 12345.0) ua[i] = a;
}
__global__ void write_only(double *ua, double *rl, int32_t *rc, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	ua[i] = 1.0;
	rl[i] = 2.0;
	rc[i] = 1;
}

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
	double *h;
	int32_t *hrc;
	CK(hipHostMalloc((void **)&h, 5 * n * 8, hipHostMallocDefault));
	CK(hipHostMalloc((void **)&hrc, n * 4, hipHostMallocDefault));
	for (int64_t i = 0; i < 5 * n; i++) h[i] = 0.001 * (double)(i % 977);
	double *d;
	int32_t *drc;
	CK(hipHostGetDevicePointer((void **)&d, h, 0));
	CK(hipHostGetDevicePointer((void **)&drc, hrc, 0));
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	auto timeit = [&](const char *name, auto launch) {
		for (int k = 0; k < 5; k++) { launch(); (void)hipStreamSynchronize(st); }
		double best = 1e30, sum = 0;
		const int R = 50;
		for (int r = 0; r < R; r++) {
			auto t0 = std::chrono::steady_clock::now();
			launch();
			(void)hipStreamSynchronize(st);
			const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			best = us < best ? us : best;
			sum += us;
		}
		std::printf("%-44s best %.1f us, mean %.1f us per call\n", name, best, sum / R);
	};
	std::printf("%lld instances: %.2f MB in, %.2f MB out\n", (long long)n, 24.0 * n / 1e6, 20.0 * n / 1e6);
	const unsigned full = (unsigned)((n + 255) / 256);
	timeit("read only, one lane per instance", [&] { hipLaunchKernelGGL(read_only, dim3(full), dim3(256), 0, st, d, d + n, d + 2 * n, d + 3 * n, n); });
	timeit("write only, one lane per instance", [&] { hipLaunchKernelGGL(write_only, dim3(full), dim3(256), 0, st, d + 3 * n, d + 4 * n, drc, n); });
	timeit("move, one lane per instance (as today)", [&] { hipLaunchKernelGGL(move_all, dim3(full), dim3(256), 0, st, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, drc, n); });
	for (unsigned waves : {64u, 128u, 256u, 512u}) {
		char nm[96];
		std::snprintf(nm, sizeof nm, "move, grid-stride loop on %u waves", waves);
		timeit(nm, [&] { hipLaunchKernelGGL(move_loop, dim3(waves), dim3(64), 0, st, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, drc, n); });
	}
	timeit("empty-ish launch + sync (n = 0)", [&] { hipLaunchKernelGGL(move_all, dim3(1), dim3(64), 0, st, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, drc, (int64_t)0); });
	return 0;
}
