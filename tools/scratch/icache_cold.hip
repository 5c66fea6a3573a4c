// icache_cold.hip -- what straight-line code costs a short kernel: the same 2 048 FP64 FMAs per wave (four independent
// chains) as TOTAL / U trips through a body of U unrolled FMAs, U = 16 ... 2 048: code of 128 B ... 16 KB for the same
// arithmetic.  1 024 waves (one per SIMD), kernel time by HIP events over 200 back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 icache_cold.hip -o icache_cold && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int TOTAL = 2048;

template <int U>
__global__ __launch_bounds__(256) void chain(double *out, double a, double b)
{
	double x0 = threadIdx.x, x1 = x0 + 1.0, x2 = x0 + 2.0, x3 = x0 + 3.0;
#pragma unroll 1
	for (int it = 0; it < TOTAL / U; it++) {
#pragma unroll
		for (int k = 0; k < U / 4; k++) {
			x0 = __builtin_fma(x0, a, b);
			x1 = __builtin_fma(x1, a, b);
			x2 = __builtin_fma(x2, a, b);
			x3 = __builtin_fma(x3, a, b);
		}
		asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)); // keep the trips apart
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = (x0 + x1) + (x2 + x3);
}

template <int U>
static void run(double *out)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int i = 0; i < 20; i++) hipLaunchKernelGGL(chain<U>, dim3(256), dim3(256), 0, 0, out, 1.0000001, 1e-9);
	hipDeviceSynchronize();
	const int K = 200;
	hipEventRecord(e0, 0);
	for (int i = 0; i < K; i++) hipLaunchKernelGGL(chain<U>, dim3(256), dim3(256), 0, 0, out, 1.0000001, 1e-9);
	hipEventRecord(e1, 0);
	hipDeviceSynchronize();
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	std::printf("body of %4d FMAs (%5d B of code), %d trips: %.2f us per launch\n", U, U * 8, TOTAL / U, ms / K * 1e3);
}

int main()
{
	double *out;
	hipMalloc(&out, 256 * 256 * sizeof(double));
	for (int rep = 0; rep < 2; rep++) {
		run<16>(out);
		run<64>(out);
		run<256>(out);
		run<512>(out);
		run<1024>(out);
		run<2048>(out);
	}
	return 0;
}
