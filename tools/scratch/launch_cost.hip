// Scratch: host cost of one kernel launch against the size of its argument block (why the explicit filter's light
// kernel takes its options by pointer).  hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Small { double a[8]; };
struct Big { double a[280]; };
__global__ void ks(Small s, double *o) { if (threadIdx.x == 0 && blockIdx.x == 0) o[0] = s.a[1]; }
__global__ void kb(Big s, double *o) { if (threadIdx.x == 0 && blockIdx.x == 0) o[0] = s.a[1]; }
int main()
{
	double *d;
	hipMalloc(&d, 8);
	hipStream_t st;
	hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	Small s = {};
	Big b = {};
	const int K = 5000;
	for (int rep = 0; rep < 2; rep++) {
		for (int which = 0; which < 2; which++) {
			for (int k = 0; k < 100; k++) hipLaunchKernelGGL(ks, dim3(256), dim3(256), 0, st, s, d);
			hipStreamSynchronize(st);
			auto t0 = std::chrono::steady_clock::now();
			for (int k = 0; k < K; k++) {
				if (which == 0) hipLaunchKernelGGL(ks, dim3(256), dim3(256), 0, st, s, d);
				else hipLaunchKernelGGL(kb, dim3(256), dim3(256), 0, st, b, d);
			}
			auto t1 = std::chrono::steady_clock::now();
			hipStreamSynchronize(st);
			auto t2 = std::chrono::steady_clock::now();
			std::printf("%s args: host %.2f us / launch, total %.2f us / launch\n", which ? "2240-byte" : "64-byte",
			            std::chrono::duration<double, std::micro>(t1 - t0).count() / K,
			            std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
		}
	}
	return 0;
}
