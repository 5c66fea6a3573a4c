// Scratch: host-side cost of one direct launch through the three entry points of the HIP runtime, same kernel, same
// 176-byte argument block (the explicit filter's light kernel carries ExplicitOpts + FilterArgs = 176 B):
//   hipLaunchKernelGGL (what the library uses), hipLaunchKernel with a prebuilt argument-pointer array,
//   hipModuleLaunchKernel on the hipFunction_t of the same kernel (hipGetFuncBySymbol) with the packed block as `extra`.
//   hipcc --offload-arch=gfx950 -O2 launch_api.hip -o launch_api
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Blk { double a[21]; };
__global__ void kv(Blk s, double *o) { if (o == nullptr) o[0] = s.a[20]; }

template <class F> static void timeit(const char *name, hipStream_t st, F launch)
{
	for (int k = 0; k < 300; k++) launch();
	(void)hipStreamSynchronize(st);
	double best = 1e30, bestcall = 1e30;
	for (int r = 0; r < 5; r++) {
		const int K = 4000;
		auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < K; k++) launch();
		auto t1 = std::chrono::steady_clock::now();
		(void)hipStreamSynchronize(st);
		auto t2 = std::chrono::steady_clock::now();
		const double call = std::chrono::duration<double, std::micro>(t1 - t0).count() / K, all = std::chrono::duration<double, std::micro>(t2 - t0).count() / K;
		if (all < best) best = all;
		if (call < bestcall) bestcall = call;
	}
	std::printf("%-44s %.3f us per launch to completion, %.3f us in the call (best of 5)\n", name, best, bestcall);
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	double *d;
	(void)hipMalloc(&d, 4096);
	hipStream_t st;
	(void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const dim3 g(256), b(256);
	Blk blk = {};
	timeit("hipLaunchKernelGGL", st, [&] { hipLaunchKernelGGL(kv, g, b, 0, st, blk, d); });
	void *args[2] = {&blk, &d};
	timeit("hipLaunchKernel, prebuilt pointer array", st, [&] { (void)hipLaunchKernel((const void *)kv, g, b, args, 0, st); });
	hipFunction_t fn = nullptr;
	const hipError_t e = hipGetFuncBySymbol(&fn, (const void *)kv);
	std::printf("hipGetFuncBySymbol: %d\n", (int)e);
	if (e == hipSuccess) {
		timeit("hipModuleLaunchKernel, kernelParams", st, [&] { (void)hipModuleLaunchKernel(fn, 256, 1, 1, 256, 1, 1, 0, st, args, nullptr); });
		struct { Blk s; double *o; } packed = {blk, d};
		size_t sz = sizeof(packed);
		void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
		timeit("hipModuleLaunchKernel, packed extra", st, [&] { (void)hipModuleLaunchKernel(fn, 256, 1, 1, 256, 1, 1, 0, st, nullptr, extra); });
	}
	timeit("hipLaunchKernelGGL again", st, [&] { hipLaunchKernelGGL(kv, g, b, 0, st, blk, d); });
	return 0;
}
