// Scratch: host-side cost of a DIRECT kernel launch against the shape of its argument block (node_floor.hip: an empty
// kernel with one pointer 2.7 us per launch, with seven arguments 3.9).  Size or count?  Empty kernels, 256 x 256,
// 4 000 launches back to back on one stream; run once with HIP_FORCE_DEV_KERNARG unset and once with =0.
//   hipcc --offload-arch=gfx950 -O2 launch_args.hip -o launch_args
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
template <int N> struct Blk { double a[N]; };
template <int N> __global__ void kv(Blk<N> s, double *o) { if (o == nullptr) o[0] = s.a[N - 1]; }
__global__ void k1(double *o) { if (o == nullptr) o[0] = 1; }
__global__ void k2(double *o, int64_t n) { if (o == nullptr) o[0] = n; }
__global__ void k4(double *o, double *a, double *b, int64_t n) { if (o == nullptr) o[0] = a[0] + b[0] + n; }
__global__ void k7(double *o, double *a, double *b, double *c, double *d, int32_t *e, int64_t n) { if (o == nullptr) o[0] = a[0] + b[0] + c[0] + d[0] + e[0] + n; }
__global__ void k14(double *o, double *a, double *b, double *c, double *d, int32_t *e, int64_t n, double p0, double p1, double p2, double p3, double p4,
                    double p5, double p6) { if (o == nullptr) o[0] = a[0] + b[0] + c[0] + d[0] + e[0] + n + p0 + p1 + p2 + p3 + p4 + p5 + p6; }

template <class F> static void timeit(const char *name, hipStream_t st, F launch)
{
	for (int k = 0; k < 300; k++) launch();
	(void)hipStreamSynchronize(st);
	double best = 1e30;
	for (int r = 0; r < 3; r++) {
		const int K = 4000;
		auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < K; k++) launch();
		auto t1 = std::chrono::steady_clock::now();
		(void)hipStreamSynchronize(st);
		auto t2 = std::chrono::steady_clock::now();
		const double call = std::chrono::duration<double, std::micro>(t1 - t0).count() / K, all = std::chrono::duration<double, std::micro>(t2 - t0).count() / K;
		if (all < best) best = all;
		if (r == 2) std::printf("%-38s %.3f us per launch to completion (best of 3), %.3f us in the call\n", name, best, call);
	}
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	double *d;
	(void)hipMalloc(&d, 4096);
	hipStream_t st;
	(void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	const dim3 g(256), b(256);
	int32_t *e = (int32_t *)d;
	timeit("1 pointer (8 B)", st, [&] { hipLaunchKernelGGL(k1, g, b, 0, st, d); });
	timeit("2 arguments (16 B)", st, [&] { hipLaunchKernelGGL(k2, g, b, 0, st, d, (int64_t)5); });
	timeit("4 arguments (32 B)", st, [&] { hipLaunchKernelGGL(k4, g, b, 0, st, d, d, d, (int64_t)5); });
	timeit("7 arguments (56 B)", st, [&] { hipLaunchKernelGGL(k7, g, b, 0, st, d, d, d, d, d, e, (int64_t)5); });
	timeit("14 arguments (112 B)", st, [&] { hipLaunchKernelGGL(k14, g, b, 0, st, d, d, d, d, d, e, (int64_t)5, 1., 2., 3., 4., 5., 6., 7.); });
	timeit("struct 8 B + pointer", st, [&] { hipLaunchKernelGGL(kv<1>, g, b, 0, st, Blk<1>{}, d); });
	timeit("struct 48 B + pointer", st, [&] { hipLaunchKernelGGL(kv<6>, g, b, 0, st, Blk<6>{}, d); });
	timeit("struct 104 B + pointer", st, [&] { hipLaunchKernelGGL(kv<13>, g, b, 0, st, Blk<13>{}, d); });
	timeit("struct 112 B + pointer", st, [&] { hipLaunchKernelGGL(kv<14>, g, b, 0, st, Blk<14>{}, d); });
	timeit("struct 120 B + pointer (128 in all)", st, [&] { hipLaunchKernelGGL(kv<15>, g, b, 0, st, Blk<15>{}, d); });
	timeit("struct 128 B + pointer", st, [&] { hipLaunchKernelGGL(kv<16>, g, b, 0, st, Blk<16>{}, d); });
	timeit("struct 144 B + pointer", st, [&] { hipLaunchKernelGGL(kv<18>, g, b, 0, st, Blk<18>{}, d); });
	timeit("struct 168 B + pointer (176)", st, [&] { hipLaunchKernelGGL(kv<21>, g, b, 0, st, Blk<21>{}, d); });
	timeit("struct 184 B + pointer (192)", st, [&] { hipLaunchKernelGGL(kv<23>, g, b, 0, st, Blk<23>{}, d); });
	timeit("struct 192 B + pointer", st, [&] { hipLaunchKernelGGL(kv<24>, g, b, 0, st, Blk<24>{}, d); });
	timeit("struct 248 B + pointer (256)", st, [&] { hipLaunchKernelGGL(kv<31>, g, b, 0, st, Blk<31>{}, d); });
	timeit("struct 1016 B + pointer", st, [&] { hipLaunchKernelGGL(kv<127>, g, b, 0, st, Blk<127>{}, d); });
	timeit("1 pointer (8 B) again", st, [&] { hipLaunchKernelGGL(k1, g, b, 0, st, d); });
	return 0;
}
