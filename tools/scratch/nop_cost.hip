// Scratch: the compiler puts `s_nop 0` between some back-to-back dependent FP64 instructions of the trajectory loops
// (five per Euler step of the pendulum).  What does one cost a lone wave?  One wave per SIMD (256 workgroups of 64),
// chains of v_fma_f64: (a) one dependent chain, (b) the same with an s_nop 0 after every fma, (c) four independent
// chains, (d) the same with an s_nop 0 after every fma, (e) dependent chain with an independent fma in place of the nop.
//   hipcc --offload-arch=gfx950 -O2 nop_cost.hip -o nop_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(int n, double *out)
{
	double a = 1.0 + threadIdx.x, b = 2.0, c = 3.0, d = 4.0, e = 5.0;
	const double m = 1.0000001, q = 1e-9;
#pragma unroll 1
	for (int it = 0; it < n; it++) {
#pragma unroll
		for (int r = 0; r < 32; r++) {
			if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(q));
			if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2\n\ts_nop 0" : "+v"(a) : "v"(m), "v"(q));
			if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
			                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
			if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %4, %5\n\ts_nop 0\n\tv_fma_f64 %1, %1, %4, %5\n\ts_nop 0\n\tv_fma_f64 %2, %2, %4, %5\n\ts_nop 0\n\tv_fma_f64 %3, %3, %4, %5\n\ts_nop 0"
			                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(q));
			if (MODE == 5) asm volatile("s_mov_b32 s20, 0x12345678\n\ts_mov_b32 s21, 0x3ff12345\n\tv_fma_f64 %0, %0, s[20:21], %4\n\ts_mov_b32 s20, 0x22345678\n\ts_mov_b32 s21, 0x3ff22345\n\tv_fma_f64 %1, %1, s[20:21], %4\n\ts_mov_b32 s20, 0x32345678\n\ts_mov_b32 s21, 0x3ff32345\n\tv_fma_f64 %2, %2, s[20:21], %4\n\ts_mov_b32 s20, 0x42345678\n\ts_mov_b32 s21, 0x3ff42345\n\tv_fma_f64 %3, %3, s[20:21], %4"
			                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(q) : "s20", "s21");
			if (MODE == 6) asm volatile("v_fma_f64 %0, %0, %4, %8\n\tv_fma_f64 %1, %1, %5, %8\n\tv_fma_f64 %2, %2, %6, %8\n\tv_fma_f64 %3, %3, %7, %8"
			                            : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(m + 1e-9), "v"(m + 2e-9), "v"(m + 3e-9), "v"(q));
			if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %2, %3\n\tv_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(e) : "v"(m), "v"(q));
		}
	}
	out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e;
}
template <int MODE> static void run(const char *name, int fmas_per_iter, double *d)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	const int n = 20000;
	hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 0, 0, 100, d);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 0, 0, n, d);
	(void)hipEventRecord(e1, 0);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	std::printf("%-58s %.3f ms: %.2f ns per v_fma_f64 (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / ((double)n * 32 * fmas_per_iter),
	            ms * 1e6 / ((double)n * 32 * fmas_per_iter) * 2.4);
}
int main()
{
	double *d;
	(void)hipMalloc(&d, 256 * 64 * 8);
	run<0>("one dependent chain", 1, d);
	run<1>("one dependent chain, s_nop 0 after every fma", 1, d);
	run<2>("four independent chains", 4, d);
	run<3>("four independent chains, s_nop 0 after every fma", 4, d);
	run<4>("dependent chain + one independent fma in between (per pair)", 2, d);
	run<5>("four independent chains, each constant a 64-bit literal (s_mov x2)", 4, d);
	run<6>("four independent chains, the four constants in VGPRs", 4, d);
	return 0;
}
