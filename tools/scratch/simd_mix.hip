// Scratch: C3's rows kernel runs 1.55x slower per wave when all four SIMDs of a CU hold a wave (>= 1 024 waves) than
// with one or two (256 / 512 waves); plain v_fma_f64 chains (fp64_peak.hip) do not.  Which part of the instruction mix
// is shared across a CU's SIMDs?  Same harness, the loop body varied:
//   0 fma only   1 fma + compare + select   2 fma + a never-taken exec-mask branch   3 separate multiply and add
//   4 fma + 32-bit integer VALU   5 fma + v_max / v_min   6 / 7 fma + an exec-mask branch (skipped / entered)
//   hipcc --offload-arch=gfx950 -O2 simd_mix.hip -o simd_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(64) void mix(int n, double seed, double thr, double *out)
{
	constexpr int CH = 4;
	double a[CH];
	int q[CH];
	for (int c = 0; c < CH; c++) { a[c] = seed + threadIdx.x + c; q[c] = threadIdx.x + c; }
	const double m = 1.0000001, b = 1e-9;
	int sacc = n;
	if (MODE == 12 || MODE == 13) { // what do vector instructions cost under an EMPTY exec mask (no branch around them)?
		double e0 = seed * 0.5;
#pragma unroll 1
		for (int k = 0; k < n; k++) {
#pragma unroll
			for (int r = 0; r < 32; r++) {
				a[0] = __builtin_fma(a[0], m, b);
				unsigned long long vcc, save;
				asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(vcc) : "v"(thr), "v"(a[0]));
				if (MODE == 12)
					asm volatile("s_and_saveexec_b64 %1, %2\n\t"
					             "v_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\t"
					             "v_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\tv_fma_f64 %0, %0, %3, %4\n\t"
					             "s_or_b64 exec, exec, %1"
					             : "+v"(e0), "=&s"(save) : "s"(vcc), "v"(m), "v"(b) : "scc");
				else
					asm volatile("s_and_saveexec_b64 %1, %2\n\t"
					             "s_or_b64 exec, exec, %1"
					             : "+v"(e0), "=&s"(save) : "s"(vcc), "v"(m), "v"(b) : "scc");
			}
		}
		a[1] += e0;
	} else
	if (MODE == 10 || MODE == 11) { // is the branch's cost a latency that independent work can fill?
		double e[6];
		for (int c = 0; c < 6; c++) e[c] = seed * 0.5 + c;
#pragma unroll 1
		for (int k = 0; k < n; k++) {
#pragma unroll
			for (int r = 0; r < 32; r++) {
				a[0] = __builtin_fma(a[0], m, b);
				unsigned long long vcc;
				asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(vcc) : "v"(thr), "v"(a[0]));
				if (MODE == 10) { // the independent work BEFORE the branch
#pragma unroll
					for (int c = 0; c < 6; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(e[c]) : "v"(m), "v"(b));
				}
				if (vcc != 0) { asm volatile("s_nop 0"); a[0] += 1.0; }
				if (MODE == 11) { // the same work AFTER it
#pragma unroll
					for (int c = 0; c < 6; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(e[c]) : "v"(m), "v"(b));
				}
			}
		}
		for (int c = 0; c < 6; c++) a[1] += e[c];
	} else
#pragma unroll 1
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int r = 0; r < 8; r++)
#pragma unroll
			for (int c = 0; c < CH; c++) {
				a[c] = __builtin_fma(a[c], m, b);
				if (MODE == 1) a[c] = a[c] > thr ? thr : a[c];
				if (MODE == 2) { if (a[c] > thr) a[c] = __builtin_sqrt(a[c]) + 1.0; }
				if (MODE == 3) {
#pragma clang fp contract(off)
					a[c] = a[c] * m + b;
				}
				if (MODE == 4) q[c] = (q[c] * 3) ^ (q[c] >> 1);
				if (MODE == 5) {
					double t;
					asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(a[c]), "v"(b));
					asm("v_min_f64 %0, %1, %2" : "=v"(a[c]) : "v"(t), "v"(thr));
				}
				if (MODE == 8) { if (__any(a[c] > thr)) { asm volatile("s_nop 0"); a[c] = a[c] > thr ? a[c] + 1.0 : a[c]; } } // wave-uniform branch, selects inside
				if (MODE == 9) { const bool hit = a[c] > thr; asm volatile("" ::: "memory"); a[c] = hit ? a[c] + 1.0 : a[c]; } // no branch at all: always computed, selected
				if (MODE == 6 || MODE == 7) { if (a[c] > thr) { asm volatile("s_nop 0"); a[c] += 1.0; } } // a real branch: the asm keeps it from being if-converted
			}
	}
	double s = 0;
	for (int c = 0; c < CH; c++) s += a[c] + q[c];
	out[blockIdx.x * 64 + threadIdx.x] = s + sacc;
}

template <int MODE>
static void run(const char *what, double thr = 1e300)
{
	double *out;
	hipMalloc(&out, sizeof(double) * 64 * 4096);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	float ms[4] = {0, 0, 0, 0};
	const int ws[4] = {256, 512, 1024, 2048};
	const int n = 6000;
	for (int i = 0; i < 4; i++) {
		hipLaunchKernelGGL(mix<MODE>, dim3(ws[i]), dim3(64), 0, 0, 1000, 1.0, thr, out);
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(mix<MODE>, dim3(ws[i]), dim3(64), 0, 0, n, 1.0, thr, out);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		hipEventElapsedTime(&ms[i], e0, e1);
	}
	std::printf("%-44s 256 waves %7.3f ms | 512: %5.2fx | 1024: %5.2fx | 2048: %5.2fx\n", what, ms[0], ms[1] / ms[0],
	            ms[2] / ms[0], ms[3] / ms[0]);
	hipFree(out);
}

int main(int argc, char **argv)
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	const int only = argc > 1 ? atoi(argv[1]) : -1;
	if (only < 0 || only == 0) run<0>("fma only");
	if (only < 0 || only == 1) run<1>("fma + compare + select");
	if (only < 0 || only == 2) run<2>("fma + sqrt under a condition (if-converted)");
	if (only < 0 || only == 3) run<3>("multiply and add, not fused");
	if (only < 0 || only == 4) run<4>("fma + 32-bit integer VALU");
	if (only < 0 || only == 5) run<5>("fma + v_max_f64 / v_min_f64");
	if (only < 0 || only == 6) run<6>("fma + exec-mask branch around a block, skipped", 1e300);
	if (only < 0 || only == 7) run<7>("fma + exec-mask branch around a block, entered", -1e300);
	if (only < 0 || only == 8) run<8>("fma + wave-uniform branch (any lane), skipped", 1e300);
	if (only < 0 || only == 9) run<9>("fma + compare + add + select, no branch", 1e300);
	if (only < 0 || only == 10) run<10>("compare, 6 independent fma, THEN the branch", 1e300);
	if (only < 0 || only == 11) run<11>("compare, the branch, then the 6 fma", 1e300);
	if (only < 0 || only == 12) run<12>("compare, saveexec, 8 fma under an EMPTY mask, restore", 1e300);
	if (only < 0 || only == 13) run<13>("compare, saveexec, restore", 1e300);
	return 0;
}
