// Scratch: what FP64 vector rate does this MI355X sustain, and how does it depend on how many SIMDs hold a wave?
// C3 at 16 384 instances (256 waves) takes 0.86 ms, at 65 536 (1 024 waves, one per SIMD) 1.51 ms -- not 0.86: the
// waves slow each other down although each has a SIMD to itself.  This measures the plain thing: W one-wave workgroups of
// independent v_fma_f64 chains, wall clock by HIP events over >= 20 ms, W from 64 to 8 192; s_memtime ticks per FMA
// beside it (if the ticks per FMA stay put while the wall clock per FMA grows, the clock went down).
//   hipcc --offload-arch=gfx950 -O2 fp64_peak.hip -o fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int CHAINS>
__global__ __launch_bounds__(64) void chains(int n, T seed, T *out, long long *ticks)
{
	T a[CHAINS];
	for (int c = 0; c < CHAINS; c++) a[c] = seed + (T)threadIdx.x + (T)c;
	const T m = (T)1.0000001, b = (T)1e-9;
	const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int r = 0; r < 16; r++)
#pragma unroll
			for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fma(a[c], m, b);
	}
	const long long t1 = __builtin_amdgcn_s_memtime();
	T s = 0;
	for (int c = 0; c < CHAINS; c++) s += a[c];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <typename T>
static void sweep(const char *name)
{
	constexpr int CH = 8;
	const int maxw = 8192;
	T *out;
	long long *ticks;
	hipMalloc(&out, sizeof(T) * 64 * maxw);
	hipMalloc(&ticks, sizeof(long long) * maxw);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	std::vector<long long> h(maxw);
	for (int w : {64, 256, 512, 768, 1024, 2048, 4096, 8192}) {
		const int n = 40000; // 40000 * 16 * 8 = 5.1 M FMA instructions per wave: ~12 ms at 5.5 cycles and 2.4 GHz
		hipLaunchKernelGGL((chains<T, CH>), dim3(w), dim3(64), 0, 0, 2000, (T)1.0, out, ticks); // warm-up
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL((chains<T, CH>), dim3(w), dim3(64), 0, 0, n, (T)1.0, out, ticks);
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		hipMemcpy(h.data(), ticks, sizeof(long long) * w, hipMemcpyDeviceToHost);
		long long mn = h[0], mx = h[0];
		for (int i = 0; i < w; i++) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
		const double per_wave = (double)n * 16 * CH;
		const double flops = per_wave * w * 64 * 2;
		std::printf("%s waves %5d: %8.3f ms  %7.2f TFLOP/s  %6.3f ns per wave-FMA (wall / instructions of one wave)  "
		            "s_memtime ticks per FMA min %.3f max %.3f\n", name, w, ms, flops / (ms * 1e-3) / 1e12,
		            ms * 1e6 / per_wave, (double)mn / per_wave, (double)mx / per_wave);
	}
	hipFree(out);
	hipFree(ticks);
}

int main()
{
	sweep<double>("f64");
	sweep<float>("f32");
	return 0;
}
