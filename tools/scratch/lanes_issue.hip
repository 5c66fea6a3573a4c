// Scratch: does a wave with fewer active lanes issue FP64 vector instructions faster?
// C3 holds 256 waves on 1 024 SIMDs, each at the single-wave issue limit; if a 16-lane wave issued a v_fma_f64 in one
// quarter of the time, dealing 16 instances per wave would fill the idle SIMDs AND shorten each wave's step.
// One wave per workgroup; dependent chains (latency) and four independent chains (issue rate); lanes 64 / 32 / 16 / 1
// active under an exec mask; cycles from s_memtime.   hipcc --offload-arch=gfx950 -O2 lanes_issue.hip -o lanes_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ __launch_bounds__(64) void chain(int active, int n, double seed, double *out, long long *cyc)
{
	const int lane = threadIdx.x;
	double a[CHAINS];
	for (int c = 0; c < CHAINS; c++) a[c] = seed + lane + c;
	const double m = 1.0000001, b = 1e-9;
	long long t0 = 0, t1 = 0;
	if (lane < active) { // exec mask: `active` lanes run the loop
		t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
		for (int k = 0; k < n; k++) {
#pragma unroll
			for (int r = 0; r < 16; r++)
#pragma unroll
				for (int c = 0; c < CHAINS; c++) a[c] = __builtin_fma(a[c], m, b);
		}
		t1 = __builtin_amdgcn_s_memtime();
	}
	double s = 0;
	for (int c = 0; c < CHAINS; c++) s += a[c];
	out[blockIdx.x * 64 + lane] = s;
	if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	const int n = 2000, blocks = 1024; // 1024 waves: one per SIMD
	double *out;
	long long *cyc;
	hipMalloc(&out, sizeof(double) * 64 * blocks);
	hipMalloc(&cyc, sizeof(long long) * blocks);
	std::vector<long long> h(blocks);
	for (int grid : {1, 256, 1024}) {
		for (int chains : {1, 4}) {
			for (int active : {64, 32, 16, 8, 1}) {
				for (int rep = 0; rep < 2; rep++) {
					if (chains == 1) hipLaunchKernelGGL(chain<1>, dim3(grid), dim3(64), 0, 0, active, n, 1.0, out, cyc);
					else hipLaunchKernelGGL(chain<4>, dim3(grid), dim3(64), 0, 0, active, n, 1.0, out, cyc);
					hipDeviceSynchronize();
				}
				hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
				long long mn = h[0], mx = h[0];
				for (int i = 0; i < grid; i++) { mn = h[i] < mn ? h[i] : mn; mx = h[i] > mx ? h[i] : mx; }
				const double per = (double)mn / ((double)n * 16 * chains);
				std::printf("waves %4d chains %d active lanes %2d: %.2f cycles per v_fma_f64 (min wave; max wave %.2f)\n", grid,
				            chains, active, per, (double)mx / ((double)n * 16 * chains));
			}
		}
	}
	return 0;
}
