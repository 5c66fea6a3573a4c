// Scratch: C3's rows kernel takes 1.53x longer per wave with four one-wave workgroups on a CU than with one or two,
// whatever their LDS size; plain FMA chains (fp64_peak.hip) do not.  Is it where the waves are PLACED?  One-wave
// workgroups of independent v_fma_f64 chains again, with (a) nothing else, (b) 40 000 B of dynamic LDS each, (c) 216
// VGPRs each (two waves per SIMD at most), (d) both; each wave also reports the SIMD it ran on (HW_ID).
//   hipcc --offload-arch=gfx950 -O2 wave_place.hip -o wave_place
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>

template <bool BIGV>
__global__ __launch_bounds__(64) void chains(int n, double seed, double *out, unsigned *where)
{
	extern __shared__ double lds[];
	double a[8];
	for (int c = 0; c < 8; c++) a[c] = seed + threadIdx.x + c;
	if (BIGV) asm volatile("v_mov_b32 v215, 0" ::: "v215");
	const double m = 1.0000001, b = 1e-9;
#pragma unroll 1
	for (int k = 0; k < n; k++) {
#pragma unroll
		for (int r = 0; r < 16; r++)
#pragma unroll
			for (int c = 0; c < 8; c++) a[c] = __builtin_fma(a[c], m, b);
	}
	double s = 0;
	for (int c = 0; c < 8; c++) s += a[c];
	if (n < 0) lds[threadIdx.x] = s;
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) {
		unsigned id;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
		where[blockIdx.x] = id;
	}
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	double *out;
	unsigned *where;
	hipMalloc(&out, sizeof(double) * 64 * 2048);
	hipMalloc(&where, sizeof(unsigned) * 2048);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	std::vector<unsigned> h(2048);
	for (int variant = 0; variant < 4; variant++) {
		const bool bigv = variant & 2;
		const size_t ldsb = (variant & 1) ? 40000 : 0;
		if (ldsb) {
			hipFuncSetAttribute((const void *)chains<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
			hipFuncSetAttribute((const void *)chains<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
		}
		for (int w : {256, 512, 768, 1024}) {
			const int n = 8000;
			float ms = 0;
			for (int rep = 0; rep < 2; rep++) {
				hipEventRecord(e0, 0);
				if (bigv) hipLaunchKernelGGL(chains<true>, dim3(w), dim3(64), ldsb, 0, n, 1.0, out, where);
				else hipLaunchKernelGGL(chains<false>, dim3(w), dim3(64), ldsb, 0, n, 1.0, out, where);
				hipEventRecord(e1, 0);
				hipEventSynchronize(e1);
				hipEventElapsedTime(&ms, e0, e1);
			}
			hipMemcpy(h.data(), where, sizeof(unsigned) * w, hipMemcpyDeviceToHost);
			// HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] ...
			std::map<unsigned, int> per_simd, per_cu;
			for (int i = 0; i < w; i++) {
				per_simd[h[i] & 0xfffffff0u & 0x0000fff0u | (h[i] & 0xf0000000u)]++; // simd + cu + sh + se (+ xcc in the top bits, if any)
				per_cu[(h[i] & 0x0000ff00u) | (h[i] & 0xf0000000u)]++;
			}
			int mx = 0, mxcu = 0;
			for (auto &kv : per_simd) mx = kv.second > mx ? kv.second : mx;
			for (auto &kv : per_cu) mxcu = kv.second > mxcu ? kv.second : mxcu;
			std::printf("lds %5zu B, %s VGPRs, %4d waves: %7.3f ms; distinct (se,sh,cu,simd) %4zu, most waves on one %d; distinct CUs %3zu, most on one %d\n",
			            ldsb, bigv ? "216" : "few", w, ms, per_simd.size(), mx, per_cu.size(), mxcu);
		}
	}
	return 0;
}
