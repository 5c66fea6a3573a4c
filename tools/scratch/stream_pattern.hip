// Scratch: what does the memory system give the explicit filter's ACCESS PATTERN at 16 M instances?
// Three 8-byte input columns, two 8-byte output columns written only where the solve succeeded (78 % of the lanes), one
// 4-byte code column -- against the same with unconditional stores, with 16-byte accesses, and against a plain 16-byte
// copy of the same number of bytes.  No arithmetic to speak of: this is the ceiling the filter kernel runs under.
//   hipcc --offload-arch=gfx950 -O2 stream_pattern.hip -o stream_pattern
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void pat8(int64_t B, const double *x0, const double *x1, const double *ud, double *ua,
                                            double *rl, int32_t *rc, int conditional)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= B) return;
	const double a = x0[i], b = x1[i], c = ud[i];
	const bool ok = !conditional || (a * a + b * b < 1.3); // ~78 % of U[-1.2,1.2]^2
	if (ok) {
		ua[i] = a + c;
		rl[i] = b - c;
	}
	rc[i] = ok ? 1 : -1;
}
__global__ __launch_bounds__(256) void pat16(int64_t B, const double *x0, const double *x1, const double *ud, double *ua,
                                             double *rl, int32_t *rc)
{
	const int64_t i = 2 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
	if (i + 1 >= B) return;
	const double2 a = *(const double2 *)(x0 + i), b = *(const double2 *)(x1 + i), c = *(const double2 *)(ud + i);
	*(double2 *)(ua + i) = make_double2(a.x + c.x, a.y + c.y);
	*(double2 *)(rl + i) = make_double2(b.x - c.x, b.y - c.y);
	*(int2 *)(rc + i) = make_int2(1, -1);
}
__global__ __launch_bounds__(256) void copy16(int64_t n16, const float4 *src, float4 *dst)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n16) dst[i] = src[i];
}
// the filter's pattern with the work of several instances per lane issued up front: K instances per lane, strided by the
// grid so that every access stays a contiguous 512 bytes per wave
template <int K>
__global__ __launch_bounds__(256) void pat8k(int64_t B, const double *x0, const double *x1, const double *ud, double *ua,
                                             double *rl, int32_t *rc)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
	double a[K], b[K], c[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		const int64_t i = t + k * stride;
		a[k] = i < B ? x0[i] : 0.0;
		b[k] = i < B ? x1[i] : 0.0;
		c[k] = i < B ? ud[i] : 0.0;
	}
#pragma unroll
	for (int k = 0; k < K; k++) {
		const int64_t i = t + k * stride;
		if (i < B) {
			const bool ok = a[k] * a[k] + b[k] * b[k] < 1.3;
			if (ok) {
				ua[i] = a[k] + c[k];
				rl[i] = b[k] - c[k];
			}
			rc[i] = ok ? 1 : -1;
		}
	}
}

int main()
{
	const int64_t B = 16777216;
	double *x0, *x1, *ud, *ua, *rl;
	int32_t *rc;
	hipMalloc(&x0, 8 * B); hipMalloc(&x1, 8 * B); hipMalloc(&ud, 8 * B); hipMalloc(&ua, 8 * B); hipMalloc(&rl, 8 * B);
	hipMalloc(&rc, 4 * B);
	const int64_t n16 = B * 22 / 16; // the copy moves as many bytes in and out as the filter does: 22 B per instance each way
	float4 *csrc, *cdst;
	hipMalloc(&csrc, 16 * n16);
	hipMalloc(&cdst, 16 * n16);
	hipMemset(csrc, 1, 16 * n16);
	double *h = (double *)malloc(8 * B);
	uint64_t z = 12345;
	for (int64_t i = 0; i < B; i++) { z = z * 6364136223846793005ull + 1442695040888963407ull; h[i] = -1.2 + 2.4 * (double)(z >> 11) / 9007199254740992.0; }
	hipMemcpy(x0, h, 8 * B, hipMemcpyHostToDevice);
	for (int64_t i = 0; i < B; i++) { z = z * 6364136223846793005ull + 1442695040888963407ull; h[i] = -1.2 + 2.4 * (double)(z >> 11) / 9007199254740992.0; }
	hipMemcpy(x1, h, 8 * B, hipMemcpyHostToDevice);
	hipMemcpy(ud, h, 8 * B, hipMemcpyHostToDevice);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const double bytes = 44.0 * B;
	auto time = [&](const char *name, auto launch, double by) {
		for (int k = 0; k < 3; k++) launch();
		hipDeviceSynchronize();
		hipEventRecord(e0, 0);
		const int K = 20;
		for (int k = 0; k < K; k++) launch();
		hipEventRecord(e1, 0);
		hipEventSynchronize(e1);
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		std::printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms / K * 1e3, by / (ms / K * 1e-3) / 1e12);
	};
	const int blk = 256;
	time("8 B columns, unconditional stores", [&] { hipLaunchKernelGGL(pat8, dim3((B + blk - 1) / blk), dim3(blk), 0, 0, B, x0, x1, ud, ua, rl, rc, 0); }, bytes);
	time("8 B columns, stores on 78 % of the lanes", [&] { hipLaunchKernelGGL(pat8, dim3((B + blk - 1) / blk), dim3(blk), 0, 0, B, x0, x1, ud, ua, rl, rc, 1); }, 24.0 * B + (16 * 0.78 + 4) * B);
	time("16 B per lane, unconditional", [&] { hipLaunchKernelGGL(pat16, dim3((B / 2 + blk - 1) / blk), dim3(blk), 0, 0, B, x0, x1, ud, ua, rl, rc); }, bytes);
	time("4 instances per lane up front, conditional", [&] { hipLaunchKernelGGL(pat8k<4>, dim3((B / 4 + blk - 1) / blk), dim3(blk), 0, 0, B, x0, x1, ud, ua, rl, rc); }, 24.0 * B + (16 * 0.78 + 4) * B);
	time("8 instances per lane up front, conditional", [&] { hipLaunchKernelGGL(pat8k<8>, dim3((B / 8 + blk - 1) / blk), dim3(blk), 0, 0, B, x0, x1, ud, ua, rl, rc); }, 24.0 * B + (16 * 0.78 + 4) * B);
	time("plain 16 B copy, 369 MB in + 369 MB out", [&] { hipLaunchKernelGGL(copy16, dim3((n16 + blk - 1) / blk), dim3(blk), 0, 0, n16, (const float4 *)csrc, cdst); }, 32.0 * n16);
	return 0;
}
