// Scratch: what is the shortest a step of 65 536 lanes can be on this part?  Chains of 200 kernel nodes replayed from a
// HIP graph (and the same kernels launched directly, back to back): an empty kernel on 1 / 256 / 1 024 workgroups, and a
// kernel that moves the explicit filter's 60 bytes per lane (three doubles in, two doubles and an int out) with no
// arithmetic.  The explicit light kernel's 2.3-2.5 us per replayed step is to be read against these.
//   hipcc --offload-arch=gfx950 -O2 node_floor.hip -o node_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_k(int *o) { if (o == nullptr && threadIdx.x == 999) o[0] = 1; }
__global__ void move_k(const double *x0, const double *x1, const double *ud, double *ua, double *rl, int32_t *rc, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double a = x0[i], b = x1[i], c = ud[i];
	ua[i] = a + c;
	rl[i] = b;
	rc[i] = a > b ? 1 : -1;
}

__global__ void load_k(const double *x0, const double *x1, const double *ud, double *ua, double *rl, int32_t *rc, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double a = x0[i], b = x1[i], c = ud[i];
	if (a + b + c == 12345.0) ua[i] = a; // never: loads only
}

int main()
{
	setvbuf(stdout, nullptr, _IONBF, 0);
	const int64_t n = 65536;
	double *buf;
	int32_t *rc;
	CK(hipMalloc(&buf, 5 * n * 8));
	CK(hipMalloc(&rc, n * 4));
	CK(hipMemset(buf, 0, 5 * n * 8));
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const int NODES = 200, REPS = 20;
	struct Case { const char *name; int kind; unsigned grid, block; } cases[] = {
		{"empty, 1 x 64", 0, 1, 64},      {"empty, 256 x 256", 0, 256, 256}, {"empty, 1024 x 64", 0, 1024, 64},
		{"move 60 B/lane, 256 x 256", 1, 256, 256}, {"move 60 B/lane, 1024 x 64", 1, 1024, 64},
		{"7 arguments, n = 0, 256 x 256", 2, 256, 256}, {"loads only (24 B/lane), 256 x 256", 3, 256, 256},
	};
	for (const Case &c : cases) {
		auto launch = [&]() {
			if (c.kind == 0) hipLaunchKernelGGL(empty_k, dim3(c.grid), dim3(c.block), 0, st, (int *)buf);
			else if (c.kind == 2) hipLaunchKernelGGL(move_k, dim3(c.grid), dim3(c.block), 0, st, buf, buf + n, buf + 2 * n, buf + 3 * n, buf + 4 * n, rc, (int64_t)0);
			else if (c.kind == 3) hipLaunchKernelGGL(load_k, dim3(c.grid), dim3(c.block), 0, st, buf, buf + n, buf + 2 * n, buf + 3 * n, buf + 4 * n, rc, n);
			else hipLaunchKernelGGL(move_k, dim3(c.grid), dim3(c.block), 0, st, buf, buf + n, buf + 2 * n, buf + 3 * n, buf + 4 * n, rc, n);
		};
		hipGraph_t g;
		hipGraphExec_t ge;
		CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
		for (int k = 0; k < NODES; k++) launch();
		CK(hipStreamEndCapture(st, &g));
		CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		for (int r = 0; r < 3; r++) CK(hipGraphLaunch(ge, st));
		CK(hipStreamSynchronize(st));
		float best = 1e30f, sum = 0;
		for (int r = 0; r < REPS; r++) {
			CK(hipEventRecord(e0, st));
			CK(hipGraphLaunch(ge, st));
			CK(hipEventRecord(e1, st));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			best = ms < best ? ms : best;
			sum += ms;
		}
		// direct launches, back to back
		for (int k = 0; k < 200; k++) launch();
		CK(hipStreamSynchronize(st));
		const int K = 2000;
		auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < K; k++) launch();
		CK(hipStreamSynchronize(st));
		auto t1 = std::chrono::steady_clock::now();
		std::printf("%-36s graph: %.3f us per node (best of %d replays of %d nodes; mean %.3f)   direct: %.3f us per launch (%d back to back)\n",
		            c.name, best * 1e3 / NODES, REPS, NODES, sum / REPS * 1e3 / NODES,
		            std::chrono::duration<double, std::micro>(t1 - t0).count() / K, K);
		CK(hipGraphExecDestroy(ge));
		CK(hipGraphDestroy(g));
	}
	return 0;
}
