// scratch: what v_permlane16_swap_b32 returns on gfx950 (qp_inv.hpp's 32-lane reductions)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *o, const double *in)
{
	double v = in[threadIdx.x];
	unsigned lo = __double2loint(v), hi = __double2hiint(v);
	auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
	auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
	o[threadIdx.x] = __hiloint2double(rh[0], rl[0]);
	o[64 + threadIdx.x] = __hiloint2double(rh[1], rl[1]);
}
int main()
{
	double h[64], r[128], *di, *dout;
	for (int i = 0; i < 64; i++) h[i] = i;
	hipMalloc(&di, sizeof(h));
	hipMalloc(&dout, sizeof(r));
	hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
	hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
	printf("first :");
	for (int i = 0; i < 64; i++) printf(" %g", r[i]);
	printf("\nsecond:");
	for (int i = 0; i < 64; i++) printf(" %g", r[64 + i]);
	printf("\n");
	return 0;
}
