// k_math_probe.hip -- self-test entry for the device math the trajectory kernels use in place of library calls
// (models.hpp: sincos_fast, tanh_abs_accurate, rcp_newton; backup_traj.hpp: sqrt_plain_range, div_plain_range, bevel_arc), so that
// their accuracy claims are tested on the device itself (tests/test_gpu_math_probe.py) and not only through the rows
// they feed.  HOST pointers in and out; n values per call.
#include <hip/hip_runtime.h>
#include "asif_hip.h"
#include "models.hpp"
#include "backup_traj.hpp"

namespace asif {

__global__ void math_probe_kernel(int kind, int64_t n, const double *a, const double *b, double *o0, double *o1)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double x = a[i], y = b ? b[i] : 0.0;
	double r0 = 0.0, r1 = 0.0;
	switch (kind) {
	case ASIF_HIP_PROBE_SINCOS: sincos_fast<kTrigUnchecked>(x, r0, r1); break; // the raw fast path, no range policing
	case ASIF_HIP_PROBE_SINCOS_CHECKED: sincos_fast<kTrigChecked>(x, r0, r1); break;
	case ASIF_HIP_PROBE_TANH: r0 = tanh_abs_accurate(x); break;
	case ASIF_HIP_PROBE_RCP: r0 = rcp_newton(x); break;
	case ASIF_HIP_PROBE_SQRT_PLAIN: r0 = BackupLoop<InvertedPendulum>::sqrt_plain_range(x); break;
	case ASIF_HIP_PROBE_DIV_PLAIN: r0 = BackupLoop<InvertedPendulum>::div_plain_range(x, y); break;
	case ASIF_HIP_PROBE_SINCOS_CARRY: { // one block of the trajectory loop: fresh evaluation at x, then 15 carried steps of y
		TrigCarry cy;
		sincos_fast<kTrigUnchecked>(x, cy.s, cy.c);
		cy.x = x;
		double xx = x;
		for (int k = 0; k < 15; k++) {
			xx += y;
			sincos_carry(xx, cy);
		}
		r0 = cy.s;
		r1 = cy.c;
		break;
	}
	case ASIF_HIP_PROBE_BEVEL_ARC: BackupLoop<InvertedPendulum>::bevel_arc(x, y, r0, r1); break;
	default: break;
	}
	o0[i] = r0;
	if (o1) o1[i] = r1;
}

} // namespace asif

extern "C" int asif_hip_math_probe(int device, int32_t kind, int64_t n, const double *a, const double *b, double *out0,
                                   double *out1)
{
	if (n < 0 || kind < 0 || kind > ASIF_HIP_PROBE_BEVEL_ARC || (n > 0 && (!a || !out0))) return ASIF_HIP_EINVAL;
	if ((kind == ASIF_HIP_PROBE_DIV_PLAIN || kind == ASIF_HIP_PROBE_SINCOS_CARRY || kind == ASIF_HIP_PROBE_BEVEL_ARC) &&
	    n > 0 && !b) return ASIF_HIP_EINVAL;
	if (n == 0) return ASIF_HIP_OK;
	hipError_t e = hipSetDevice(device);
	if (e != hipSuccess) return ASIF_HIP_ENODEVICE;
	double *d = nullptr;
	const size_t sz = sizeof(double) * (size_t)n;
	if ((e = hipMalloc((void **)&d, 4 * sz)) != hipSuccess) return (int)e;
	e = hipMemcpy(d, a, sz, hipMemcpyHostToDevice);
	if (e == hipSuccess && b) e = hipMemcpy(d + n, b, sz, hipMemcpyHostToDevice);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(asif::math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, (int)kind, n,
		                   (const double *)d, b ? (const double *)(d + n) : (const double *)nullptr, d + 2 * n,
		                   out1 ? d + 3 * n : (double *)nullptr);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpy(out0, d + 2 * n, sz, hipMemcpyDeviceToHost);
	if (e == hipSuccess && out1) e = hipMemcpy(out1, d + 3 * n, sz, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	return (int)e;
}
