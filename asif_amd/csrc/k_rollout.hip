// k_rollout.hip -- the caller's side of filter() for the two-stage filters (implicit, implicit-RB, TB): the plant's
// forward-Euler step between two filter calls, e.g. examples/InvertedPendulum_Implicit.cpp:119-136 and
// examples/segway_implicit_tb.cpp's main loop:
//     rc = asif->filter(xNow, uDesNow, uActNow, relax);   fCl = f(x) + g(x) uActNow;   xNow += dt * fCl
// asif_hip_rollout_batch strings T x (rows kernel, QP kernel, this kernel) on one stream: no host round trip
// between control steps, the state never leaves HBM.  (The explicit filter has the fully fused
// explicit_rollout_kernel; for these classes one control step is a 0.3-11 k-step backup trajectory, so fusing
// the launches would save < 0.1 % and cost the second stage's lane re-deal.)
#include "launchers.hpp"

namespace asif {

template <class M>
__global__ __launch_bounds__(256) void plant_step_kernel(DevOptions o, int64_t B, int64_t ld, double dt, double *x,
                                                         const double *uact, const int32_t *rc, int32_t *nfail,
                                                         double *xlog, double *ulog, int32_t *rclog)
{
	constexpr int NX = M::NX;
	static_assert(M::NU == 1, "single-input models");
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= B) return;
	double xs[NX], f[NX], g[NX];
#pragma unroll
	for (int k = 0; k < NX; k++) xs[k] = x[k * ld + i];
	const double u = uact[i];
	const int32_t r = rc[i];
	if (xlog) {
#pragma unroll
		for (int k = 0; k < NX; k++) xlog[k * ld + i] = xs[k];
	}
	if (ulog) ulog[i] = u;
	if (rclog) rclog[i] = r;
	if (r < 0) nfail[i] += 1;
	M::dynamics(o, xs, f, g);
#pragma unroll
	for (int k = 0; k < NX; k++) {
		double fcl = 0.0; // :121-131: fCl = 0; fCl += f; fCl += g*u
		fcl += f[k];
		fcl += g[k] * u;
		x[k * ld + i] = xs[k] + dt * fcl;
	}
}

template <class M>
static int launch_step(const DevOptions &o, int64_t B, int64_t ld, double dt, double *x, const double *uact,
                       const int32_t *rc, int32_t *nfail, double *xlog, double *ulog, int32_t *rclog, hipStream_t s)
{
	hipLaunchKernelGGL((plant_step_kernel<M>), dim3(grid_for(B, 1, 256)), dim3(256), 0, s, o, B, ld, dt, x, uact, rc,
	                   nfail, xlog, ulog, rclog);
	return (int)hipGetLastError();
}

int launch_plant_step(int model, const DevOptions &o, int64_t B, int64_t ld, double dt, double *x, const double *uact,
                      const int32_t *rc, int32_t *nfail, double *xlog, double *ulog, int32_t *rclog, hipStream_t s)
{
	switch (model) {
	case ASIF_HIP_MODEL_INVERTED_PENDULUM:
		return launch_step<InvertedPendulum>(o, B, ld, dt, x, uact, rc, nfail, xlog, ulog, rclog, s);
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_IMPLICIT:
		return launch_step<DoubleIntegratorImplicit>(o, B, ld, dt, x, uact, rc, nfail, xlog, ulog, rclog, s);
	case ASIF_HIP_MODEL_SEGWAY:
		return launch_step<Segway>(o, B, ld, dt, x, uact, rc, nfail, xlog, ulog, rclog, s);
	case ASIF_HIP_MODEL_INVERTED_PENDULUM_TB:
		return launch_step<InvertedPendulumTB>(o, B, ld, dt, x, uact, rc, nfail, xlog, ulog, rclog, s);
	case ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB:
		return launch_step<DoubleIntegratorTB>(o, B, ld, dt, x, uact, rc, nfail, xlog, ulog, rclog, s);
	default:
		return ASIF_HIP_EUNSUPPORTED;
	}
}

} // namespace asif
