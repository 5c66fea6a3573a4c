// models.hpp -- device models: the user callbacks of the reference's example programs as
// compile-time device functors (host std::function callbacks cannot run in a kernel).
// Layouts follow the reference: Dh column-major npSS x nx (Dh[i + j*npSS]), g column-major nx x nu,
// Df column-major nx x nx, Dg index i + k*nx + j*nx*nu, Du column-major nu x nx.
// The numeric constants are the workload definition and are restated from the cited example files.
#pragma once
#include <hip/hip_runtime.h>
#include "asif_hip.h"

namespace asif {

constexpr double kPi = 3.14159265358979323846;

// Kernel-side copy of the options (plain data, passed by value as a kernel argument -> SGPRs).
struct DevOptions {
	double relaxCost, relaxLb, relaxReachLb, relaxTTS, relaxMinOrtho;
	double backTrajHorizon, backTrajDt, backTrajMinOrtho, satSharpness, inf;
	double lb[ASIF_HIP_MAX_NU], ub[ASIF_HIP_MAX_NU];
	double pMin, pMax;
	double halfPlanes[2 * ASIF_HIP_MAX_HALFPLANES];
	int nHalfPlanes;
	int npBT;          // backup-trajectory samples (host-computed, src/asif_implicit.cpp:211-216)
	double trajDt;     // effective Euler step (may differ from backTrajDt when npBT was clamped)
	// soft-saturation bevel constants, evaluated ONCE on the host with libm exactly as the reference
	// evaluates them per call (src/asif_implicit.cpp:689-703): r*tan(pi/8), 1-cos(pi/4)*bevelL, 1+bevelL
	double bevelL, bevelStart, bevelStop;
};

// ---------------------------------------------------------------------------------------------
// Double integrator, examples/DoubleIntegrator.cpp:12-61.  x = (position, velocity).
struct DoubleIntegrator {
	static constexpr int NX = 2, NU = 1, NPSS = 4;

	// :24-38  safe box |x|<=1, |v|<=1 with the braking parabola on the side the velocity points to
	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
		const double xlo = -1.0, xhi = 1.0, vlo = -1.0, vhi = 1.0;
		const double brake = (x[1] * x[1]) / 2.0;
		const bool fwd = x[1] > 0;
		h[0] = fwd ? (xhi - x[0] - brake) : (-x[0] + xhi);
		h[1] = fwd ? (x[0] - xlo) : (x[0] - xlo - brake);
		h[2] = x[1] - vlo;
		h[3] = -x[1] + vhi;
		Dh[0] = -1.0; Dh[4] = fwd ? -x[1] : 0.0;
		Dh[1] = 1.0;  Dh[5] = fwd ? 0.0 : -x[1];
		Dh[2] = 0.0;  Dh[6] = 1.0;
		Dh[3] = 0.0;  Dh[7] = -1.0;
	}
	// :40-61  f = [0 1; 0 0] x, g = (0, 1)
	__device__ static void dynamics(const DevOptions &, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		f[0] = x[1];
		f[1] = 0.0;
		g[0] = 0.0;
		g[1] = 1.0;
	}
};

} // namespace asif
