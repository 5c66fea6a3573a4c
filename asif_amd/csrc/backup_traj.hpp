// backup_traj.hpp -- the backup closed loop and its sensitivity, shared by the implicit and TB filters.
//
// Restates, for one instance per lane and nu == 1:
//   inputSaturateSoft   src/asif_implicit.cpp:682-737  (== src/asif_implicit_tb.cpp:764-819)
//   backupCLdynamics    src/asif_implicit.cpp:751-815  (separate dynamics/dynamicsGradients branch)
//   ODErhs              src/asif_implicit.cpp:817-827  z = [x; vec Q], zdot = [fCL; DfCL Q]
// plus a register-resident running selection of the K smallest safety margins along the trajectory,
// which replaces the reference's full trajectory storage (5001 x 18 doubles on the stack,
// src/asif_implicit.cpp:441-443) and its std::sort of every sample (:487).
#pragma once
#include <hip/hip_runtime.h>
#include "models.hpp"

namespace asif {

template <class M>
struct BackupLoop {
	static constexpr int NX = M::NX, NZ = NX + NX * NX;
	static_assert(M::NU == 1, "backup loop is written for single-input models (every shipped example)");

	// bevelled smooth saturation of the backup input; DuSat is d uSat / d u in the reference's own
	// (normalised-arc) convention -- reproduced literally
	__device__ __forceinline__ static void saturateSoft(const DevOptions &o, double u, double &uSat, double &DuSat)
	{
		const double r = o.satSharpness;
		const double mi = o.lb[0], ma = o.ub[0];
		const double range = o.satRange;
		const double middle = o.satMiddle;
		const double uc = (u - middle) * o.twoOverRange;
		const double xc = o.bevelStop;
		const double yc = 1 - r;
		// four regions: linear, clamped high / low (selects), and the two bevels (sqrt + divide).  The bevels are
		// a narrow band of u; they sit behind ONE wave-level branch so that the common step pays a compare and a
		// not-taken scalar branch instead of four nested exec-mask sequences.
		const bool hi = uc >= o.bevelStop, lo = uc <= -o.bevelStop;
		uSat = hi ? ma : (lo ? mi : u);
		DuSat = (hi || lo) ? 0.0 : 1.0;
		const bool bevelUp = !hi && uc > o.bevelStart, bevelDn = !lo && uc < -o.bevelStart;
		if (__any(bevelUp || bevelDn)) {
			if (bevelUp) {
				const double s = sqrt(r * r - (uc - xc) * (uc - xc));
				DuSat = (xc - uc) / s;
				uSat = 0.5 * (s + yc) * range + middle;
			} else if (bevelDn) {
				const double s = sqrt(r * r - (uc + xc) * (uc + xc));
				DuSat = (xc + uc) / s;
				uSat = 0.5 * (-s - yc) * range + middle;
			}
		}
	}

	// Zero-order hold of the backup input, ASIFimplicitRB only (src/asif_implicit_robust.cpp:891-903, members
	// t_last_zoh_ / u_zoh_): re-sampled when t >= t_last + backContDt - 1e-4; the first rhs of a trajectory
	// (t <= backTrajDt) resets the clock.  Du_zoh_ is read only by the fused-gradient branch (:913-919); the
	// separate dynamics / dynamicsGradients branch every device model takes keeps the FRESH Du (:936-938).
	struct Hold {
		double u, tLast;
	};

	template <bool HOLD>
	__device__ __forceinline__ static void closedLoopT(const DevOptions &o, const double (&x)[NX], double (&fCL)[NX],
	                                                   double (&DfCL)[NX * NX], Hold &hold, double t)
	{
		double f[NX], g[NX], Df[NX * NX], Dg[NX * NX], u[1], Du[NX], uSat, DuSat;
		M::backupController(o, x, u, Du);
		double us = u[0];
		if (HOLD && o.backContDt > 0) { // backContDt == 0: plain ASIFimplicit routed through the RB kernel (learning)
			if (t <= o.trajDt) hold.tLast = -1.;
			if (t >= (hold.tLast + o.backContDt - 0.0001)) {
				hold.u = u[0];
				hold.tLast = t;
			}
			us = hold.u;
		}
		saturateSoft(o, us, uSat, DuSat);
		M::dynamicsAndGradients(o, x, f, g, Df, Dg);
		if constexpr (M::kInputOnLastState) {
			// g = e_last, Dg = 0: the general expression below with its constant factors folded by hand
			// (the compiler may not fold x*0 or 0+x for doubles)
#pragma unroll
			for (int i = 0; i < NX; i++) {
#pragma unroll
				for (int j = 0; j < NX; j++)
					DfCL[i + j * NX] = (i == NX - 1) ? Df[i + j * NX] + DuSat * Du[j] : Df[i + j * NX];
				fCL[i] = (i == NX - 1) ? uSat + f[i] : f[i];
			}
		} else {
#pragma unroll
			for (int i = 0; i < NX; i++) {
#pragma unroll
				for (int j = 0; j < NX; j++)
					DfCL[i + j * NX] = Df[i + j * NX] + (Dg[i + j * NX] * uSat + g[i] * DuSat * Du[j]);
				fCL[i] = g[i] * uSat + f[i];
			}
		}
	}

	__device__ __forceinline__ static void closedLoop(const DevOptions &o, const double (&x)[NX], double (&fCL)[NX],
	                                                  double (&DfCL)[NX * NX])
	{
		Hold none = {0.0, 0.0};
		closedLoopT<false>(o, x, fCL, DfCL, none, 0.0);
	}

	__device__ __forceinline__ static void eulerStep(const DevOptions &o, double (&z)[NZ])
	{
		Hold none = {0.0, 0.0};
		eulerStepT<false>(o, z, none, 0.0);
	}

	// one forward-Euler step of [x; vec Q] (src/asif_implicit.cpp:470-477: rhs*dt + previous); t is the time
	// the reference stamps on this rhs (src/asif_implicit_robust.cpp:567: i*backTrajDt for the step INTO sample i)
	template <bool HOLD>
	__device__ __forceinline__ static void eulerStepT(const DevOptions &o, double (&z)[NZ], Hold &hold, double t)
	{
		double x[NX], fCL[NX], DfCL[NX * NX], zd[NZ];
#pragma unroll
		for (int i = 0; i < NX; i++) x[i] = z[i];
		closedLoopT<HOLD>(o, x, fCL, DfCL, hold, t);
#pragma unroll
		for (int i = 0; i < NX; i++) zd[i] = fCL[i];
#pragma unroll
		for (int i = 0; i < NX; i++)
#pragma unroll
			for (int j = 0; j < NX; j++) {
				double s = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) s += DfCL[i + k * NX] * z[NX + k + j * NX];
				zd[NX + i + j * NX] = s;
			}
#pragma unroll
		for (int k = 0; k < NZ; k++) z[k] = zd[k] * o.trajDt + z[k];
	}
};

// The K smallest keys seen so far, ascending, ties -> earlier sample first (this build's fixed
// tie rule; the reference's std::sort leaves it implementation-defined).
// Each entry owns a payload slot in LDS; insert() returns the slot the caller must overwrite, or -1.
template <int K>
struct TopK {
	double key[K];
	int idx[K], slot[K];
	__device__ __forceinline__ void init()
	{
#pragma unroll
		for (int p = 0; p < K; p++) {
			key[p] = __builtin_huge_val();
			idx[p] = -1;
			slot[p] = p;
		}
	}
	__device__ __forceinline__ int insert(double kv, int iv)
	{
		if (!(kv < key[K - 1])) return -1;
		const int s = slot[K - 1];
		int pos = 0;
#pragma unroll
		for (int p = 0; p < K - 1; p++) pos += (key[p] <= kv) ? 1 : 0;
#pragma unroll
		for (int p = K - 1; p >= 1; p--) {
			const bool shift = p > pos, here = p == pos;
			key[p] = here ? kv : (shift ? key[p - 1] : key[p]);
			idx[p] = here ? iv : (shift ? idx[p - 1] : idx[p]);
			slot[p] = here ? s : (shift ? slot[p - 1] : slot[p]);
		}
		if (pos == 0) {
			key[0] = kv;
			idx[0] = iv;
			slot[0] = s;
		}
		return s;
	}
};

} // namespace asif
