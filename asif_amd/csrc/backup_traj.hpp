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

template <class M, class = void>
struct unit_row_mask : std::integral_constant<unsigned, 0u> {};
template <class M>
struct unit_row_mask<M, std::void_t<decltype(M::kDfUnitRowMask)>> : std::integral_constant<unsigned, M::kDfUnitRowMask> {};
template <class M, class = void>
struct dg_mask : std::integral_constant<unsigned, 0xffffffffu> {};
template <class M>
struct dg_mask<M, std::void_t<decltype(M::kDgMask)>> : std::integral_constant<unsigned, M::kDgMask> {};

template <class M>
struct BackupLoop {
	static constexpr int NX = M::NX, NZ = NX + NX * NX;
	// Structure the models declare (models.hpp) so that the loop can drop products with literal zeros and ones -- the
	// compiler may not (0 * x is not 0 for every x); for finite operands the values are the same:
	//   row i of Df is the unit row e_{i+1}' with g_i = 0 and row i of Dg zero  ->  row i of DfCL Q is row i+1 of Q
	//   and fCL_i = f_i (kDfUnitRowMask; kDfFirstRowShift is the two-state models' older name for row 0);
	//   entries of Dg that can be non-zero (kDgMask; default: all).
	static constexpr unsigned unitRowMask() { return unit_row_mask<M>::value | ((M::kDfFirstRowShift && NX == 2) ? 1u : 0u); }
	static constexpr bool unitRow(int i) { return ((unitRowMask() >> i) & 1u) != 0; }
	static constexpr bool dgEntry(int e) { return ((dg_mask<M>::value >> e) & 1u) != 0; }
	static_assert(M::NU == 1, "backup loop is written for single-input models (every shipped example)");

	// sqrt and divide as the compiler expands them (v_rsq / v_rcp, Newton steps in FMA, v_div_fixup), minus the
	// rescaling of operands near the ends of the exponent range (v_div_scale, the ldexp pair around the sqrt).  Bit for
	// bit the IEEE results whenever the rescaling would not have triggered: x = 0 or x >= 2^-767 for the sqrt; for a / b
	// both magnitudes (a may be 0) within 2^+-255.  (The trajectory loop's bevel now takes the joint sequence bevel_arc below;
	// these two stay as the sequences it is compared with on the device, k_math_probe.hip.)
	__device__ __forceinline__ static double sqrt_plain_range(double x)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const double y = __builtin_amdgcn_rsq(x);
		double g = x * y, h = 0.5 * y;
		const double r = fma(-h, g, 0.5);
		g = fma(g, r, g);
		h = fma(h, r, h);
		double d = fma(-g, g, x);
		g = fma(d, h, g);
		d = fma(-g, g, x);
		g = fma(d, h, g);
		return (x == 0.0 || x == __builtin_huge_val()) ? x : g;
#else
		return sqrt(x);
#endif
	}
	__device__ __forceinline__ static double div_plain_range(double a, double b)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		double r = __builtin_amdgcn_rcp(b);
		double e = fma(-b, r, 1.0);
		r = fma(r, e, r);
		e = fma(-b, r, 1.0);
		r = fma(r, e, r);
		double q = a * r;
		const double t = fma(-b, q, a);
		q = fma(t, r, q);
		return __builtin_amdgcn_div_fixup(q, b, a);
#else
		return a / b;
#endif
	}
	// The bevel's pair sq = sqrt(d), q = n / sq in one go, for d in [r^2/2, r^2] and |n| <= r (the arc of the soft
	// saturation; r within satFastOk's range).  The square root's own refinement leaves h = 1/(2 sqrt(d)) to ~2^-50: 2h
	// after ONE Newton step is the reciprocal the division's sequence would have built from v_rcp with two -- a
	// transcendental and a Newton step less; d is never 0 or inf and the quotient never special here, so the sqrt's
	// select and v_div_fixup go too.  Same bits as sqrt_plain_range / div_plain_range (checked on the device over the
	// arc's whole range, tests/test_gpu_math_probe.py).
	__device__ __forceinline__ static void bevel_arc(double d, double n, double &sq, double &q)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const double y = __builtin_amdgcn_rsq(d);
		double g = d * y, h = 0.5 * y;
		const double r0 = fma(-h, g, 0.5);
		g = fma(g, r0, g);
		h = fma(h, r0, h);
		double e = fma(-g, g, d);
		g = fma(e, h, g);
		e = fma(-g, g, d);
		g = fma(e, h, g);
		sq = g;
		double r = h + h;
		const double e1 = fma(-g, r, 1.0);
		r = fma(r, e1, r);
		double qq = n * r;
		const double t = fma(-g, qq, n);
		q = fma(t, r, qq);
#else
		sq = sqrt(d);
		q = n / sq;
#endif
	}
	// bevelled smooth saturation of the backup input; DuSat is d uSat / d u in the reference's own
	// (normalised-arc) convention -- reproduced literally.
	// Four regions: linear, clamped high / low (selects), and the two bevels (sqrt + divide) as ONE evaluation on |uc|,
	// bit for bit the reference's two expressions:
	//   up:   s = sqrt(r^2 - (uc - xc)^2),  DuSat = (xc - uc)/s,  uSat = 0.5 ( s + yc) range + middle
	//   down: s = sqrt(r^2 - (uc + xc)^2),  DuSat = (xc + uc)/s,  uSat = 0.5 (-s - yc) range + middle
	// ((uc + xc)^2 = (|uc| - xc)^2 and 0.5 (-s - yc) range = -(0.5 (s + yc) range) exactly.)  Some lane of a wave sits in
	// a bevel on ~40 % of the Euler steps: one divergent block (exec mask + a branch the wave takes when no lane is in
	// a bevel).  Two divergent arcs cost 1.40 ms on C3, this form 1.33, and running the block on every step for every
	// lane 1.47 against 1.20 at the time of that measurement.
	// FAST (only where the host has set DevOptions::satFastOk: 0 < bevelStart < bevelStop, sharpness and bevelStop
	// within 2^+-100) decides the regions on |uc| and the sign of uc -- three compares instead of six; with ordered
	// positive thresholds the same regions, NaN in none of them either way -- and takes sqrt and divide without the
	// rescaling steps: nonzero d >= ulp(r^2)/2, sq in [r 2^-27, r], |xc - au| in {0} U [ulp(xc)/2, bevelL], far from the
	// thresholds where the IEEE sequences rescale.  Same bits, ~10 VALU issues fewer per step.
	// NOBEVEL (with FAST): the step of a block that is EXPECTED to stay clear of both bevels (bevel_rate below): no
	// divergent block and no branch at all -- the linear and the clamped regions only -- and *seen latches a lane that
	// was in a bevel after all, for the caller to repeat the block with the full step.
	template <bool FAST = false, bool NOBEVEL = false>
	__device__ __forceinline__ static void saturateSoft(const DevOptions &o, double u, double &uSat, double &DuSat,
	                                                    bool *seen = nullptr)
	{
#pragma clang fp contract(on) // fused where written as one, nowhere else: the same bits in every kernel this is inlined into
		const double r = o.satSharpness;
		const double mi = o.lb[0], ma = o.ub[0]; // in VGPRs already: see the rows kernels' prologue
		const double range = o.satRange;
		const double middle = o.satMiddle;
		const double uc = (u - middle) * o.twoOverRange;
		const double xc = o.bevelStop;
		const double yc = 1 - r;
		if constexpr (FAST) {
			const double au = fabs(uc);
			const bool clamped = au >= o.bevelStop;
			// linear region: u lies strictly inside [mi, ma] (bevelStart < 1); clamped: beyond them (bevelStop > 1): a
			// clamp gives both in two instructions, the bevel in between is overwritten below.  (The clamp drops a NaN
			// input where the select kept it: u is NaN only next to a NaN or overflowing state, which the callers'
			// alarm -- kStateSane -- sends through the checking step.)
			uSat = min_num(max_num(u, mi), ma);
			DuSat = clamped ? 0.0 : 1.0;
			if constexpr (NOBEVEL) {
				*seen = *seen || (au > o.bevelStart && !clamped);
			} else if (au > o.bevelStart && !clamped) { // divergent: skipped by the wave when no lane is in a bevel
				const bool neg = uc < 0.0;
				const double t = au - xc;
				const double d = fma(-t, t, r * r);
				double sq;
				bevel_arc(d, xc - au, sq, DuSat);
				const double us = 0.5 * (sq + yc) * range;
				uSat = neg ? middle - us : us + middle;
			}
		} else { // the reference's chain, region by region and in its order (src/asif_implicit.cpp:704-735)
			if (uc >= o.bevelStop) {
				uSat = ma;
				DuSat = 0.0;
			} else if (uc <= -o.bevelStop) {
				uSat = mi;
				DuSat = 0.0;
			} else if (uc <= o.bevelStart && uc >= -o.bevelStart) {
				uSat = u;
				DuSat = 1.0;
			} else if (uc > o.bevelStart) {
				const double t = uc - xc;
				const double sq = sqrt(fma(-t, t, r * r));
				DuSat = (xc - uc) / sq;
				const double us = 0.5 * (sq + yc) * range;
				uSat = us + middle;
			} else if (uc < -o.bevelStart) {
				const double t = uc + xc;
				const double sq = sqrt(fma(-t, t, r * r));
				DuSat = (xc + uc) / sq;
				const double us = 0.5 * (-sq - yc) * range;
				uSat = us + middle;
			} else { // NaN
				uSat = u;
				DuSat = 1.0;
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

	template <bool HOLD, int POISON = kTrigChecked, bool NOBEVEL = false>
	__device__ __forceinline__ static void closedLoopT(const DevOptions &o, const double (&x)[NX], double (&fCL)[NX],
	                                                   double (&DfCL)[NX * NX], Hold &hold, double t,
	                                                   TrigCarry *cy = nullptr, bool reset = true, bool *seen = nullptr)
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
		saturateSoft<POISON != kTrigChecked, NOBEVEL>(o, us, uSat, DuSat, seen);
		if constexpr (POISON == kTrigCarried) M::dynamicsAndGradientsCarried(o, x, f, g, Df, Dg, *cy, reset);
		else M::template dynamicsAndGradients<POISON>(o, x, f, g, Df, Dg);
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
				if (unitRow(i)) { // row i of DfCL is not formed: the products below copy row i+1 of Q instead
#pragma unroll
					for (int j = 0; j < NX; j++) DfCL[i + j * NX] = Df[i + j * NX];
					fCL[i] = f[i];
					continue;
				}
#pragma unroll
				for (int j = 0; j < NX; j++) {
					const double gd = g[i] * DuSat * Du[j];
					DfCL[i + j * NX] = Df[i + j * NX] + (dgEntry(i + j * NX) ? Dg[i + j * NX] * uSat + gd : gd);
				}
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

	// ODErhs (src/asif_implicit.cpp:817-827): zdot = [fCL(x); DfCL(x) Q], time-invariant (no held input)
	__device__ __forceinline__ static void rhs(const DevOptions &o, const double (&z)[NZ], double (&zd)[NZ])
	{
		double x[NX], fCL[NX], DfCL[NX * NX];
#pragma unroll
		for (int i = 0; i < NX; i++) x[i] = z[i];
		closedLoop(o, x, fCL, DfCL);
#pragma unroll
		for (int i = 0; i < NX; i++) zd[i] = fCL[i];
#pragma unroll
		for (int i = 0; i < NX; i++)
#pragma unroll
			for (int j = 0; j < NX; j++) {
				if (unitRow(i)) { // 0 * Q(0,j) + ... + 1 * Q(i+1,j) + ...: the same value without the FMAs
					zd[NX + i + j * NX] = z[NX + i + 1 + j * NX];
					continue;
				}
				double s = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) s += DfCL[i + k * NX] * z[NX + k + j * NX];
				zd[NX + i + j * NX] = s;
			}
	}

	__device__ __forceinline__ static void eulerStep(const DevOptions &o, double (&z)[NZ])
	{
		Hold none = {0.0, 0.0};
		eulerStepT<false>(o, z, none, 0.0);
	}

	// ---- the Euler step in two roles (k_tb.hip: tb_rows_split_kernel).  x does not depend on Q, so the step of
	// [x; vec Q] splits into the step of x -- controller, soft saturation, f, g -- and the step of Q, which needs
	// the gradients at the same x, the same saturation values and the same sin / cos.  Expressions as in closedLoopT /
	// eulerStepT above, term for term, so that the two roles together give what the one step gives.
	struct StepRecord { // what the x role hands the Q role for one step: what the gradients at the step's state need
		double xg[2];      // the two states they depend on directly (M::kGradStates)
		double uSat, DuSat; // the saturated input and its slope there
		double s, c;        // sin / cos of the model's angle there
		double iden, rg;    // the two reciprocals of the model's shared terms there
	};
	static constexpr int kRecordDoubles = 8;
	// fast step of x alone (trig carried along the steps, soft saturation's short forms); returns the record of this step
	// k: the model's constants as the caller's loop holds them (M::constants<PIN>() made before the loop: the groups of
	// PIN in vector registers; values unchanged).  CARRYV: the same choice for the seven constants of sincos_carry.
	template <bool CARRYV = false, class KC>
	__device__ __forceinline__ static StepRecord stepX(const DevOptions &o, double (&x)[NX], TrigCarry &cy, const KC &k)
	{
#pragma clang fp contract(on)
		static_assert(!M::kInputOnLastState, "role split is written for the general closed loop (the segway)");
		StepRecord r;
		double f[NX], g[NX], u[1], Du[NX];
		r.xg[0] = x[M::kGradStates[0]];
		r.xg[1] = x[M::kGradStates[1]];
		M::backupController(o, x, u, Du, k);
		saturateSoft<true>(o, u[0], r.uSat, r.DuSat);
		// (the segway's fused kernels have no vector register to spare for the seven constants; its x role has)
		sincos_carry<(M::NX <= 2) || CARRYV>(x[M::kTrigAngle], cy);
		r.s = cy.s;
		r.c = cy.c;
		typename M::Trig t;
		t.s1 = cy.s;
		t.c1 = cy.c;
		t.s2 = 2.0 * t.s1 * t.c1;
		t.c2 = (t.c1 - t.s1) * (t.c1 + t.s1);
		const typename M::Shared h = M::dynamicsT(x, t, f, g, k);
		r.iden = h.iden;
		r.rg = h.rg;
#pragma unroll
		for (int i = 0; i < NX; i++) {
			const double fcl = unitRow(i) ? f[i] : g[i] * r.uSat + f[i];
			x[i] = fcl * o.trajDt + x[i];
		}
		return r;
	}
	// the same step's update of Q (vec Q in q, column-major) from the record
	template <bool TANHV = false, class KC>
	__device__ __forceinline__ static void stepQ(const DevOptions &o, const StepRecord &r, double (&q)[NX * NX], const KC &k)
	{
#pragma clang fp contract(on)
		double xr[NX], g[NX], Df[NX * NX], Dg[NX * NX], u[1], Du[NX], DfCL[NX * NX], zd[NX * NX];
#pragma unroll
		for (int i = 0; i < NX; i++) xr[i] = 0.0;
		xr[M::kGradStates[0]] = r.xg[0];
		xr[M::kGradStates[1]] = r.xg[1];
		M::backupController(o, xr, u, Du, k); // Du is the constant gain; u is not used here
		typename M::Trig t;
		t.s1 = r.s;
		t.c1 = r.c;
		t.s2 = 2.0 * t.s1 * t.c1;
		t.c2 = (t.c1 - t.s1) * (t.c1 + t.s1);
		typename M::Shared h;
		h.iden = r.iden;
		h.rg = r.rg;
		M::gainT(t, h, g, k);
		M::template gradientsGiven<TANHV>(xr, t, h, Df, Dg, k);
#pragma unroll
		for (int i = 0; i < NX; i++) {
			if (unitRow(i)) continue; // row i of DfCL Q is row i+1 of Q
#pragma unroll
			for (int j = 0; j < NX; j++) {
				const double gd = g[i] * r.DuSat * Du[j];
				DfCL[i + j * NX] = Df[i + j * NX] + (dgEntry(i + j * NX) ? Dg[i + j * NX] * r.uSat + gd : gd);
			}
		}
#pragma unroll
		for (int i = 0; i < NX; i++)
#pragma unroll
			for (int j = 0; j < NX; j++) {
				if (unitRow(i)) {
					zd[i + j * NX] = q[i + 1 + j * NX];
					continue;
				}
				double sacc = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) sacc += DfCL[i + k * NX] * q[k + j * NX];
				zd[i + j * NX] = sacc;
			}
#pragma unroll
		for (int k = 0; k < NX * NX; k++) q[k] = zd[k] * o.trajDt + q[k];
	}

	// the fused fast step of a model that has the two roles: the roles themselves, back to back in one wave (the
	// compiler merges what they compute twice), so that a trajectory is the same bits whichever kernel integrates it
	__device__ __forceinline__ static void eulerStepRoles(const DevOptions &o, double (&z)[NZ], TrigCarry &cy)
	{
		double x[NX], q[NX * NX];
#pragma unroll
		for (int i = 0; i < NX; i++) x[i] = z[i];
#pragma unroll
		for (int i = 0; i < NX * NX; i++) q[i] = z[NX + i];
		const typename M::Consts k = M::template constants<0>(); // plain literals
		const StepRecord r = stepX(o, x, cy, k);
		stepQ(o, r, q, k);
#pragma unroll
		for (int i = 0; i < NX; i++) z[i] = x[i];
#pragma unroll
		for (int i = 0; i < NX * NX; i++) z[NX + i] = q[i];
	}

	// one forward-Euler step of [x; vec Q] (src/asif_implicit.cpp:470-477: rhs*dt + previous); t is the time
	// the reference stamps on this rhs (src/asif_implicit_robust.cpp:567: i*backTrajDt for the step INTO sample i)
	// POISON: the model's sin / cos never branch; out-of-range arguments turn the state into NaN (see sincos_fast)
	template <bool HOLD, int POISON = kTrigChecked, bool NOBEVEL = false>
	__device__ __forceinline__ static void eulerStepT(const DevOptions &o, double (&z)[NZ], Hold &hold, double t,
	                                                  TrigCarry *cy = nullptr, bool reset = true, bool *seen = nullptr)
	{
		static_assert(!NOBEVEL || POISON != kTrigChecked, "the bevel-free step is a form of the fast step");
		double x[NX], fCL[NX], DfCL[NX * NX], zd[NZ];
#pragma unroll
		for (int i = 0; i < NX; i++) x[i] = z[i];
		closedLoopT<HOLD, POISON, NOBEVEL>(o, x, fCL, DfCL, hold, t, cy, reset, seen);
#pragma unroll
		for (int i = 0; i < NX; i++) zd[i] = fCL[i];
#pragma unroll
		for (int i = 0; i < NX; i++)
#pragma unroll
			for (int j = 0; j < NX; j++) {
				if (unitRow(i)) { // 0 * Q(0,j) + ... + 1 * Q(i+1,j) + ...: the same value without the FMAs
					zd[NX + i + j * NX] = z[NX + i + 1 + j * NX];
					continue;
				}
				double s = 0.0;
#pragma unroll
				for (int k = 0; k < NX; k++) s += DfCL[i + k * NX] * z[NX + k + j * NX];
				zd[NX + i + j * NX] = s;
			}
#pragma unroll
		for (int k = 0; k < NZ; k++) z[k] = zd[k] * o.trajDt + z[k];
	}
};

// The reference's USE_ODEINT build of the backup trajectory (src/asif_implicit.cpp:427-460):
//   make_dense_output(backTrajAbsTol, backTrajRelTol, runge_kutta_dopri5<state_t>()) observed every backTrajDt.
// One instance per lane, every lane with its own step size.  Boost.odeint is not in the reference tree nor in the
// image: the Dormand-Prince tableau and its continuous extension are the published ones, the step controller is
// odeint's restated from memory (see oracle/or_assembly.c, dopri5_step: same formulas, same order) -- unpinned.
template <class M>
struct Dopri5 {
	static constexpr int NZ = BackupLoop<M>::NZ;
	double t, tOld, dt;
	double z[NZ], zOld[NZ];
	double k1[NZ], k3[NZ], k4[NZ], k5[NZ], k6[NZ], k7[NZ]; // of the last accepted step; k7 = f(z) (first same as last)
	// odeint throws after 500 failed attempts of one step, and a NaN error estimate (dropped by the max) never recovers;
	// the reference has no handler.  Here such a lane stops stepping and every later sample of it is NaN: non-finite
	// rows, filter() fails like a failed solve (oracle/or_assembly.c: dopri5_t::failed, same rule).
	bool failed;
	int rejects;

	__device__ __forceinline__ void init(const DevOptions &o, const double (&z0)[NZ], double dt0)
	{
		t = 0.0;
		tOld = 0.0;
		dt = dt0;
		failed = false;
		rejects = 0;
#pragma unroll
		for (int i = 0; i < NZ; i++) {
			z[i] = z0[i];
			zOld[i] = z0[i];
			k1[i] = k3[i] = k4[i] = k5[i] = k6[i] = 0.0;
		}
		BackupLoop<M>::rhs(o, z, k7);
	}

	// one attempt of a step for the lanes with `need`; accepted -> state advances, rejected -> dt shrinks
	__device__ __forceinline__ void tryStep(const DevOptions &o, bool need)
	{
		constexpr double a21 = 1.0 / 5, a31 = 3.0 / 40, a32 = 9.0 / 40, a41 = 44.0 / 45, a42 = -56.0 / 15, a43 = 32.0 / 9,
		                 a51 = 19372.0 / 6561, a52 = -25360.0 / 2187, a53 = 64448.0 / 6561, a54 = -212.0 / 729,
		                 a61 = 9017.0 / 3168, a62 = -355.0 / 33, a63 = 46732.0 / 5247, a64 = 49.0 / 176,
		                 a65 = -5103.0 / 18656, c1 = 35.0 / 384, c3 = 500.0 / 1113, c4 = 125.0 / 192,
		                 c5 = -2187.0 / 6784, c6 = 11.0 / 84;
		constexpr double dc1 = 35.0 / 384 - 5179.0 / 57600, dc3 = 500.0 / 1113 - 7571.0 / 16695,
		                 dc4 = 125.0 / 192 - 393.0 / 640, dc5 = -2187.0 / 6784 - (-92097.0 / 339200),
		                 dc6 = 11.0 / 84 - 187.0 / 2100, dc7 = -1.0 / 40;
		const double h = dt;
		double zt[NZ], n2[NZ], n3[NZ], n4[NZ], n5[NZ], n6[NZ], n7[NZ], zn[NZ];
		{
#pragma clang fp contract(off) // the oracle's gcc build keeps every product and sum separately rounded
#pragma unroll
			for (int i = 0; i < NZ; i++) zt[i] = z[i] + h * a21 * k7[i];
			BackupLoop<M>::rhs(o, zt, n2);
#pragma unroll
			for (int i = 0; i < NZ; i++) zt[i] = z[i] + h * (a31 * k7[i] + a32 * n2[i]);
			BackupLoop<M>::rhs(o, zt, n3);
#pragma unroll
			for (int i = 0; i < NZ; i++) zt[i] = z[i] + h * (a41 * k7[i] + a42 * n2[i] + a43 * n3[i]);
			BackupLoop<M>::rhs(o, zt, n4);
#pragma unroll
			for (int i = 0; i < NZ; i++) zt[i] = z[i] + h * (a51 * k7[i] + a52 * n2[i] + a53 * n3[i] + a54 * n4[i]);
			BackupLoop<M>::rhs(o, zt, n5);
#pragma unroll
			for (int i = 0; i < NZ; i++)
				zt[i] = z[i] + h * (a61 * k7[i] + a62 * n2[i] + a63 * n3[i] + a64 * n4[i] + a65 * n5[i]);
			BackupLoop<M>::rhs(o, zt, n6);
#pragma unroll
			for (int i = 0; i < NZ; i++) zn[i] = z[i] + h * (c1 * k7[i] + c3 * n3[i] + c4 * n4[i] + c5 * n5[i] + c6 * n6[i]);
			BackupLoop<M>::rhs(o, zn, n7);
		}
		double err = 0.0;
		bool nan = false;
		{
#pragma clang fp contract(off)
#pragma unroll
			for (int i = 0; i < NZ; i++) {
				const double xe = h * (dc1 * k7[i] + dc3 * n3[i] + dc4 * n4[i] + dc5 * n5[i] + dc6 * n6[i] + dc7 * n7[i]);
				const double e = fabs(xe) / (o.trajAbsTol + o.trajRelTol * (fabs(z[i]) + fabs(h) * fabs(k7[i])));
				err = fmax(err, e);
				nan = nan | (e != e);
			}
		}
		need = need & !failed;
		const bool reject = need && !nan && err > 1.0;
		const bool accept = need && !nan && !(err > 1.0);
		rejects = reject ? rejects + 1 : (accept ? 0 : rejects);
		failed = failed | (need & nan) | (rejects >= 500);
		const double shrink = fmax(0.9 * pow(err, -1.0 / 3.0), 0.2);
		const double grow = 0.9 * pow(fmax(err, 1.0 / 3125.0), -1.0 / 5.0);
		dt = reject ? h * shrink : ((accept && err < 0.5) ? h * grow : dt);
		tOld = accept ? t : tOld;
		t = accept ? t + h : t;
#pragma unroll
		for (int i = 0; i < NZ; i++) {
			zOld[i] = accept ? z[i] : zOld[i];
			k1[i] = accept ? k7[i] : k1[i];
			z[i] = accept ? zn[i] : z[i];
			k3[i] = accept ? n3[i] : k3[i];
			k4[i] = accept ? n4[i] : k4[i];
			k5[i] = accept ? n5[i] : k5[i];
			k6[i] = accept ? n6[i] : k6[i];
			k7[i] = accept ? n7[i] : k7[i];
		}
	}

	// this lane still has to step to reach sample time ts (n_step_iterator's less_with_sign: by more than epsilon)
	__device__ __forceinline__ bool behind(double ts) const { return !failed & (ts - t > 2.220446049250313e-16); }
	// continuous extension on the last step [tOld, t] (odeint runge_kutta_dopri5::calc_state)
	__device__ __forceinline__ void dense(double ts, double (&out)[NZ]) const
	{
#pragma clang fp contract(off)
		constexpr double b1 = 35.0 / 384, b3 = 500.0 / 1113, b4 = 125.0 / 192, b5 = -2187.0 / 6784, b6 = 11.0 / 84;
		const double h = t - tOld;
		const bool moved = h > 0.0;
		const double th = (ts - tOld) / (moved ? h : 1.0);
		const double X1 = 5.0 * (2558722523.0 - 31403016.0 * th) / 11282082432.0;
		const double X3 = 100.0 * (882725551.0 - 15701508.0 * th) / 32700410799.0;
		const double X4 = 25.0 * (443332067.0 - 31403016.0 * th) / 1880347072.0;
		const double X5 = 32805.0 * (23143187.0 - 3489224.0 * th) / 199316789632.0;
		const double X6 = 55.0 * (29972135.0 - 7076736.0 * th) / 822651844.0;
		const double X7 = 10.0 * (7414447.0 - 829305.0 * th) / 29380423.0;
		const double thm1 = th - 1.0, thsq = th * th;
		const double A = thsq * (3.0 - 2.0 * th), B = thsq * thm1, C = thsq * thm1 * thm1, D = th * thm1 * thm1;
		const double bt1 = A * b1 - C * X1 + D, bt3 = A * b3 + C * X3, bt4 = A * b4 - C * X4, bt5 = A * b5 + C * X5,
		             bt6 = A * b6 - C * X6, bt7 = B + C * X7;
#pragma unroll
		for (int i = 0; i < NZ; i++) {
			const double v = zOld[i] + h * (bt1 * k1[i] + bt3 * k3[i] + bt4 * k4[i] + bt5 * k5[i] + bt6 * k6[i] + bt7 * k7[i]);
			out[i] = failed ? __builtin_nan("") : (moved ? v : z[i]);
		}
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
