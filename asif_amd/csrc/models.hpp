// models.hpp -- device models: the user callbacks of the reference's example programs as
// compile-time device functors (host std::function callbacks cannot run in a kernel).
// Layouts follow the reference: Dh column-major npSS x nx (Dh[i + j*npSS]), g column-major nx x nu,
// Df column-major nx x nx, Dg index i + k*nx + j*nx*nu, Du column-major nu x nx.
// The numeric constants are the workload definition and are restated from the cited example files.
#pragma once
#include <type_traits>
#include <utility>
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
	// 1 when the saturation's constants are ordinary: 0 < bevelStart < bevelStop, satSharpness and bevelStop within
	// 2^+-100, finite bounds lb < ub.  The trajectory kernels then run their steps on BackupLoop's fast forms (fewer
	// compares, sqrt / divide without rescaling steps; same bits); otherwise on the generic ones.
	int satFastOk;
	// 1 (default): blocks of Euler steps that start clear of both bevels run without the bevel code (bevel_rate below);
	// 0 (ASIF_HIP_BEVEL_FREE=0, a developer switch): every block on the full fast step; 2: the prediction without its
	// margin (and, in the TB pass, without asking whether a lane is about to reach the backup set), so that many blocks
	// meet a bevel or an arrival after all and are repeated (a test's way to that path).  Same bits always.
	int bevelFree;
	// input range of the soft saturation, host-evaluated: ub-lb, (ub+lb)/2 and 2/(ub-lb).  The kernel
	// forms uc = (u - middle) * twoOverRange where the reference divides, 2*(u-middle)/range
	// (src/asif_implicit.cpp:696): one rounding of difference, no FP64 divide in the 5000-step loop.
	double satRange, satMiddle, twoOverRange;
	// ASIFimplicitRB (include/asif_implicit_robust.h:24-37)
	double backContDt;
	double xUnc[ASIF_HIP_MAX_NX];
	int nDebug;      // already validated against npBT (initialize(), src/asif_implicit_robust.cpp:298-303)
	int useLearning;
	int npKeep;      // class ASIF: rows kept per call (npSSmax clamped to npSS, src/asif.cpp:21)
	int integrator;  // 0 forward Euler, 1 dopri5 with dense output (the reference's USE_ODEINT build)
	double trajAbsTol, trajRelTol;
	// LearningData (include/asif_learning_utils.h:8-32) uploaded by asif_hip_set_learning: device pointers
	struct Learn {
		int dHidden[2], dHidden2[2], dOut[2]; // [0] drift network, [1] actuation network
		const double *w1[2], *b1[2], *w2[2], *b2[2], *w3[2], *b3[2];
	} learn;
};

// lower end of the box safety set {-x0+hi, x0-lo, x1-lo, -x1+hi} over x +- unc, evaluated the way libaffa
// evaluates the same expressions on AAF operands (src/asif_implicit_robust.cpp:640-647):
//   AAF(interval(a,b)): centre (b+a)/2, one symbol with coefficient (b-a)/2   aa_aafcommon.cpp:81-100
//   unary minus / +- double: centre only, coefficient sign                    aa_aafarithm.cpp
//   convert().left(): centre - sum |coefficient|                              aa_aafcommon.cpp:217-245
// Additions only (the halvings are exact), so no contraction issue: bit-identical to the reference.
__device__ __forceinline__ void box_safety_lo(const double (&x)[2], const double *unc, double lo, double hi,
                                              double (&h)[4])
{
	double c[2], d[2];
#pragma unroll
	for (int i = 0; i < 2; i++) {
		const double a = x[i] - unc[i], b = x[i] + unc[i];
		c[i] = (b + a) / 2;
		d[i] = (b - a) / 2;
		d[i] = d[i] >= 0.0 ? d[i] : -d[i];
	}
	h[0] = (-c[0] + hi) - d[0];
	h[1] = (c[0] - lo) - d[0];
	h[2] = (c[1] - lo) - d[1];
	h[3] = (-c[1] + hi) - d[1];
}

// sin and cos together for the trajectory loops (|x| <= 1e5; beyond that ocml's sincos takes over):
// two-term Cody-Waite reduction by pi/2 carried by FMAs, then the fdlibm minimax kernels on
// [-pi/4, pi/4]; within 2 ulp of the correctly rounded values the oracle's glibc returns.  ocml's
// general sincos costs about twice the instructions (Payne-Hanek path, extra branches) and this call
// sits inside a 5000-step dependent loop.
// a*b + c with a wave-uniform c held in an SGPR pair (VOP3 v_fma_f64: one scalar operand is allowed)
__device__ __forceinline__ double fma_sc(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
	return r;
#else
	return fma(a, b, c); // the host examples reuse the device functors as plain functions
#endif
}

// a*b + c as the three-address instruction, every operand in vector registers
__device__ __forceinline__ double fma3(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
#else
	return fma(a, b, c);
#endif
}

// A wave-uniform 64-bit constant held in a VGPR pair for as long as it is live.  FP64 instructions of gfx950 take no
// 64-bit literal: a constant that is not in a register when it is needed costs two s_mov_b32 per use, and the ~100
// scalar registers do not hold the segway step's ~110 constants -- the compiler re-materialises them, 30-40 scalar
// moves per Euler step, and spills scalars to vector lanes on top (v_readlane to get them back).  The empty asm makes
// the value opaque (nothing to re-materialise) and is free of side effects, so it is hoisted out of the loops and
// merged where it repeats; which constants are pinned is chosen per role against the vector registers the role has
// to spare (Segway::kPinX / kPinQ).  The value itself is untouched: same bits.
template <bool PIN>
__device__ __forceinline__ double vk(double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if constexpr (PIN) asm("" : "+v"(c));
#endif
	return c;
}

// How the fast path's argument range (|x| <= 1e5) is policed (the same constant selects the backup loop's step form:
// anything but kTrigChecked also takes BackupLoop::saturateSoft<FAST>, which needs DevOptions::satFastOk):
//   kTrigChecked    one wave-level test per call and a branch that sane trajectories never take;
//   kTrigPoison     no per-call branch at all -- a lane whose argument is outside the range (or NaN) gets NaN for both
//                   results instead.  The NaN travels with the trajectory state, the caller looks for it once per block
//                   of samples and repeats that block with the checking version (rollback in k_implicit.hip), so the
//                   results are those of the checking version bit for bit while the common step saves a wave-level
//                   branch (~9 % of the pendulum's Euler step);
//   kTrigUnchecked  nothing here: the caller bounds the arguments of a whole block by other means (the pendulum's
//                   safety margin pi - max(|x0|,|x1|) is tracked per sample anyway and bounds |x0|) and repeats the
//                   block with the checking version when the bound fails -- five VALU issues per step less than the
//                   poison.  NaN and +-inf arguments need no policing: the reduction below turns them into NaN for
//                   both results, as libm does.
constexpr int kTrigChecked = 0, kTrigPoison = 1, kTrigUnchecked = 2, kTrigCarried = 3;
constexpr double kTrigFastRange = 1e5;
// the fast trajectory step's alarm: a state component that is NaN or beyond this magnitude at the end of a block (of the
// pass) sends the block (the pass) through the checking step; below it no product of the closed loop can overflow
constexpr double kStateSane = 1e150;
//   kTrigCarried    (models that declare kTrigCarry) sin / cos of the state's angle are CARRIED along the Euler steps:
//                   evaluated by the fast path at the first sample of a block, then rotated by the angle's increment,
//                   sin(a + d) = sin a + (sin a (cos d - 1) + cos a sin d) and its twin, with short polynomials for
//                   sin d and cos d - 1 (|d| <= kTrigCarryMaxStep: truncation below 1e-17).  The correction is O(d), so
//                   each step adds half an ulp of the result from the final addition only: <= 1.5 + n / 2 ulp after n
//                   steps of a block (16: <= 9.5, typically 2-3) -- against 1.5 ulp for an evaluation per step, at
//                   17 instead of 35 vector instructions.  Policed per block like kTrigUnchecked, with the bound on the
//                   angle's increment added (trigCarryBounded).  Pass 2 re-integrates a block from its first sample with
//                   the same arithmetic, so both passes see the same bits.
struct TrigCarry {
	double x, s, c; // angle at which s = sin, c = cos hold
};
constexpr double kTrigCarryMaxStep = 0.05; // truncation of the two polynomials there: 5e-18 and 3e-20
// VREGS: the Horner steps as three-address v_fma with every constant in a vector register (14 registers for the 7
// constants); false leaves the choice to the compiler (constants in scalar registers, two-address v_fmac) -- for the
// segway's kernels, which have no vector register to spare.  Same values either way.
template <bool VREGS = true>
__device__ __forceinline__ void sincos_carry(double x, TrigCarry &cy)
{
#pragma clang fp contract(on)
	const double d = x - cy.x;
	const double d2 = d * d;
	// sin d to d^7 and cos d - 1 to d^8: truncation d^9 / 9! and d^10 / 10!, 5e-18 and 3e-20 at |d| = 0.05
	// (Horner steps as three-address v_fma: the compiler's two-address v_fmac first copies the constant addend into
	// the destination, one v_mov_b64 per step in this loop-carried context)
	auto h = [](double a, double b, double c) { return VREGS ? fma3(a, b, c) : fma(a, b, c); };
	const double sd = fma(d * d2, h(d2, h(d2, -1.98412698412698412698e-04, 8.33333333333333333333e-03), -1.66666666666666666667e-01), d);
	const double cm = d2 * h(d2, h(d2, h(d2, 2.48015873015873015873e-05, -1.38888888888888888889e-03), 4.16666666666666666667e-02), -0.5);
	const double s = cy.s, c = cy.c;
	cy.s = s + fma(s, cm, c * sd);
	cy.c = c + fma(c, cm, -(s * sd));
	cy.x = x;
}
// models whose tracked safety margin bounds their trig arguments declare kTrigBoundedByMargin + trigArgsBounded(hmin)
template <class M, class = void>
struct trig_carry : std::false_type {};
template <class M>
struct trig_carry<M, std::enable_if_t<M::kTrigCarry>> : std::true_type {};
// models that declare kBevelRate (how fast the backup input moves, in units of its normalised range per second, on the
// trajectories of the model's example) let the rows kernels run a block of steps without the saturation's bevel code
// when no lane starts the block within  rate x block length x backTrajDt  of a bevel; a prediction only -- a lane that
// meets a bevel anyway is seen and the block repeated with the full step
template <class M, class = void>
struct bevel_rate { static constexpr double value = 0.0; };
template <class M>
struct bevel_rate<M, std::enable_if_t<(M::kBevelRate > 0.0)>> { static constexpr double value = M::kBevelRate; };
template <class M, class = void>
struct trig_by_margin : std::false_type {};
template <class M>
struct trig_by_margin<M, std::enable_if_t<M::kTrigBoundedByMargin>> : std::true_type {};
template <int POISON = kTrigChecked>
__device__ __forceinline__ void sincos_fast(double x, double &s, double &c)
{
#pragma clang fp contract(on) // the same bits in every kernel this is inlined into (k_implicit.hip / k_tb.hip: two-role passes)
	const double n = rint(x * 6.36619772367581382433e-01);
	double r = fma(-n, 1.57079632679489655800e+00, x);
	r = fma(-n, 6.12323399573676603587e-17, r);
	const double z = r * r;
	// Horner steps as three-address v_fma_f64 with the coefficient in an SGPR pair.  Left to itself the
	// compiler keeps the coefficients in VGPR halves and emits v_mov_b64 + v_fmac_f64 per step (the two-address
	// form needs the addend in the destination): 16 extra VALU issues per Euler step of a 5000-step loop.
	const double ps = fma_sc(z, fma_sc(z, fma_sc(z, fma_sc(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
	                                             2.75573137070700676789e-06), -1.98412698298579493134e-04),
	                         8.33333333332248946124e-03);
	const double sr = fma(z * r, fma_sc(z, ps, -1.66666666666666324348e-01), r);
	const double pc = fma_sc(z, fma_sc(z, fma_sc(z, fma_sc(z, fma_sc(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
	                                                    -2.75573143513906633035e-07), 2.48015872894767294178e-05),
	                                -1.38888888888741095749e-03), 4.16666666666666019037e-02);
	const double cr = 1.0 - (0.5 * z - z * (z * pc));
	// quadrant q = n mod 4: odd q swaps the two, bit 1 of q negates the sine, bit 1 of q + 1 the cosine -- the signs
	// as XORs of bit 31 (shift the quadrant bit there) instead of compare + negate + select
	const int qi = (int)n;
	const bool odd = (qi & 1) != 0;
	const double ss = odd ? cr : sr, cc = odd ? sr : cr;
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned t = (unsigned)qi << 30;
	s = __hiloint2double(__double2hiint(ss) ^ (int)(t & 0x80000000u), __double2loint(ss));
	c = __hiloint2double(__double2hiint(cc) ^ (int)((t + 0x40000000u) & 0x80000000u), __double2loint(cc));
#else
	s = (qi & 2) ? -ss : ss;
	c = ((qi + 1) & 2) ? -cc : cc;
#endif
	if constexpr (POISON == kTrigUnchecked) return;
	if constexpr (POISON == kTrigPoison) {
		const bool big = !(fabs(x) <= kTrigFastRange);
		s = big ? __builtin_nan("") : s;
		c = big ? __builtin_nan("") : c;
		return;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	// The reduction above is good to |x| <= 1e5.  Beyond that (or NaN) ocml's sincos overwrites the lane's result;
	// one wave-level test and a branch that is never taken on sane trajectories, instead of an if/else whose two
	// exec-mask sequences cost ~140 cycles per Euler step in a one-wave-per-SIMD dependent loop.
	const bool big = !(fabs(x) <= kTrigFastRange);
	if (__builtin_expect(__any(big), 0)) {
		if (big) sincos(x, &s, &c);
	}
#else
	if (!(fabs(x) <= kTrigFastRange)) sincos(x, &s, &c);
#endif
}

// ---- the smallest safety margin over a run of samples ------------------------------------------------------------
// fmin / fmax of values the compiler cannot prove canonical (loop-carried, selected) cost a canonicalising v_max each in
// IEEE mode; the instructions themselves ignore a NaN operand and quiet a signalling one -- exactly fmax's contract --
// so the hot loops name them directly: one instruction per maximum.
__device__ __forceinline__ double max_num(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r;
	asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
#else
	return fmax(a, b);
#endif
}
__device__ __forceinline__ double min_num(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r;
	asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
#else
	return fmin(a, b);
#endif
}
__device__ __forceinline__ double max_abs2(double a, double b) // fmax(|a|, |b|)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r;
	asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
	return r;
#else
	return fmax(fabs(a), fabs(b));
#endif
}
// A model whose smallest margin is a non-increasing function of ONE magnitude (box |x_i| <= c: c - max |x_i|; in
// general -max_i of the negated margins, negation being exact) declares kMarginByMagnitude with
//   safetyMagnitude(o, x)  and  marginOfMagnitude(m)   such that   safetyMin(o, x) == marginOfMagnitude(safetyMagnitude(o, x))
// bit for bit.  Rounding is monotone, so the smallest margin of a run of samples is marginOfMagnitude(largest magnitude),
// again bit for bit: the run costs one maximum per sample and one subtraction per run (RunningMargin below) instead of
// the margin and a minimum per sample.
template <class M, class = void>
struct margin_by_magnitude : std::false_type {};
template <class M>
struct margin_by_magnitude<M, std::enable_if_t<M::kMarginByMagnitude>> : std::true_type {};

template <class M>
struct RunningMargin {
	double v;
	__device__ __forceinline__ void reset()
	{
		v = margin_by_magnitude<M>::value ? -__builtin_huge_val() : __builtin_huge_val();
	}
	__device__ __forceinline__ void add(const DevOptions &o, const double (&x)[M::NX])
	{
		if constexpr (margin_by_magnitude<M>::value) v = max_num(v, M::safetyMagnitude(o, x));
		else v = fmin(v, M::safetyMin(o, x));
	}
	__device__ __forceinline__ double value() const
	{
		if constexpr (margin_by_magnitude<M>::value) return M::marginOfMagnitude(v);
		else return v;
	}
};

// ---------------------------------------------------------------------------------------------
// Double integrator, examples/DoubleIntegrator.cpp:12-61.  x = (position, velocity).
struct DoubleIntegrator {
	static constexpr bool kIgnoresOptions = true; // no functor below reads DevOptions (k_explicit.hip: explicit_light_kernel)
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

// ---------------------------------------------------------------------------------------------
// Inverted pendulum with an LQR-like backup controller, examples/InvertedPendulum_Implicit.cpp:13-80.
struct InvertedPendulum {
	// plain ASIFimplicit: the rows kernel solves the instance's 3 x 41 QP itself (k_implicit.hip); its 123 doubles of rows
	// do not fit the 256 vector registers next to the solver's state -- the compiler parks them in AGPRs (the wave has the
	// SIMD's 512 to itself) and 0.7 KB of scratch: C3 762 -> 753 us, and nothing staged (34 -> 1 MB per step)
	static constexpr bool kImFuseQp = true;
	static constexpr int NX = 2, NU = 1, NPSS = 4, NPBS = 1, NPBTSS = 10;
	// samples per checkpoint block of the two-pass critical-sample search (k_implicit.hip): ~sqrt(npBT / 2 npBTSS)
	static constexpr int kTrajBlock = 16; // measured: 8 is 1.7 % faster at twice the checkpoint memory, 32 is 2.7 % slower
	// u = -3 (x0 + x1) moves by at most 0.05 of its half-range over 16 steps of 1 ms on the example's trajectories
	// (tools/scratch/bevel_coherence.py: 60 % of a wave's blocks start that far from both bevels, 0.008 % of those meet
	// one all the same)
#ifndef ASIF_PENDULUM_BEVEL_DELTA
#define ASIF_PENDULUM_BEVEL_DELTA 0.05
#endif
	static constexpr double kBevelRate = ASIF_PENDULUM_BEVEL_DELTA / (16 * 0.001);
	// g = e_last and Dg = 0 for every state: lets the backup loop drop the generic formula's products with
	// those constants (x*0, x*1, 0+x) -- same values, ~10 fewer FP64 issues per Euler step
	static constexpr bool kInputOnLastState = true;
	// first row of Df is the constant (0, 1) (x0' = x1): row 0 of DfCL Q is row 1 of Q, bit for bit (0*a + 1*b = b)
	static constexpr bool kDfFirstRowShift = true;

	// :31-37  box |theta| <= pi, |omega| <= pi
	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
		h[0] = -x[0] + kPi; Dh[0] = -1.0; Dh[4] = 0.0;
		h[1] = x[0] + kPi;  Dh[1] = 1.0;  Dh[5] = 0.0;
		h[2] = x[1] + kPi;  Dh[2] = 0.0;  Dh[6] = 1.0;
		h[3] = -x[1] + kPi; Dh[3] = 0.0;  Dh[7] = -1.0;
	}
	// min(-x0 + pi, x0 + pi, x1 + pi, -x1 + pi) = pi - max(|x0|, |x1|), bit for bit: the smaller of the two sums of
	// a pair is fl(pi - |x_k|), and rounding is monotone
	static constexpr bool kMarginByMagnitude = true;
	__device__ static double safetyMagnitude(const DevOptions &, const double (&x)[NX]) { return max_abs2(x[0], x[1]); }
	__device__ static double marginOfMagnitude(double m) { return kPi - m; }
	__device__ static double safetyMin(const DevOptions &o, const double (&x)[NX]) { return marginOfMagnitude(safetyMagnitude(o, x)); }
	// |x0| <= pi - safetyMin: while the smallest margin of a block of samples stays above this floor, no sample of
	// the block hands sincos_fast an argument near the end of its range (kTrigFastRange; the floor keeps 10 % clear of
	// it so that rounding in the margin cannot matter) -- see kTrigUnchecked
	static constexpr bool kTrigBoundedByMargin = true;
	__device__ static bool trigArgsBounded(double hmin) { return hmin >= -0.9 * kTrigFastRange; }
	// the same set on interval_t operands over x +- x_unc (ASIFimplicitRB's safetySet_int), lower ends
	__device__ static void safetySetLo(const DevOptions &o, const double (&x)[NX], double (&h)[NPSS])
	{
		box_safety_lo(x, o.xUnc, -kPi, kPi, h);
	}
	// :39-52  ellipsoid Pv - x'Px >= 0, P = [1.25 .25; .25 .25]; gradient -(P+P')x
	__device__ static void backupSet(const DevOptions &, const double (&x)[NX], double &h, double (&Dh)[NX],
	                                 double (&DDh)[NX * NX])
	{
		const double P00 = 1.25, P10 = 0.25, P01 = 0.25, P11 = 0.25;
		double v = 0.05;
		v -= P00 * x[0] * x[0];
		v -= P01 * x[0] * x[1];
		v -= P10 * x[1] * x[0];
		v -= P11 * x[1] * x[1];
		h = v;
		Dh[0] = -2.5 * x[0] + -0.5 * x[1];
		Dh[1] = -0.5 * x[0] + -0.5 * x[1];
		DDh[0] = -2.5; DDh[1] = -0.5; DDh[2] = -0.5; DDh[3] = -0.5; // not used by ASIFimplicit
	}
	// :55-62 and :73-80  f = (omega, sin theta), g = (0,1); Df = [0 1; cos theta 0], Dg = 0
	__device__ static void dynamics(const DevOptions &, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		f[0] = x[1];
		f[1] = sin(x[0]);
		g[0] = 0.0;
		g[1] = 1.0;
	}
	template <int POISON = kTrigChecked>
	__device__ static void dynamicsAndGradients(const DevOptions &, const double (&x)[NX], double (&f)[NX],
	                                            double (&g)[NX * NU], double (&Df)[NX * NX], double (&Dg)[NX * NU * NX])
	{
		double s, c;
		sincos_fast<POISON>(x[0], s, c);
		f[0] = x[1];
		f[1] = s;
		g[0] = 0.0;
		g[1] = 1.0;
		Df[0] = 0.0; Df[2] = 1.0;
		Df[1] = c;   Df[3] = 0.0;
#pragma unroll
		for (int i = 0; i < NX * NU * NX; i++) Dg[i] = 0.0;
	}
	// the same with the angle's sin / cos carried from the previous step (kTrigCarried); reset: first step of a block
	static constexpr bool kTrigCarry = true;
	static constexpr int kTrigAngle = 0; // the state component whose sin / cos are carried
	__device__ static void dynamicsAndGradientsCarried(const DevOptions &, const double (&x)[NX], double (&f)[NX],
	                                                   double (&g)[NX * NU], double (&Df)[NX * NX],
	                                                   double (&Dg)[NX * NU * NX], TrigCarry &cy, bool reset)
	{
		if (reset) {
			sincos_fast<kTrigUnchecked>(x[0], cy.s, cy.c);
			cy.x = x[0];
		} else {
			sincos_carry(x[0], cy);
		}
		f[0] = x[1];
		f[1] = cy.s;
		g[0] = 0.0;
		g[1] = 1.0;
		Df[0] = 0.0; Df[2] = 1.0;
		Df[1] = cy.c; Df[3] = 0.0;
#pragma unroll
		for (int i = 0; i < NX * NU * NX; i++) Dg[i] = 0.0;
	}
	// |x0' | = |x1|: the angle's increment per step is dt |x1| <= dt (pi - hmin)
	__device__ static bool trigCarryBounded(const DevOptions &o, double hmin)
	{
		return trigArgsBounded(hmin) && (kPi - hmin) * o.trajDt <= kTrigCarryMaxStep;
	}
	// :63-71  u = K x, K = (-3,-3)
	__device__ static void backupController(const DevOptions &, const double (&x)[NX], double (&u)[NU], double (&Du)[NU * NX])
	{
#pragma clang fp contract(on)
		u[0] = -3.0 * x[0] + -3.0 * x[1];
		Du[0] = -3.0;
		Du[1] = -3.0;
	}
};

// 1 / d by the hardware seed and two Newton steps: within an ulp of the IEEE quotient for a normal d (no rescaling, no
// fix-up: a third of the division's instructions).  For the segway's four reciprocals of O(1-10) denominators.
__device__ __forceinline__ double rcp_newton(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
	double r = __builtin_amdgcn_rcp(d);
	r = fma(fma(-d, r, 1.0), r, r);
	r = fma(fma(-d, r, 1.0), r, r);
	return r;
#else
	return 1.0 / d;
#endif
}

// tanh for the segway's friction term tanh(1000 v): (1 - E) / (1 + E) with E = exp(-2|y|) -- |y| clipped at 25, where
// E is far below half an ulp of 1 -- E by ln2 reduction and a degree-12 Taylor polynomial on |r| <= ln2 / 2 (truncation
// 1.7e-16 relative), the quotient by reciprocal seed, two Newton steps and one residual correction.  Absolute error
// below 4e-16 over the whole line, which is what matters here: the model uses th and th^2 additively (a relative
// error bound near y = 0 would cost the expm1 form).  ocml's tanh is ~160 vector instructions of double-double
// arithmetic, a fifth of the segway's Euler step; this is ~35.  NaN stays NaN.
// The constants come in through `c` (tanh_const(0..12) in order): a caller that runs this in a loop may hold them in
// vector registers (vk above; made before the loop) and then asks for V3, the Horner steps as the three-address
// instruction -- with the addend in a vector register the compiler's two-address form would copy it first.
constexpr int kTanhConsts = 13;
constexpr double tanh_const(int i)
{
	constexpr double t[kTanhConsts] = {
		1.44269504088896338700e+00, 6.93147180369123816490e-01, 1.90821492927058770002e-10, // 1/ln 2, ln 2 high, low
		2.08767569878680989792e-09, 2.50521083854417187751e-08, 2.75573192239858906526e-07, // 1/12!, 1/11!, 1/10!
		2.75573192239858906526e-06, 2.48015873015873015873e-05, 1.98412698412698412698e-04, // 1/9!, 1/8!, 1/7!
		1.38888888888888888889e-03, 8.33333333333333333333e-03, 4.16666666666666666667e-02, // 1/6!, 1/5!, 1/4!
		1.66666666666666666667e-01,                                                         // 1/3!
	};
	return t[i];
}
struct TanhConsts { double v[kTanhConsts]; };
template <bool PIN, int... I>
__device__ __forceinline__ void fill_tanh_consts(TanhConsts &c, std::integer_sequence<int, I...>)
{
	((c.v[I] = vk<PIN>(tanh_const(I))), ...);
}
template <bool PIN = false>
__device__ __forceinline__ TanhConsts tanh_consts()
{
	TanhConsts c;
	fill_tanh_consts<PIN>(c, std::make_integer_sequence<int, kTanhConsts>());
	return c;
}
template <bool V3 = false>
__device__ __forceinline__ double tanh_abs_accurate(double y, const TanhConsts &c = tanh_consts<false>())
{
#if defined(__HIP_DEVICE_COMPILE__)
	auto h = [](double a, double b, double k) { return V3 ? fma3(a, b, k) : fma(a, b, k); };
	const double a = fabs(y);
	const double x = -2.0 * (a < 25.0 ? a : 25.0);
	const double n = rint(x * c.v[0]);
	double r = fma(-n, c.v[1], x);
	r = fma(-n, c.v[2], r);
	double p = c.v[3];
	p = h(p, r, c.v[4]);
	p = h(p, r, c.v[5]);
	p = h(p, r, c.v[6]);
	p = h(p, r, c.v[7]);
	p = h(p, r, c.v[8]);
	p = h(p, r, c.v[9]);
	p = h(p, r, c.v[10]);
	p = h(p, r, c.v[11]);
	p = h(p, r, c.v[12]);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	const double E = ldexp(p, (int)n);
	const double num = 1.0 - E, den = 1.0 + E;
	double rc = __builtin_amdgcn_rcp(den);
	rc = fma(fma(-den, rc, 1.0), rc, rc);
	rc = fma(fma(-den, rc, 1.0), rc, rc);
	double q = num * rc;
	q = fma(fma(-den, q, num), rc, q);
	q = copysign(q, y);
	return (y != y) ? y : q;
#else
	return tanh(y);
#endif
}

// ---------------------------------------------------------------------------------------------
// Synthetic two-input model for class ASIF.  NOT one of the reference's examples (none of them has nu > 1); it
// exists so that the nu > 1 code of src/asif.cpp (:279-303 Lgh = Dh g with nu columns, :314-352 cost and clamp per
// input) has a device path: x' = F x + G u with a non-diagonal G, five half-planes r_i - a_i . x >= 0.
struct PlanarTwoInput {
	static constexpr int NX = 2, NU = 2, NPSS = 5;
	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
		const double a[5][2] = {{1., 0.}, {-1., 0.}, {0., 1.}, {0., -1.}, {0.6, 0.8}}, r[5] = {1., 1., 1., 1., 1.2};
#pragma unroll
		for (int i = 0; i < 5; i++) {
			h[i] = r[i] - a[i][0] * x[0] - a[i][1] * x[1];
			Dh[i] = -a[i][0];
			Dh[i + 5] = -a[i][1];
		}
	}
	__device__ static void dynamics(const DevOptions &, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		f[0] = -0.5 * x[0] + 0.2 * x[1];
		f[1] = 0.1 * x[0] + -0.3 * x[1];
		g[0] = 1.0; g[1] = 0.0;
		g[2] = 0.3; g[3] = 1.0;
	}
};

// ---------------------------------------------------------------------------------------------
// Double integrator with an LQR-like backup controller, examples/DoubleIntegrator_implicit.cpp:13-90:
// plain box |x|,|v| <= 1 (not the braking parabola of examples/DoubleIntegrator.cpp), ellipsoidal backup set.
// Sums are accumulated from 0.0 like the reference's matrixVectorMultiply (include/asif_utils.h:46-62).
struct DoubleIntegratorImplicit {
	static constexpr bool kImFuseQp = true; // plain ASIFimplicit: the rows kernel solves the instance's 3 x 17 QP itself (k_implicit.hip)
	static constexpr int NX = 2, NU = 1, NPSS = 4, NPBS = 1, NPBTSS = 4;
	static constexpr int kTrajBlock = 4; // 201-sample trajectory, 4 critical samples (measured: 4 < 8 < 16)
	// g = e_last and Dg = 0 for every state: lets the backup loop drop the generic formula's products with
	// those constants (x*0, x*1, 0+x) -- same values, ~10 fewer FP64 issues per Euler step
	static constexpr bool kInputOnLastState = true;
	// first row of Df is the constant (0, 1) (x0' = x1): row 0 of DfCL Q is row 1 of Q, bit for bit (0*a + 1*b = b)
	static constexpr bool kDfFirstRowShift = true;

	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
		h[0] = -x[0] + 1.0;    Dh[0] = -1.0; Dh[4] = 0.0;
		h[1] = x[0] - (-1.0);  Dh[1] = 1.0;  Dh[5] = 0.0;
		h[2] = x[1] - (-1.0);  Dh[2] = 0.0;  Dh[6] = 1.0;
		h[3] = -x[1] + 1.0;    Dh[3] = 0.0;  Dh[7] = -1.0;
	}
	// min(-x0 + 1, x0 - (-1), x1 - (-1), -x1 + 1) = 1 - max(|x0|, |x1|), bit for bit (see InvertedPendulum)
	static constexpr bool kMarginByMagnitude = true;
	__device__ static double safetyMagnitude(const DevOptions &, const double (&x)[NX]) { return max_abs2(x[0], x[1]); }
	__device__ static double marginOfMagnitude(double m) { return 1.0 - m; }
	__device__ static double safetyMin(const DevOptions &o, const double (&x)[NX]) { return marginOfMagnitude(safetyMagnitude(o, x)); }
	// the same set on interval_t operands over x +- x_unc (ASIFimplicitRB's safetySet_int), lower ends
	__device__ static void safetySetLo(const DevOptions &o, const double (&x)[NX], double (&h)[NPSS])
	{
		box_safety_lo(x, o.xUnc, -1.0, 1.0, h);
	}
	// :42-55  h = Pv - x'Px;  Dh = mPpPt x with the shipped mPpPt = {-1, -0.577.., -0.577.., +1} (last entry is
	// not -(P+P')(1,1) = -1; reproduced as is)
	__device__ static void backupSet(const DevOptions &, const double (&x)[NX], double &h, double (&Dh)[NX],
	                                 double (&DDh)[NX * NX])
	{
		const double P00 = 0.500000000000000, P10 = 0.288675134594813, P01 = 0.288675134594813, P11 = 0.5;
		double v = 0.002;
		v -= P00 * x[0] * x[0];
		v -= P01 * x[0] * x[1];
		v -= P10 * x[1] * x[0];
		v -= P11 * x[1] * x[1];
		h = v;
		Dh[0] = (0.0 + -1.0 * x[0]) + -0.577350269189626 * x[1];
		Dh[1] = (0.0 + -0.577350269189626 * x[0]) + 1.0 * x[1];
		DDh[0] = 0.0; DDh[1] = 0.0; DDh[2] = 0.0; DDh[3] = 0.0; // not used by ASIFimplicit
	}
	// :57-63, :75-80  f = A x, g = B;  Df = A, Dg = 0
	__device__ static void dynamics(const DevOptions &, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		f[0] = (0.0 + 0.0 * x[0]) + 1.0 * x[1];
		f[1] = (0.0 + 0.0 * x[0]) + 0.0 * x[1];
		g[0] = 0.0;
		g[1] = 1.0;
	}
	template <int POISON = kTrigChecked>
	__device__ static void dynamicsAndGradients(const DevOptions &o, const double (&x)[NX], double (&f)[NX],
	                                            double (&g)[NX * NU], double (&Df)[NX * NX], double (&Dg)[NX * NU * NX])
	{
		dynamics(o, x, f, g);
		Df[0] = 0.0; Df[2] = 1.0;
		Df[1] = 0.0; Df[3] = 0.0;
#pragma unroll
		for (int i = 0; i < NX * NU * NX; i++) Dg[i] = 0.0;
	}
	// :65-73  u = K x, K = (-10, -20)
	__device__ static void backupController(const DevOptions &, const double (&x)[NX], double (&u)[NU], double (&Du)[NU * NX])
	{
#pragma clang fp contract(on)
		u[0] = (0.0 + -10.0 * x[0]) + -20.0 * x[1];
		Du[0] = -10.0;
		Du[1] = -20.0;
	}
};

// ---------------------------------------------------------------------------------------------
// Double integrator of the time-to-backup-set example, examples/DoubleIntegrator_implicit_tb.cpp:13-103: box,
// dynamics and backup controller of the implicit example above; the backup set is the disc x'x <= Pv^2 (P = I,
// Pv = 0.01).  The example fills two of the Hessian's four entries (:49) and the class reads four
// (src/asif_implicit_tb.cpp:494,623): taken as the whole of mPpPt = -2 I here and in the oracle (or_models.c).
struct DoubleIntegratorTB : DoubleIntegratorImplicit {
	static constexpr int kTrajBlock = 16; // 2 101-sample trajectory (7 001 after the example's updateOptions)
	static constexpr bool kTbFuseQp = true; // the rows kernel solves the instance's 2 x 18 QP itself (k_tb.hip): C12 360 -> 344 us
	// (no kBevelRate: the bevel-free form of the block measured 8 % SLOWER on this small step, C12 350 -> 378 us)
	__device__ static bool arrivalFar(const DevOptions &, const double (&)[NX], int) { return false; } // (unused: no quiet blocks)
	static constexpr bool kTbUnrollSteps = true;
	// ... and a full block of this small step in line, no loop control (C12 411 -> 350 us; the pendulum's larger step
	// gains nothing from eight or sixteen copies)
	static constexpr bool kTbUnrollWholeBlock = true;
	// :41-57  h = Pv^2 - sum_ij P_ij x_i x_j in the example's loop order, Dh = mPpPt x, DDh = mPpPt
	// (fp contract(off) in these two: every product and difference separately rounded, as the reference's loop and the
	// oracle have them, in whatever kernel they are inlined into -- hipcc's default fuses by use counts that differ between
	// the Euler pass, the dopri5 kernel and the rows; the sign of this value picks the hit sample)
	__device__ static double backupSetValue(const DevOptions &, const double (&x)[NX])
	{
#pragma clang fp contract(off)
		double v = 0.01 * 0.01;
		v -= 1.0 * x[0] * x[0];
		v -= 0.0 * x[0] * x[1];
		v -= 0.0 * x[1] * x[0];
		v -= 1.0 * x[1] * x[1];
		return v;
	}
	__device__ static bool backupSetInside(const DevOptions &o, const double (&x)[NX]) { return backupSetValue(o, x) >= 0.0; }
	__device__ static void backupSet(const DevOptions &o, const double (&x)[NX], double &h, double (&Dh)[NX],
	                                 double (&DDh)[NX * NX])
	{
#pragma clang fp contract(off)
		h = backupSetValue(o, x);
		Dh[0] = (0.0 + -2.0 * x[0]) + 0.0 * x[1];
		Dh[1] = (0.0 + 0.0 * x[0]) + -2.0 * x[1];
		DDh[0] = -2.0; DDh[1] = 0.0; DDh[2] = 0.0; DDh[3] = -2.0;
	}
};

// ---------------------------------------------------------------------------------------------
// Inverted pendulum of the time-to-backup-set example, examples/InvertedPendulum_ImplicitTB.cpp:14-99:
// asymmetric box, half-space backup set x0 >= pi/2 - 0.1, backup controller tracking the velocity pi/10.
// Dynamics and their gradients are the pendulum's (:67-74, :87-94).
struct InvertedPendulumTB {
	static constexpr int NX = 2, NU = 1, NPSS = 4, NPBS = 1, NPBTSS = 4;
	static constexpr int kTrajBlock = 32; // 11 551-sample trajectory, 4 critical samples (measured: 16 = 32 < 64 < 128)
	// what the backup input moves in a 32-step block, with room: margins of 0.05 / 0.1 / 0.2 of the half-range measured
	// 2 043 / 2 009 / 1 981 us on C8 (2 245 without the bevel-free form)
	static constexpr double kBevelRate = 0.2 / (32 * 0.001);
	static constexpr bool kTbUnrollSteps = true; // k_tb.hip: full blocks unrolled by four
	static constexpr bool kTbUnrollWholeBlock = false;
	// g = e_last and Dg = 0 for every state: lets the backup loop drop the generic formula's products with
	// those constants (x*0, x*1, 0+x) -- same values, ~10 fewer FP64 issues per Euler step
	static constexpr bool kInputOnLastState = true;
	// first row of Df is the constant (0, 1) (x0' = x1): row 0 of DfCL Q is row 1 of Q, bit for bit (0*a + 1*b = b)
	static constexpr bool kDfFirstRowShift = true;

	// :28-34  -pi/2 <= theta <= pi, |omega| <= pi/2
	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
		h[0] = -x[0] + kPi;        Dh[0] = -1.0; Dh[4] = 0.0;
		h[1] = x[0] - (-kPi / 2.); Dh[1] = 1.0;  Dh[5] = 0.0;
		h[2] = x[1] - (-kPi / 2.); Dh[2] = 0.0;  Dh[6] = 1.0;
		h[3] = -x[1] + kPi / 2.;   Dh[3] = 0.0;  Dh[7] = -1.0;
	}
	// min(-x0 + pi, x0 + pi/2, x1 + pi/2, -x1 + pi/2) = -max(x0 - pi, -x0 - pi/2, |x1| - pi/2), bit for bit: fl(a - b) =
	// -fl(b - a), the smaller of the last two sums is fl(pi/2 - |x1|), and rounding is monotone
	static constexpr bool kMarginByMagnitude = true;
	__device__ static double safetyMagnitude(const DevOptions &, const double (&x)[NX])
	{
		return max_num(max_num(x[0] - kPi, -x[0] - kPi / 2.), fabs(x[1]) - kPi / 2.);
	}
	__device__ static double marginOfMagnitude(double m) { return -m; }
	__device__ static double safetyMin(const DevOptions &o, const double (&x)[NX]) { return marginOfMagnitude(safetyMagnitude(o, x)); }
	// -pi/2 - hmin <= x0 <= pi - hmin: see InvertedPendulum::trigArgsBounded
	static constexpr bool kTrigBoundedByMargin = true;
	__device__ static bool trigArgsBounded(double hmin) { return hmin >= -0.9 * kTrigFastRange; }
	// :36-65
	__device__ static void backupSet(const DevOptions &, const double (&x)[NX], double &h, double (&Dh)[NX],
	                                 double (&DDh)[NX * NX])
	{
		h = x[0] - kPi / 2. + 0.1;
		Dh[0] = 1.;
		Dh[1] = 0.;
		DDh[0] = 0.; DDh[1] = 0.; DDh[2] = 0.; DDh[3] = 0.;
	}
	__device__ static double backupSetValue(const DevOptions &, const double (&x)[NX]) { return x[0] - kPi / 2. + 0.1; }
	__device__ static bool backupSetInside(const DevOptions &o, const double (&x)[NX]) { return backupSetValue(o, x) >= 0.0; }
	// cannot arrive within `steps` Euler steps: the angle moves by at most (|omega| + 0.5) dt per step over a block (the
	// 0.5 rad/s covers what the bounded input and gravity add to omega in 32 ms several times over).  A prediction for
	// k_tb.hip's quiet blocks; an arrival it missed is seen and the block repeated.
	__device__ static bool arrivalFar(const DevOptions &o, const double (&x)[NX], int steps)
	{
		return backupSetValue(o, x) + (double)steps * o.trajDt * (fabs(x[1]) + 0.5) < 0.0;
	}
	// :76-85  u = K (vDes - omega)
	__device__ static void backupController(const DevOptions &, const double (&x)[NX], double (&u)[NU], double (&Du)[NU * NX])
	{
#pragma clang fp contract(on)
		u[0] = 10. * ((kPi / 10.) - x[1]);
		Du[0] = 0.;
		Du[1] = -10.;
	}
	__device__ static void dynamics(const DevOptions &o, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		InvertedPendulum::dynamics(o, x, f, g);
	}
	template <int POISON = kTrigChecked>
	__device__ static void dynamicsAndGradients(const DevOptions &o, const double (&x)[NX], double (&f)[NX],
	                                            double (&g)[NX * NU], double (&Df)[NX * NX], double (&Dg)[NX * NU * NX])
	{
		InvertedPendulum::dynamicsAndGradients<POISON>(o, x, f, g, Df, Dg);
	}
	static constexpr bool kTrigCarry = true;
	static constexpr int kTrigAngle = 0;
	__device__ static void dynamicsAndGradientsCarried(const DevOptions &o, const double (&x)[NX], double (&f)[NX],
	                                                   double (&g)[NX * NU], double (&Df)[NX * NX],
	                                                   double (&Dg)[NX * NU * NX], TrigCarry &cy, bool reset)
	{
		InvertedPendulum::dynamicsAndGradientsCarried(o, x, f, g, Df, Dg, cy, reset);
	}
	// |x1| <= pi/2 - hmin
	__device__ static bool trigCarryBounded(const DevOptions &o, double hmin)
	{
		return trigArgsBounded(hmin) && (kPi / 2. - hmin) * o.trajDt <= kTrigCarryMaxStep;
	}
};

// ---------------------------------------------------------------------------------------------
// Segway, examples/segway_implicit_tb.cpp:13-212 (MATLAB-generated dynamics and Jacobians).
// x = (position, velocity, pitch, pitch rate).  The friction factor of f is multiplied by 0.0 in the
// reference (:78) so its terms vanish identically; the Jacobian was generated WITH the tanh friction
// model and is reproduced as shipped.
struct Segway {
	static constexpr int NX = 4, NU = 1, NPSS = 4, NPBS = 1, NPBTSS = 4;
	static constexpr int kTrajBlock = 4; // 316-sample trajectory, 4 critical samples (measured: 4 < 8 < 2 < 16)
	static constexpr bool kTbUnrollSteps = false;
	static constexpr bool kTbUnrollWholeBlock = false;
	__device__ static bool arrivalFar(const DevOptions &, const double (&)[4], int) { return false; } // (unused: no quiet blocks)
	// the gradients (tanh, Df, Dg) and the 4 x 4 sensitivity are the larger half of the Euler step and x does not
	// depend on them: pass 1 of the TB kernel may deal x and Q to two waves (k_tb.hip: tb_rows_split_kernel)
	static constexpr bool kTbSplitRoles = true;
	static constexpr bool kTbFuseQp = true; // the rows kernel solves the instance's 2 x 18 QP itself (k_tb.hip)
	static constexpr bool kInputOnLastState = false; // g depends on the pitch
	static constexpr bool kDfFirstRowShift = false;

	__device__ static double xb(int i) { return i < 2 ? 3.0 : (i == 2 ? kPi / 6 : kPi); }
	// f0 = x1 and f2 = x3 with g0 = g2 = 0: rows 0 and 2 of Df are unit rows, of Dg zero (BackupLoop: rows 0 and 2 of
	// DfCL Q are rows 1 and 3 of Q); Dg is non-zero at (1,2) and (3,2) only (column-major entries 9 and 11)
	static constexpr unsigned kDfUnitRowMask = 0x5u;
	static constexpr unsigned kDgMask = (1u << 9) | (1u << 11);

	// :27-38  h_i = xBound_i^2 - x_i^2
	__device__ static void safetySet(const DevOptions &, const double (&x)[NX], double (&h)[NPSS], double (&Dh)[NPSS * NX])
	{
#pragma unroll
		for (int i = 0; i < NPSS * NX; i++) Dh[i] = 0.0;
#pragma unroll
		for (int i = 0; i < NX; i++) {
			h[i] = (xb(i) * xb(i)) - (x[i] * x[i]);
			Dh[i * (NX + 1)] = -2.0 * x[i];
		}
	}
	__device__ static double safetyMin(const DevOptions &, const double (&x)[NX])
	{
		double m = (xb(0) * xb(0)) - (x[0] * x[0]);
#pragma unroll
		for (int i = 1; i < NX; i++) m = fmin(m, (xb(i) * xb(i)) - (x[i] * x[i]));
		return m;
	}
	// pitch^2 <= xb(2)^2 - hmin: above this floor |2 pitch| stays below 9e4, inside the trig fast path's range
	static constexpr bool kTrigBoundedByMargin = true;
	__device__ static bool trigArgsBounded(double hmin) { return hmin >= -2.0e9; }
	// :40-54  Pv^2 - sum (x_i/xBound_i)^2 with gradient and (diagonal) Hessian
	__device__ static void backupSet(const DevOptions &, const double (&x)[NX], double &h, double (&Dh)[NX],
	                                 double (&DDh)[NX * NX])
	{
#pragma unroll
		for (int i = 0; i < NX * NX; i++) DDh[i] = 0.0;
		double v = 0.05 * 0.05;
#pragma unroll
		for (int i = 0; i < NX; i++) {
			const double q = x[i] / xb(i);
			v -= q * q;
			Dh[i] = -2.0 * x[i] / (xb(i) * xb(i));
			DDh[i * (NX + 1)] = -2.0 / (xb(i) * xb(i));
		}
		h = v;
	}
	__device__ static double backupSetValue(const DevOptions &, const double (&x)[NX])
	{
		double v = 0.05 * 0.05;
#pragma unroll
		for (int i = 0; i < NX; i++) {
			const double q = x[i] / xb(i);
			v -= q * q;
		}
		return v;
	}
	// sign test of the value above, once per trajectory sample (the hit test, src/asif_implicit_tb.cpp:505-528).
	// Screened: the same sum with reciprocal multiplies (four FP64 divisions less on the dependent chain of the
	// loop) decides whenever it is further from zero than its own rounding can reach; the rare lane that is not
	// gets the exact expression behind one wave-level branch, so the decision is always the exact one.
	__device__ static bool backupSetInside(const DevOptions &o, const double (&x)[NX])
	{
		double v = 0.05 * 0.05;
#pragma unroll
		for (int i = 0; i < NX; i++) {
			const double q = x[i] * (1.0 / xb(i)); // 1/xb folds to a constant
			v -= q * q;
		}
		bool in = v >= 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
		// The two sums differ by rounding only: by far less than 1e-13 (c + sum q^2) = 1e-13 (2c - v), c = Pv^2.  A value
		// within that of zero has |v| <= 2e-13 c / (1 - 1e-13) < 1e-15: the constant screens a superset, with no second
		// accumulation (NaN compares false: close).
		const bool close = !(fabs(v) > 1e-15);
		if (__any(close)) {
			if (close) in = backupSetValue(o, x) >= 0.0;
		}
#else
		in = backupSetValue(o, x) >= 0.0;
#endif
		return in;
	}
	// ---- The 64-bit constants of the Euler step.  The expressions below name them by value -- K_(group, 44.7214) --
	// and read them out of a `Consts` the caller made: `constants<0>()` (the default argument) holds plain literals and
	// folds away; `constants<PIN>()`, made ONCE before a loop, holds the groups named in PIN in vector registers
	// (vk above) for as long as the loop runs.  A negative literal reads the entry of its magnitude, negated (free: a
	// source modifier).  Which groups a role of the two-role TB pass pins: kPinX / kPinQ, chosen against the vector
	// registers that role has to spare.  Values, and therefore bits, do not depend on any of it.
	static constexpr unsigned kPinCtl = 1, kPinDyn = 2, kPinRg = 4, kPinGain = 8, kPinTanh = 16, kPinDfA = 32, kPinDfB = 64,
	                          kPinDfC = 128, kPinCarry = 256;
	// The two roles of the TB pass (k_tb.hip): the x role has registers to spare for every constant it reads, the
	// sincos_carry polynomial's included; the Q role for the gradients', the gain's, the controller's and tanh's
	// (62 + 13 constants: 150 VGPRs between the two roles, none spilled).  C4 244 -> 194 us together with the roles'
	// exec-mask form, DESIGN 4.2.
	static constexpr unsigned kPinX = kPinCtl | kPinDyn | kPinRg | kPinGain | kPinCarry;
	static constexpr unsigned kPinQ = kPinCtl | kPinGain | kPinTanh | kPinDfA | kPinDfB | kPinDfC;
	struct KEntry { unsigned grp; double mag; };
	static constexpr int kNK = 60;
	static constexpr KEntry kentry(int i)
	{
		constexpr KEntry t[kNK] = {
			{kPinCtl, 44.7214}, {kPinCtl, 44.6528}, {kPinCtl, 150.1612}, {kPinCtl, 37.6492},
			{kPinCtl, 0.1383244254},
			{kPinDyn, 14.553176960783997}, {kPinDyn, 2.0831375273848773}, {kPinDyn, 0.59146430898882}, {kPinDyn, 0.33003710190723146},
			{kPinDyn, 2.3707272057666411}, {kPinDyn, 5.8022648711803244}, {kPinDyn, 20.435579143645651}, {kPinDyn, 40.918271887954823},
			{kPinDyn, 293.92471275850022}, {kPinDyn, 2.0831375273848769},
			{kPinRg, 8.3593271361634187}, {kPinRg, 2.1243074194638587}, {kPinRg, 0.04116989207898096}, {kPinRg, 0.29573215449441},
			{kPinGain, 1.4575004011882324}, {kPinGain, 0.20290365220710288}, {kPinGain, 0.551244194154502}, {kPinGain, 4.1706936767483551},
			{kPinGain, 5.65378660671284}, {kPinGain, 2.0043013906215941},
			{kPinDfA, 15.13175750513302}, {kPinDfA, 40.918271887954823}, {kPinDfA, 3.3849959169972448}, {kPinDfA, 30.26351501026604},
			{kPinDfA, 8443.5211353581435}, {kPinDfA, 410.77609832706019}, {kPinDfA, 2950.692713500939}, {kPinDfA, 20808.641003022261},
			{kPinDfA, 2106.5440939849238}, {kPinDfA, 15131.75750513302},
			{kPinDfB, 1.18292861797764}, {kPinDfB, 4.1662750547697547}, {kPinDfB, 40.8711582872913}, {kPinDfB, 11.604529742360651},
			{kPinDfB, 2.3707272057666411}, {kPinDfB, 0.41077609832706019}, {kPinDfB, 0.0975}, {kPinDfB, 5.8022648711803244},
			{kPinDfB, 20.435579143645651}, {kPinDfB, 8.443521135358143}, {kPinDfB, 293.92471275850022}, {kPinDfB, 2.1065440939849238},
			{kPinDfB, 20.808641003022259}, {kPinDfB, 0.59146430898881985}, {kPinDfB, 2.0831375273848769},
			{kPinDfC, 0.6600742038144628}, {kPinDfC, 4.7414544115332831}, {kPinDfC, 0.1118494602519098}, {kPinDfC, 0.80343863413287053},
			{kPinDfC, 0.59146430898882}, {kPinDfC, 2.0831375273848773}, {kPinDfC, 2.2990706749044238}, {kPinDfC, 1.1471739513016379},
			{kPinDfC, 8.24039624751662}, {kPinDfC, 11.33189235811229},
		};
		return t[i];
	}
	static constexpr int kfind(unsigned grp, double lit)
	{
		for (int i = 0; i < kNK; i++)
			if (kentry(i).grp == grp && (kentry(i).mag == lit || kentry(i).mag == -lit)) return i;
		return -1;
	}
	template <int I>
	struct KAt {
		static_assert(I >= 0, "a constant of the segway's step that Segway::kentry does not list");
		static constexpr int value = I;
	};
	struct Consts { double v[kNK]; TanhConsts th; };
	template <unsigned PIN, int... I>
	__device__ __forceinline__ static void fillConsts(Consts &k, std::integer_sequence<int, I...>)
	{
		((k.v[I] = vk<(PIN & kentry(I).grp) != 0>(kentry(I).mag)), ...);
	}
	template <unsigned PIN = 0>
	__device__ __forceinline__ static Consts constants()
	{
		Consts k;
		fillConsts<PIN>(k, std::make_integer_sequence<int, kNK>());
		k.th = tanh_consts<(PIN & kPinTanh) != 0>();
		return k;
	}
#define K_(G, lit) ((lit) < 0 ? -k.v[KAt<kfind(G, lit)>::value] : k.v[KAt<kfind(G, lit)>::value])
	// :56-68  u = K (x + x_eq)
	__device__ static void backupController(const DevOptions &, const double (&x)[NX], double (&u)[NU], double (&Du)[NU * NX],
	                                        const Consts &k = constants<0>())
	{
#pragma clang fp contract(on)
		const double K0 = K_(kPinCtl, 44.7214), K1 = K_(kPinCtl, 44.6528), K2 = K_(kPinCtl, 150.1612), K3 = K_(kPinCtl, 37.6492);
		// The example adds x_eq = (0, 0, -0.138.., 0) to every state.  `0. + x` differs from x for x = -0 only, which
		// turns a product from -0 into +0; the third product is +0 whenever it is zero (a - a), and a sum that holds a +0
		// or a nonzero term does not depend on the signs of its other zeros: same u, bit for bit, without the three adds.
#if defined(__HIP_DEVICE_COMPILE__)
		u[0] = K0 * x[0] + K1 * x[1] + K2 * (K_(kPinCtl, -0.1383244254) + x[2]) + K3 * x[3];
#else
		u[0] = K0 * (0. + x[0]) + K1 * (0. + x[1]) + K2 * (K_(kPinCtl, -0.1383244254) + x[2]) + K3 * (0. + x[3]);
#endif
		Du[0] = K0; Du[1] = K1; Du[2] = K2; Du[3] = K3;
	}
	struct Trig { double s1, c1, s2, c2; };
	template <int POISON = kTrigChecked>
	__device__ static Trig trig(double pitch)
	{
		Trig t;
		sincos_fast<POISON>(pitch, t.s1, t.c1);
#if defined(__HIP_DEVICE_COMPILE__)
		// sin 2p = 2 sin p cos p, cos 2p = (cos p - sin p)(cos p + sin p): within 3e-16 (absolute) of the separate
		// sin / cos of 2p the example calls, for four multiplies instead of a second 35-instruction evaluation
		t.s2 = 2.0 * t.s1 * t.c1;
		t.c2 = (t.c1 - t.s1) * (t.c1 + t.s1);
#else
		sincos_fast<POISON>(2.0 * pitch, t.s2, t.c2);
#endif
		return t;
	}
	// :70-111.  The example's expressions are generated code with every term spelled out; here like terms are collected
	// (three w^2 sin terms with one 0.195 each -> one coefficient, ...), which leaves a third of the multiplications and
	// of the 64-bit literals (each literal is two scalar moves per Euler step: they do not fit the SGPR file).  The
	// collected coefficients are the example's own numbers multiplied out in double precision; values agree with the
	// spelled-out form to 1e-15 relative (checked on 2e5 random states), the oracle keeps the spelled-out form.
	struct Shared { double iden, rg, gc, gs; };
	// (fp contract(on) in the segway's functions: a multiply-add is fused where the SOURCE expression has one, never
	// across statements -- hipcc's default also fuses across statements, by use counts that differ between the kernels
	// these functions are inlined into; the two-role pass of k_tb.hip and the fused pass must give the same bits)
	__device__ static Shared dynamicsT(const double (&X)[NX], const Trig &t, double (&f)[NX], double (&g)[NX * NU],
	                                   const Consts &k = constants<0>())
	{
#pragma clang fp contract(on)
		const double w2 = X[3] * X[3];
		Shared h;
		h.iden = rcp_newton((K_(kPinDyn, 14.553176960783997) + K_(kPinDyn, -2.0831375273848773) * t.c2) + K_(kPinDyn, -0.59146430898882) * t.s2);
		f[0] = X[1];
		f[1] = (w2 * (K_(kPinDyn, -0.33003710190723146) * t.c1 + K_(kPinDyn, 2.3707272057666411) * t.s1) +
		        (K_(kPinDyn, 5.8022648711803244) * t.c2 + K_(kPinDyn, -20.435579143645651) * t.s2)) * h.iden;
		f[2] = X[3];
		f[3] = h.iden * ((K_(kPinDyn, -40.918271887954823) * t.c1 + K_(kPinDyn, 293.92471275850022) * t.s1) +
		                 w2 * (K_(kPinDyn, 0.59146430898882) * t.c2 + K_(kPinDyn, -2.0831375273848769) * t.s2));
		h.rg = rcp_newton(((K_(kPinRg, 8.3593271361634187) + K_(kPinRg, -2.1243074194638587) * (t.c1 * t.c1)) +
		                   K_(kPinRg, -0.04116989207898096) * (t.s1 * t.s1)) + K_(kPinRg, -0.29573215449441) * t.s2);
		gainT(t, h, g, k);
		return h;
	}
	// the input gain from sin / cos and the two reciprocals (h.iden, h.rg in; h.gc, h.gs out): dynamicsT's own lines, on
	// their own so that the Q role of the two-role pass (k_tb.hip) evaluates exactly them from the x role's record
	__device__ static void gainT(const Trig &t, Shared &h, double (&g)[NX * NU], const Consts &k = constants<0>())
	{
#pragma clang fp contract(on)
		g[0] = 0.0;
		h.gc = K_(kPinGain, 1.4575004011882324) * t.c1;
		h.gs = K_(kPinGain, 0.20290365220710288) * t.s1;
		g[1] = K_(kPinGain, 0.551244194154502) * ((K_(kPinGain, 4.1706936767483551) + h.gc) + h.gs) * h.rg;
		g[2] = 0.0;
		g[3] = K_(kPinGain, -5.65378660671284) * ((K_(kPinGain, 2.0043013906215941) + h.gc) + h.gs) * h.iden;
	}
	__device__ static void dynamics(const DevOptions &, const double (&x)[NX], double (&f)[NX], double (&g)[NX * NU])
	{
		dynamicsT(x, trig(x[2]), f, g);
	}
	// :113-212
	template <int POISON = kTrigChecked>
	__device__ static void dynamicsAndGradients(const DevOptions &o, const double (&x)[NX], double (&f)[NX],
	                                            double (&g)[NX * NU], double (&Df)[NX * NX], double (&Dg)[NX * NU * NX])
	{
		gradientsWithTrig(o, x, trig<POISON>(x[2]), f, g, Df, Dg);
	}
	// the pitch's sin / cos carried along the Euler steps (kTrigCarried; the TB kernel re-synchronises at block starts)
	static constexpr bool kTrigCarry = true;
	static constexpr int kTrigAngle = 2;
	__device__ static void dynamicsAndGradientsCarried(const DevOptions &o, const double (&x)[NX], double (&f)[NX],
	                                                   double (&g)[NX * NU], double (&Df)[NX * NX],
	                                                   double (&Dg)[NX * NU * NX], TrigCarry &cy, bool reset)
	{
		if (reset) {
			sincos_fast<kTrigUnchecked>(x[2], cy.s, cy.c);
			cy.x = x[2];
		} else {
			sincos_carry<false>(x[2], cy);
		}
		Trig t;
		t.s1 = cy.s;
		t.c1 = cy.c;
		t.s2 = 2.0 * t.s1 * t.c1;
		t.c2 = (t.c1 - t.s1) * (t.c1 + t.s1);
		gradientsWithTrig(o, x, t, f, g, Df, Dg);
	}
	// |pitch rate| <= sqrt(xb(3)^2 - hmin): the pitch's increment per step is dt times that
	__device__ static bool trigCarryBounded(const DevOptions &o, double hmin)
	{
		return trigArgsBounded(hmin) && (xb(3) * xb(3) - hmin) * (o.trajDt * o.trajDt) <= kTrigCarryMaxStep * kTrigCarryMaxStep;
	}
	__device__ static void gradientsWithTrig(const DevOptions &, const double (&x)[NX], const Trig &t, double (&f)[NX],
	                                         double (&g)[NX * NU], double (&Df)[NX * NX], double (&Dg)[NX * NU * NX])
	{
		const Shared h = dynamicsT(x, t, f, g);
		gradientsGiven(x, t, h, Df, Dg);
	}
	// the gradients from sin / cos, dynamicsT's shared terms and the two states they depend on (x[1], x[3])
	static constexpr int kGradStates[2] = {1, 3};
	// TANHV: `k` holds the constants of tanh's polynomial in vector registers (kPinTanh): three-address Horner steps
	template <bool TANHV = false>
	__device__ static void gradientsGiven(const double (&x)[NX], const Trig &t, const Shared &h, double (&Df)[NX * NX],
	                                      double (&Dg)[NX * NU * NX], const Consts &k = constants<0>())
	{
#pragma clang fp contract(on)
		const double c1 = t.c1, s1 = t.s1, c2 = t.c2, s2 = t.s2;
		const double w2 = x[3] * x[3];
		const double th = tanh_abs_accurate<TANHV>(x[1] * 1000.0, k.th);
		const double t25 = th * K_(kPinDfA, 15.13175750513302) - K_(kPinDfA, 40.918271887954823);
		const double t26 = w2 * K_(kPinDfA, 3.3849959169972448) + th * K_(kPinDfA, 30.26351501026604);
		// the example's t23 = 1 / (2.083.. c2 + 0.591.. s2 - 14.553..) and its d26 are the negatives of the two
		// denominators above (same numbers, the last printed digit apart): one reciprocal each serves both
		const double t23 = -h.iden, r26 = -h.rg;
#pragma unroll
		for (int i = 0; i < NX * NX; i++) Df[i] = 0.0;
		Df[4] = 1.0;
		// (th^2 K - K) terms as K (th^2 - 1): fewer operations and no cancellation between two rounded products
		const double q = th * th - 1.0;
		Df[5] = -t23 * q * ((K_(kPinDfA, 8443.5211353581435) + K_(kPinDfA, 410.77609832706019) * s1) + K_(kPinDfA, 2950.692713500939) * c1);
		Df[7] = t23 * q * ((K_(kPinDfA, 20808.641003022261) + K_(kPinDfA, 2106.5440939849238) * s1) + K_(kPinDfA, 15131.75750513302) * c1);
		const double cth = c1 * th;
		const double sth = s1 * th;
		const double v = c2 * K_(kPinDfB, 1.18292861797764) - s2 * K_(kPinDfB, 4.1662750547697547);
		const double e3 = v * (t23 * t23);
		Df[9] = t23 * ((((c2 * K_(kPinDfB, 40.8711582872913) + s2 * K_(kPinDfB, 11.604529742360651)) - c1 * w2 * K_(kPinDfB, 2.3707272057666411)) +
		                cth * K_(kPinDfB, 0.41077609832706019)) - s1 * t26 * K_(kPinDfB, 0.0975)) -
		        e3 * (((((c2 * K_(kPinDfB, -5.8022648711803244) + s2 * K_(kPinDfB, 20.435579143645651)) + th * K_(kPinDfB, 8.443521135358143)) -
		                s1 * w2 * K_(kPinDfB, 2.3707272057666411)) + sth * K_(kPinDfB, 0.41077609832706019)) + c1 * t26 * K_(kPinDfB, 0.0975));
		const double wc = w2 * c2;
		const double ws = w2 * s2;
		Df[11] = t23 * ((((c1 * K_(kPinDfB, -293.92471275850022) - cth * K_(kPinDfB, 2.1065440939849238)) + wc * K_(kPinDfB, 4.1662750547697547)) +
		                 ws * K_(kPinDfB, 1.18292861797764)) + s1 * t25) +
		         e3 * (((((s1 * K_(kPinDfB, 293.92471275850022) + th * K_(kPinDfB, 20.808641003022259)) + wc * K_(kPinDfB, 0.59146430898881985)) +
		                 sth * K_(kPinDfB, 2.1065440939849238)) - ws * K_(kPinDfB, 2.0831375273848769)) + c1 * t25);
		Df[13] = t23 * x[3] * (c1 * K_(kPinDfC, 0.6600742038144628) - s1 * K_(kPinDfC, 4.7414544115332831));
		Df[14] = 1.0;
		Df[15] = -t23 * x[3] * v;
#pragma unroll
		for (int i = 0; i < NX * NU * NX; i++) Dg[i] = 0.0;
		// (c1 s1 = s2 / 2)
		Dg[9] = -(c1 * K_(kPinDfC, 0.1118494602519098) - s1 * K_(kPinDfC, 0.80343863413287053)) * r26 +
		        (r26 * r26) * (c2 * K_(kPinDfC, 0.59146430898882) - s2 * K_(kPinDfC, 2.0831375273848773)) *
		            ((c1 * K_(kPinDfC, 0.80343863413287053) + s1 * K_(kPinDfC, 0.1118494602519098)) + K_(kPinDfC, 2.2990706749044238));
		Dg[11] = (c1 * K_(kPinDfC, 1.1471739513016379) - s1 * K_(kPinDfC, 8.24039624751662)) * t23 -
		         e3 * ((c1 * K_(kPinDfC, 8.24039624751662) + s1 * K_(kPinDfC, 1.1471739513016379)) + K_(kPinDfC, 11.33189235811229));
	}
#undef K_
};

// ---------------------------------------------------------------------------------------------
// Robust inverted pendulum, examples/InvertedPendulum_Robust.cpp:20-79: half-plane safety set
// 1 - a.x >= 0 (data in DevOptions: the shipped SafetySetData vector is empty, :51) and dynamics in
// affine arithmetic with the input gain uncertain in [pMin, pMax].  Functor methods that take affine
// forms are declared in k_robust.hip next to their only user.
struct InvertedPendulumRobust {
	static constexpr int NX = 2, NU = 1, MAXNP = ASIF_HIP_MAX_HALFPLANES;

	// :53-61
	__device__ static void safetySet(const DevOptions &o, const double (&x)[NX], double (&h)[MAXNP],
	                                 double (&Dh)[MAXNP * NX])
	{
		const int N = o.nHalfPlanes;
		for (int i = 0; i < N; i++) {
			h[i] = 1. - o.halfPlanes[2 * i] * x[0] - o.halfPlanes[2 * i + 1] * x[1];
			Dh[i] = -o.halfPlanes[2 * i];
			Dh[i + N] = -o.halfPlanes[2 * i + 1];
		}
	}
};

} // namespace asif
