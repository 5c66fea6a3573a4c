// gi_small.hpp -- dual active-set solve (Goldfarb & Idnani 1983) of the tiny strictly convex QPs of the
// safety filter, entirely in registers:  nv <= 3 variables, diagonal cost, so every working-set system is
// a closed-form solve of size <= 3.
//
// Where it sits: it is the "finish first" stage of AdmmSmall::solve (asif_hip_solver::polish == 2): one
// attempt before the first ADMM iteration.  The ADMM iterations (admm_small.hpp) remain the general
// method and take over for whatever this stage leaves undecided (cost without curvature, step budget,
// numerically ambiguous dependence).  The reference sends the same problems through OSQP
// (src/qpwrapper_osqp.cpp:217-239); form at the boundary (include/qpwrapper_abstract.h:11-15):
//     min x'Hx + c'x   s.t.  A x >= b (== b where be),  lb <= x <= ub,   H diagonal, H > 0 here.
//
// Method.  Start at the unconstrained minimiser with an empty working set W (<= nv constraints, kept as
// nv "slots" replicated in every lane of the lane group).  Repeat: pick the most violated constraint p
// (equalities first); with N the normals in W and G = 2H,
//     r = (N'G^-1 N)^-1 N'G^-1 n_p   (dual step direction),   z = G^-1 (n_p - N r)   (primal direction);
// take the longest step that keeps the multipliers of W non-negative (t1) up to the step that makes p
// active (t2); a blocking multiplier leaves W, a full step puts p into W.  If n_p depends linearly on W and no
// multiplier blocks, the constraints of W + p are inconsistent: the QP is infeasible -- the verdict needs no
// phase 1 and no separate certificate.  Finite for strictly convex problems; a step budget guards the
// floating-point corner cases (result "undecided").
//
// Lane mapping as in admm_small.hpp: G lanes per QP, the NC general rows dealt round-robin (local row k of
// lane g is row g + k G), variables / bounds / working set replicated.  Cross-lane traffic per step: one
// max + one min (selection) and nv + 2 sums (the selected row), DPP butterflies.
//
// The file compiles for the host as well (G = 1 only): tests/host_gi_driver.cpp runs exactly this code on
// the CPU against the oracle's exact enumeration.  That is a test harness, not a fallback: nothing in
// libasif_hip.so calls it on the host.
#pragma once
#include "qp_lane.hpp"

namespace asif {

constexpr int kGiOptimal = 1;
constexpr int kGiInfeasible = 2;
constexpr int kGiUndecided = 0;
constexpr int kGiFailed = 3; // non-finite problem data: the reference's solver runs such a problem to max_iter

template <int NV, int RPL, int G>
struct GiSmall {
	static_assert(NV >= 1 && NV <= 3, "closed-form working-set solves: nv <= 3");
	static constexpr double kViolTol = 1e-12;  // a row counts as violated beyond this, relative to its own terms
	static constexpr double kNoise = 4e-16;    // rounding level of one component of n_p - N r, relative to its terms
	static constexpr double kActiveTol = 1e-9; // final check: working rows met to this relative accuracy
	static constexpr int kIdLb = 1 << 16, kIdUb = 1 << 17;
	static constexpr double kEqScore = 1e300;

	// g: this lane's index in its group.  x: the optimum (when kGiOptimal is returned).  steps: iterations used.
	//
	// A variable pinned by its bounds (lb == ub) in EVERY QP of the wave is eliminated first -- the explicit class pins
	// its relaxation variable (src/asif.cpp:88-91), which turns its 2-variable problem into a 1-variable one: with a
	// diagonal cost the pinned coordinate decouples, its column moves to the right-hand side (b - a_j x_j), and the same
	// method runs one dimension lower (a third of the instructions for nv = 2 -> 1).  Wave-uniform by construction
	// (wave_all), so no lane diverges; a wave with mixed pins takes the general path below, where a pinned variable
	// is pre-loaded into the working set instead.
	// the elimination itself, for a variable J the caller knows to be pinned in every QP of the wave
	template <int J>
	ASIF_HD static int solve_with_pinned(const QpLaneData<NV, RPL> &in, int g, int max_steps, double (&x)[NV], int &steps)
	{
		const bool nonfinite = qp_data_nonfinite<NV, RPL, G>(in.Hd, in.c, in.lb, in.ub, in.A, in.b);
		const int verdict = pinned_unchecked<J>(in, g, max_steps, x, steps);
		return nonfinite ? kGiFailed : verdict;
	}
	template <int J>
	ASIF_HD static int pinned_unchecked(const QpLaneData<NV, RPL> &in, int g, int max_steps, double (&x)[NV], int &steps)
	{
		static_assert(NV >= 2 && J >= 0 && J < NV, "a variable to eliminate and one to keep");
		QpLaneData<NV - 1, RPL> red;
		const double pin = in.lb[J];
#pragma unroll
		for (int k = 0; k < RPL; k++) {
#pragma unroll
			for (int j = 0; j < NV - 1; j++) red.A[k][j] = in.A[k][j < J ? j : j + 1];
			red.b[k] = in.b[k] - in.A[k][J] * pin;
			red.eq[k] = in.eq[k];
		}
#pragma unroll
		for (int j = 0; j < NV - 1; j++) {
			const int jj = j < J ? j : j + 1;
			red.Hd[j] = in.Hd[jj];
			red.c[j] = in.c[jj];
			red.lb[j] = in.lb[jj];
			red.ub[j] = in.ub[jj];
		}
		// a column entry of 1e308 times the pin overflows: the reduced right-hand side is no longer data the comparisons
		// below can judge (inf > inf is false) -- failed, like non-finite data handed in
		double ovf = 0.0;
#pragma unroll
		for (int k = 0; k < RPL; k++) ovf = fma(red.b[k], 1e160, ovf);
		const bool overflow = gor<G>(!(fabs(ovf) < __builtin_huge_val()) ? 1 : 0) != 0;
		double xr[NV - 1];
		const int verdict = GiSmall<NV - 1, RPL, G>::solve_unchecked(red, g, max_steps, xr, steps);
#pragma unroll
		for (int j = 0; j < NV - 1; j++) x[j < J ? j : j + 1] = xr[j];
		x[J] = pin;
		return overflow ? kGiFailed : verdict;
	}

	// Entry.  Non-finite data is asked for once, on the whole problem, and overrides whatever the comparisons below made
	// of it (they read a NaN row as met); selects, not an early return: the method holds group and wave reductions.
	ASIF_HD static int solve(const QpLaneData<NV, RPL> &in, int g, int max_steps, double (&x)[NV], int &steps)
	{
		const bool nonfinite = qp_data_nonfinite<NV, RPL, G>(in.Hd, in.c, in.lb, in.ub, in.A, in.b);
		const int verdict = solve_unchecked(in, g, max_steps, x, steps);
		return nonfinite ? kGiFailed : verdict;
	}
	ASIF_HD static int solve_unchecked(const QpLaneData<NV, RPL> &in, int g, int max_steps, double (&x)[NV], int &steps)
	{
		if constexpr (NV >= 2) {
			bool pinned[NV];
#pragma unroll
			for (int j = 0; j < NV; j++) pinned[j] = wave_all((in.lb[j] == in.ub[j]) & (fabs(in.lb[j]) < 1e300));
			int verdict = -1;
			unrolled_until<NV>([&](auto jc) {
				constexpr int J = NV - 1 - decltype(jc)::value; // last variable first: the relaxation variables sit at the end
				if (!pinned[J]) return false;
				verdict = pinned_unchecked<J>(in, g, max_steps, x, steps);
				return true;
			});
			if (verdict >= 0) return verdict;
		}
		return solve_general(in, g, max_steps, x, steps);
	}

	// One variable (where the elimination above leaves the explicit class's QP): the working set holds one constraint,
	// every step of the method is a move to the nearest end of the feasible interval, and the whole iteration collapses
	// into its fixed point -- clip the unconstrained minimiser to the intersection [lo, hi] of the half-lines the rows
	// and bounds define.  Verdicts as the general path gives them: a row without the variable (a == 0 exactly; the
	// general path's dependence test on an empty working set is exact too) or a pair of rows that exclude each other
	// is a conflict only beyond kActiveTol relative to the row's own terms, checked on the residuals at the clipped
	// point (a conflict below that is a degenerate vertex and counts as met).
	ASIF_HD static int solve_1d(const QpLaneData<NV, RPL> &in, int g, double (&x)[NV], int &steps)
	{
		const double P = 2.0 * in.Hd[0];
		steps = 0;
		if (!wave_all(P > 0.0)) return kGiUndecided;
		const double xu = -in.c[0] / (P > 0.0 ? P : 1.0);
		double lo = in.lb[0], hi = in.ub[0];
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			const double a = in.A[k][0];
			const double q = in.b[k] / (a != 0.0 ? a : 1.0);
			lo = ((a > 0.0) | (in.eq[k] & (a < 0.0))) ? fmax(lo, q) : lo;
			hi = ((a < 0.0) | (in.eq[k] & (a > 0.0))) ? fmin(hi, q) : hi;
		}
		lo = gmax<G>(lo);
		hi = gmin<G>(hi);
		const double xs = fmin(fmax(xu, lo), hi);
		int bad = 0;
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			const double t = in.A[k][0] * xs;
			const double res = in.b[k] - t;
			const double v = in.eq[k] ? fabs(res) : res;
			bad |= (v > kActiveTol * (1.0 + fabs(in.b[k]) + fabs(t))) ? 1 : 0;
		}
		bad |= (in.lb[0] - xs > kActiveTol * (1.0 + fabs(xs) + fabs(in.lb[0]))) ? 1 : 0;
		bad |= (xs - in.ub[0] > kActiveTol * (1.0 + fabs(xs) + fabs(in.ub[0]))) ? 1 : 0;
		bad = gor<G>(bad);
		x[0] = xs;
		steps = 1;
		return bad ? kGiInfeasible : kGiOptimal;
	}

	// r with sum_s r_s n_s = c for NV normals n_s (rows of sn): Cramer's rule.  A determinant is a sum of products; its
	// rounding is kNoise times the sum of their magnitudes.  False ("ambiguous") when the main determinant, or a
	// numerator that is not plainly zero, is known to fewer than three digits.
	ASIF_HD static bool solve_square(const double (&sn)[NV][NV], const double (&c)[NV], double (&r)[NV])
	{
		bool ok = true;
		if constexpr (NV == 1) {
			r[0] = c[0] / (sn[0][0] != 0.0 ? sn[0][0] : 1.0);
			ok = sn[0][0] != 0.0;
		} else if constexpr (NV == 2) {
			// columns of the system are the normals: [n_0 n_1] r = c
			auto det2 = [](double a, double b, double cc, double d, double &mag) {
				const double p = a * d, q = b * cc;
				mag = fabs(p) + fabs(q);
				return p - q;
			};
			double m, m0, m1;
			const double det = det2(sn[0][0], sn[1][0], sn[0][1], sn[1][1], m);
			const double d0 = det2(c[0], sn[1][0], c[1], sn[1][1], m0);
			const double d1 = det2(sn[0][0], c[0], sn[0][1], c[1], m1);
			ok = fabs(det) > 1e4 * kNoise * m;
			const double inv = 1.0 / (ok ? det : 1.0);
			const double num[2] = {d0, d1}, mg[2] = {m0, m1};
#pragma unroll
			for (int s = 0; s < 2; s++) {
				const bool zero = !(fabs(num[s]) > 16.0 * kNoise * mg[s]);
				ok = ok & (zero | (fabs(num[s]) > 1e4 * kNoise * mg[s]));
				r[s] = zero ? 0.0 : num[s] * inv;
			}
		} else {
			// 3 x 3 by cofactors along the replaced column; a[i][s] = component i of normal s
			auto det3 = [](const double (&a)[3][3], double &mag) {
				double d = 0.0;
				mag = 0.0;
#pragma unroll
				for (int k = 0; k < 3; k++) {
					const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
					const double p = a[0][k] * a[1][k1] * a[2][k2], q = a[0][k] * a[1][k2] * a[2][k1];
					d += p - q;
					mag += fabs(p) + fabs(q);
				}
				return d;
			};
			double a[3][3], m;
#pragma unroll
			for (int i = 0; i < 3; i++)
#pragma unroll
				for (int t = 0; t < 3; t++) a[i][t] = sn[t][i];
			const double det = det3(a, m);
			ok = fabs(det) > 1e4 * kNoise * m;
			const double inv = 1.0 / (ok ? det : 1.0);
#pragma unroll
			for (int s = 0; s < 3; s++) {
				double b[3][3], ms;
#pragma unroll
				for (int i = 0; i < 3; i++)
#pragma unroll
					for (int t = 0; t < 3; t++) b[i][t] = t == s ? c[i] : a[i][t];
				const double ds = det3(b, ms);
				const bool zero = !(fabs(ds) > 16.0 * kNoise * ms);
				ok = ok & (zero | (fabs(ds) > 1e4 * kNoise * ms));
				r[s] = zero ? 0.0 : ds * inv;
			}
		}
		return ok;
	}

	ASIF_HD static int solve_general(const QpLaneData<NV, RPL> &in, int g, int max_steps, double (&x)[NV], int &steps)
	{
		if constexpr (NV == 1) return solve_1d(in, g, x, steps);
		double Pinv[NV];
		bool convex = true;
#pragma unroll
		for (int j = 0; j < NV; j++) {
			const double P = 2.0 * in.Hd[j];
			convex = convex && (P > 0.0);
			Pinv[j] = 1.0 / (convex ? P : 1.0);
			x[j] = -in.c[j] * Pinv[j];
		}
		steps = 0;
		if (!wave_all(convex)) return kGiUndecided; // wave-uniform exit: the loop below holds group reductions

		// row ranking weights: 1 / |a_r| in the metric of the cost (any positive weights are correct; these make
		// "most violated" mean "farthest away")
		double w[RPL], wb[NV];
#pragma unroll
		for (int k = 0; k < RPL; k++) {
			double s = 0.0;
#pragma unroll
			for (int j = 0; j < NV; j++) s += in.A[k][j] * in.A[k][j] * Pinv[j];
			w[k] = s > 0.0 ? rank_rsqrt(s) : 1.0;
		}
#pragma unroll
		for (int j = 0; j < NV; j++) wb[j] = rank_rsqrt(Pinv[j]);

		// working set: slot s holds constraint  sn[s].x >= sb[s]  with multiplier su[s]; sid < 0: empty.
		// ids: general row r -> r;  lower bound of variable j -> kIdLb + j;  upper bound -> kIdUb + j.
		double sn[NV][NV], sb[NV], su[NV];
		int sid[NV];
		bool seq[NV];
#pragma unroll
		for (int s = 0; s < NV; s++) {
			sid[s] = -1;
			seq[s] = false;
			sb[s] = 0.0;
			su[s] = 0.0;
#pragma unroll
			for (int j = 0; j < NV; j++) sn[s][j] = 0.0;
		}
		bool rin[RPL]; // local row is in W
		int bin[NV];   // bound of variable j in W: 0 no, -1 lower, +1 upper, 2 pinned (equality)
		int bskip[NV]; // bit 0 / 1: lower / upper bound of variable j found met to rounding at a degenerate vertex
#pragma unroll
		for (int k = 0; k < RPL; k++) rin[k] = false;
#pragma unroll
		for (int j = 0; j < NV; j++) {
			bin[j] = 0;
			bskip[j] = 0;
		}

		// A pinned variable (lb == ub: the explicit class pins its relaxation variable, src/asif.cpp:88-91) is an
		// equality the method would add first anyway; with a diagonal cost that step is known in closed form:
		// x_j = lb_j, the other coordinates of the minimiser do not move, multiplier = d cost / d x_j there.
		{
			int used = 0;
#pragma unroll
			for (int j = 0; j < NV; j++) {
				if (in.lb[j] == in.ub[j]) {
#pragma unroll
					for (int s = 0; s < NV; s++)
						if (s == used) {
							sid[s] = kIdLb + j;
							seq[s] = true;
							sb[s] = in.lb[j];
							su[s] = 2.0 * in.Hd[j] * in.lb[j] + in.c[j];
#pragma unroll
							for (int k = 0; k < NV; k++) sn[s][k] = k == j ? 1.0 : 0.0;
						}
					used++;
					x[j] = in.lb[j];
					bin[j] = 2;
				}
			}
		}

		// candidate p:  cn.x >= cb, multiplier cu so far
		double cn[NV], cb = 0.0, cu = 0.0;
		int cid = -1;
		bool ceq = false, have = false;
#pragma unroll
		for (int j = 0; j < NV; j++) cn[j] = 0.0;

		int verdict = kGiUndecided;
		bool done = false;
#pragma unroll 1
		for (int it = 0; it < max_steps; it++) {
			if (wave_all(done)) break;
			if (!done) steps++;
			// ---- 1. selection: the most violated constraint outside W (used by lanes without a pending candidate).
			// Everything below is written as selects: the lanes of a wave sit in different phases of the method, and
			// exec-mask branches around a handful of instructions cost more than the instructions.
			double best = 0.0;
			int bid = 0x7fffffff;
			// candidates are visited in increasing id, so "strictly better" keeps the lowest id among equals.
			// (bitwise & | on bools on purpose: the short-circuit forms compile to exec-mask branches)
			auto consider = [&](bool viol, double score, int id) {
				const bool take = viol & (score > best);
				best = take ? score : best;
				bid = take ? id : bid;
			};
#pragma unroll
			for (int k = 0; k < RPL; k++) {
				double ax = 0.0, mag = fabs(in.b[k]);
#pragma unroll
				for (int j = 0; j < NV; j++) {
					const double t = in.A[k][j] * x[j];
					ax += t;
					mag += fabs(t);
				}
				const double res = in.b[k] - ax; // > 0: violated from below
				const double v = in.eq[k] ? fabs(res) : res;
				consider(!rin[k] & (v > kViolTol * (1.0 + mag)), in.eq[k] ? kEqScore : v * w[k], g + k * G);
			}
#pragma unroll
			for (int j = 0; j < NV; j++) {
				const double lo = in.lb[j] - x[j], hi = x[j] - in.ub[j];
				const double tlo = kViolTol * (1.0 + fabs(x[j]) + fabs(in.lb[j])), thi = kViolTol * (1.0 + fabs(x[j]) + fabs(in.ub[j]));
				const bool pin = in.lb[j] == in.ub[j]; // pinned variable: one equality
				const bool lo_free = ((bskip[j] & 1) == 0) & (pin ? bin[j] == 0 : bin[j] != -1);
				consider(lo_free & ((pin ? fabs(lo) : lo) > tlo), pin ? kEqScore : lo * wb[j], kIdLb + j);
				consider(!pin & (bin[j] != 1) & ((bskip[j] & 2) == 0) & (hi > thi), hi * wb[j], kIdUb + j);
			}
			const double gbest = gmax<G>(best);
			const int gid = gmin_int<G>(best == gbest ? bid : 0x7fffffff);
			const bool choosing = !have & !done;
			const bool fresh = choosing & (gbest > 0.0);
			verdict = (choosing & !fresh) ? kGiOptimal : verdict;
			done = done | (choosing & !fresh);
			cid = fresh ? gid : cid;
			cu = fresh ? 0.0 : cu;
			have = have | fresh;
			if (wave_all(done)) break; // the selection found nothing anywhere: no step to take
			{ // the candidate's data; a general row comes from its owning lane through a sum whose other terms are zero
				double rn[NV], rb = 0.0;
				int req = 0;
#pragma unroll
				for (int j = 0; j < NV; j++) rn[j] = 0.0;
#pragma unroll
				for (int k = 0; k < RPL; k++) {
					const bool mine = (g + k * G) == cid;
#pragma unroll
					for (int j = 0; j < NV; j++) rn[j] = mine ? in.A[k][j] : rn[j];
					rb = mine ? in.b[k] : rb;
					req |= (mine & in.eq[k]) ? 1 : 0;
				}
#pragma unroll
				for (int j = 0; j < NV; j++) rn[j] = gsum<G>(rn[j]);
				rb = gsum<G>(rb);
				req = gor<G>(req);
				double fn[NV], fb = rb, ax = 0.0;
				bool feq = req != 0;
#pragma unroll
				for (int j = 0; j < NV; j++) {
					const bool isl = cid == kIdLb + j, isu = cid == kIdUb + j;
					fn[j] = isl ? 1.0 : (isu ? -1.0 : rn[j]);
					fb = isl ? in.lb[j] : (isu ? -in.ub[j] : fb);
					feq = feq | (isl & (in.lb[j] == in.ub[j]));
					ax += fn[j] * x[j];
				}
				const bool flip = feq & (ax > fb); // orient an equality so that it reads n.x >= b and is violated
#pragma unroll
				for (int j = 0; j < NV; j++) cn[j] = fresh ? (flip ? -fn[j] : fn[j]) : cn[j];
				cb = fresh ? (flip ? -fb : fb) : cb;
				ceq = fresh ? feq : ceq;
			}
			// ---- 2. step directions:  M r = d,  z = G^-1 (n_p - N r)
			double M[NV][NV], Mi[NV], Mdiag[NV], r[NV], z[NV];
			bool full = true;
#pragma unroll
			for (int s = 0; s < NV; s++) {
				const bool vs = sid[s] >= 0;
				full = full & vs;
				double d = 0.0;
#pragma unroll
				for (int j = 0; j < NV; j++) d += sn[s][j] * cn[j] * Pinv[j];
				r[s] = vs ? d : 0.0;
#pragma unroll
				for (int t = 0; t <= s; t++) {
					double m = 0.0;
#pragma unroll
					for (int j = 0; j < NV; j++) m += sn[s][j] * sn[t][j] * Pinv[j];
					M[s][t] = (vs & (sid[t] >= 0)) ? m : (s == t ? 1.0 : 0.0);
				}
				Mdiag[s] = M[s][s];
			}
			const bool spd = ldl_factor<NV>(M, Mi);
			ldl_solve<NV>(M, Mi, r);
			double zn = 0.0, nn = 0.0, cond = 1.0;
#pragma unroll
			for (int s = 0; s < NV; s++) cond = fmax(cond, Mdiag[s] * Mi[s]); // 1 / sin^2 of the sharpest angle inside W
			// z component by component, each judged against the rounding of its own cancellation: a component at that
			// level is zero (n_p lies in the span of W there), one clearly above it is data -- however small: a row
			// [Lgh, h] with h = 1e-9 is NOT parallel to a bound on u, the relaxation variable just has to travel far
			bool dependent = true, ambiguous = !spd;
#pragma unroll
			for (int j = 0; j < NV; j++) {
				double t = cn[j], mag = 0.0;
#pragma unroll
				for (int s = 0; s < NV; s++) {
					t -= sn[s][j] * r[s];
					mag += fabs(sn[s][j] * r[s]);
				}
				const double noise = kNoise * (fabs(cn[j]) + cond * mag);
				const bool zero = full | !(fabs(t) > 16.0 * noise); // nv normals in W span everything
				ambiguous = ambiguous | (!zero & !(fabs(t) > 1e4 * noise)); // known to < 3 digits: give up
				t = zero ? 0.0 : t;
				dependent = dependent & zero;
				z[j] = t * Pinv[j];
				zn += t * z[j];
				nn += cn[j] * cn[j] * Pinv[j];
			}
			// A lane the general form leaves ambiguous gets a second opinion -- computed only by a wave that holds such a
			// lane -- from the two forms that do not square the condition number of W:
			//  * W full (nv normals): n_p = N r is a square system; Cramer's rule on N itself, every determinant judged
			//    against the rounding of its own terms.  Two rows [eps, h], [0, 1] with eps / h = 1e-12 (affine-arithmetic
			//    noise in Lgh: the shipped half-planes of DoubleIntegrator_Robust) leave a 2 x 2 matrix M = N'G^-1 N whose
			//    second pivot is pure rounding, where N still gives r to four digits.
			//  * W one normal short of full: the part of n_p outside span(W) lies along G w, w the normal of that span
			//    (2-D: the perpendicular of the one normal; 3-D: the cross product of the two),
			//    t = G w (n_p.w) / (w.G w): ONE determinant n_p.w carries the whole cancellation instead of 1 - 0.99999...
			//    per component when n_p is nearly parallel to a working row (a row [1e-9, h] against the bound on delta).
			if constexpr (NV >= 2) {
				const bool robust = ambiguous & have & !done;
				if (wave_any(robust)) {
					int nW = 0;
#pragma unroll
					for (int s = 0; s < NV; s++) nW += sid[s] >= 0 ? 1 : 0;
					double rd[NV];
					const bool sq_ok = solve_square(sn, cn, rd);
					const bool sq = robust & full, codim1 = robust & (nW == NV - 1);
					double w[NV], wmag[NV];
					if constexpr (NV == 2) { // empty slots hold zero normals: the sum IS the one normal in W
						w[0] = sn[0][1] + sn[1][1];
						w[1] = -(sn[0][0] + sn[1][0]);
						wmag[0] = wmag[1] = 0.0;
					} else {
#pragma unroll
						for (int k = 0; k < 3; k++) {
							const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
							double v = 0.0, m = 0.0;
#pragma unroll
							for (int a = 0; a < 3; a++)
#pragma unroll
								for (int b = a + 1; b < 3; b++) { // the pair without the empty slot is the only non-zero term
									const double p = sn[a][k1] * sn[b][k2], q = sn[a][k2] * sn[b][k1];
									v += p - q;
									m += fabs(p) + fabs(q);
								}
							w[k] = v;
							wmag[k] = m;
						}
					}
					double cw = 0.0, cwmag = 0.0, den = 0.0;
#pragma unroll
					for (int k = 0; k < NV; k++) {
						cw += cn[k] * w[k];
						cwmag += fabs(cn[k] * w[k]) + fabs(cn[k]) * wmag[k];
						den += 2.0 * in.Hd[k] * w[k] * w[k];
					}
					const bool cw_zero = !(fabs(cw) > 16.0 * kNoise * cwmag);
					const bool cw_amb = (!cw_zero & !(fabs(cw) > 1e4 * kNoise * cwmag)) | !(den > 0.0);
					const double f = cw / (den > 0.0 ? den : 1.0);
					// full: r from the square system, z = 0 by construction (spd and cond play no part on this path)
#pragma unroll
					for (int s = 0; s < NV; s++) r[s] = sq ? rd[s] : r[s];
					ambiguous = sq ? !sq_ok : (codim1 ? (!spd | cw_amb) : ambiguous);
					bool dep1 = true;
					double zn1 = 0.0;
					double z1[NV];
#pragma unroll
					for (int k = 0; k < NV; k++) {
						const double t = (cw_zero | sq) ? 0.0 : 2.0 * in.Hd[k] * w[k] * f;
						dep1 = dep1 & (t == 0.0);
						z1[k] = t * Pinv[k];
						zn1 += t * z1[k];
					}
					const bool over = sq | codim1;
					dependent = over ? dep1 : dependent;
					zn = over ? zn1 : zn;
#pragma unroll
					for (int k = 0; k < NV; k++) z[k] = over ? z1[k] : z[k];
				}
			}
			// ---- 3. dual blocking: smallest su / r over the droppable slots with r > 0
			double t1 = 1e300;
			int kdrop = -1;
#pragma unroll
			for (int s = 0; s < NV; s++) {
				const bool cand = (sid[s] >= 0) & !seq[s] & (r[s] > 0.0) & (r[s] * r[s] * Mdiag[s] > 1e-24 * nn);
				const double ratio = su[s] / (cand ? r[s] : 1.0);
				const bool take = cand & (ratio < t1);
				t1 = take ? ratio : t1;
				kdrop = take ? s : kdrop;
			}
			// ---- 4. the step
			{
				const bool active = have & !done;
				double res = cb, rmag = fabs(cb);
#pragma unroll
				for (int j = 0; j < NV; j++) {
					res -= cn[j] * x[j];
					rmag += fabs(cn[j] * x[j]);
				}
				const bool amb = active & ambiguous; // verdict stays undecided
				const bool dep = active & !ambiguous & dependent;
				const bool indep = active & !ambiguous & !dependent;
				const bool stuck = dep & (kdrop < 0); // n_p in span(W), no multiplier blocks
				// W + p inconsistent -> infeasible; but when W pins the point and p misses it by rounding only (a
				// degenerate vertex) p counts as met
				const bool infeasible = stuck & (res > kActiveTol * (1.0 + rmag));
				const bool skip = stuck & !infeasible;
				const double t2 = res / (indep ? zn : 1.0);
				const bool add = indep & (t2 <= t1);
				const bool drop = (dep & (kdrop >= 0)) | (indep & !add);
				const double t = add ? t2 : (drop ? t1 : 0.0);
				verdict = infeasible ? kGiInfeasible : verdict;
				done = done | amb | infeasible;
#pragma unroll
				for (int j = 0; j < NV; j++) x[j] = indep ? x[j] + t * z[j] : x[j];
				cu = (drop | add) ? cu + t : cu;
				const int cid_now = cid;
#pragma unroll
				for (int s = 0; s < NV; s++) {
					const bool vs = sid[s] >= 0;
					double u = (vs & (drop | add)) ? su[s] - t * r[s] : su[s];
					u = (vs & !seq[s]) ? fmax(u, 0.0) : u;
					su[s] = u;
				}
				// leave: slot kdrop
				int gone = -1;
#pragma unroll
				for (int s = 0; s < NV; s++) {
					const bool out = drop & (s == kdrop);
					gone = out ? sid[s] : gone;
					sid[s] = out ? -1 : sid[s];
					su[s] = out ? 0.0 : su[s];
#pragma unroll
					for (int j = 0; j < NV; j++) sn[s][j] = out ? 0.0 : sn[s][j];
				}
				// join: first empty slot
				bool placed = !add;
#pragma unroll
				for (int s = 0; s < NV; s++) {
					const bool in_ = !placed & (sid[s] < 0);
					placed = placed | in_;
					sid[s] = in_ ? cid_now : sid[s];
					seq[s] = in_ ? ceq : seq[s];
					sb[s] = in_ ? cb : sb[s];
					su[s] = in_ ? cu : su[s];
#pragma unroll
					for (int j = 0; j < NV; j++) sn[s][j] = in_ ? cn[j] : sn[s][j];
				}
				const bool settled = add | skip; // p is in W, or counts as met: stop offering it
#pragma unroll
				for (int k = 0; k < RPL; k++) {
					const int id = g + k * G;
					rin[k] = (settled & (id == cid_now)) ? true : ((drop & (id == gone)) ? false : rin[k]);
				}
#pragma unroll
				for (int j = 0; j < NV; j++) {
					bin[j] = (drop & ((gone == kIdLb + j) | (gone == kIdUb + j))) ? 0 : bin[j];
					bin[j] = (add & (cid_now == kIdLb + j)) ? (ceq ? 2 : -1) : bin[j];
					bin[j] = (add & (cid_now == kIdUb + j)) ? 1 : bin[j];
					bskip[j] |= (skip & (cid_now == kIdLb + j)) ? 1 : 0;
					bskip[j] |= (skip & (cid_now == kIdUb + j)) ? 2 : 0;
				}
				have = have & !settled;
			}
		}
		if (!done) verdict = kGiUndecided;
		// the working rows were excluded from the selection: make sure the updates kept them met
		if (verdict == kGiOptimal) {
#pragma unroll
			for (int s = 0; s < NV; s++) {
				double ax = 0.0, mag = fabs(sb[s]);
#pragma unroll
				for (int j = 0; j < NV; j++) {
					ax += sn[s][j] * x[j];
					mag += fabs(sn[s][j] * x[j]);
				}
				if (sid[s] >= 0 && fabs(ax - sb[s]) > kActiveTol * (1.0 + mag)) verdict = kGiUndecided;
			}
		}
		return verdict;
	}
};

} // namespace asif
