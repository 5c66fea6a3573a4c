// admm_wave.hpp -- one wavefront per QP, for problems too wide for the in-register solver
// (nv up to 32 variables, nc + nv <= 64 rows): the robust filter's full 18 x 12 problem, and any
// pre-assembled shape handed to asif_hip_qp_solve_batch.
//
// This is the layout the north star names: lane i owns constraint row i of [A; I] (its scaled
// coefficients in VGPRs, its z_i, y_i, l_i, u_i, rho_i), lane j < nv also owns variable j; the
// transposed row block, the rho-independent Schur matrix  S = sum_i w_i a_i a_i'  and the LDL' factor
// of  P + sigma I + rho S  live in LDS.  One ADMM iteration is
//     t_i = rho_i z_i - y_i                                  (lane i)
//     r_j = sigma x_j - q_j + sum_i At[j][i] t_i             (lane j, column walk in LDS, t broadcast)
//     L D L' x~ = r                                          (substitutions: v_readlane broadcast + LDS rows)
//     z~_i = a_i . x~                                        (lane i, x~_j by v_readlane, no LDS)
//     relaxation, projection on [l,u], dual update           (lane i)
// Form translation, scaling (power-of-two Ruiz), per-row rho, termination, infeasibility certificates
// and rho adaptation are those of admm_small.hpp / the OSQP paper; there is no active-set finish here,
// so accuracy is set by eps_abs/eps_rel.
#pragma once
#include "admm_small.hpp"

namespace asif {

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
	return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
	for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
	return v;
}
__device__ __forceinline__ double lane_bcast(double v, int src) // src wave-uniform
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}

template <int NVMAX>
struct AdmmWave {
	static constexpr int NVP = NVMAX + 1; // padded leading dimension of the nv x nv matrices
	static constexpr int RS = 65;         // padded row stride of the transposed row block
	static constexpr int kLdsDoubles = NVMAX * RS + 2 * NVMAX * NVP + 64;

	double *At, *Sm, *Lm, *tv; // LDS
	int lane, nv, m;
	bool isrow, isvar;
	// row state (lane i)
	double a[NVMAX], l, u, E, w, z, y, dy;
	// variable state (lane j)
	double P, q, D, x, dx, Dinv_;
	double rho, cs;

	__device__ __forceinline__ void sync() { __syncthreads(); }

	// dot of this lane's row with a variable-lane vector
	__device__ __forceinline__ double row_dot(double vvar) const
	{
		double s = 0.0;
#pragma unroll
		for (int j = 0; j < NVMAX; j++) s += a[j] * lane_bcast(vvar, j);
		return s;
	}
	// column sums: variable lane j gets sum_i At[j][i] * vrow_i
	__device__ __forceinline__ double col_dot(double vrow)
	{
		tv[lane] = vrow;
		sync();
		double s = 0.0;
		if (isvar)
			for (int i = 0; i < m; i++) s += At[lane * RS + i] * tv[i];
		sync();
		return s;
	}
	__device__ __forceinline__ void publish_rows()
	{
#pragma unroll
		for (int j = 0; j < NVMAX; j++) At[j * RS + lane] = a[j];
		sync();
	}

	__device__ __forceinline__ void scale(int iters)
	{
		D = 1.0;
		E = 1.0;
		cs = 1.0;
		for (int it = 0; it < iters; it++) {
			publish_rows();
			double Dt = 1.0;
			if (isvar) {
				double v = fabs(P);
				for (int i = 0; i < m; i++) v = fmax(v, fabs(At[lane * RS + i]));
				Dt = pow2_rsqrt(limit_scaling(v));
			}
			sync();
			double rn = 0.0;
#pragma unroll
			for (int j = 0; j < NVMAX; j++) rn = fmax(rn, fabs(a[j]));
			const double Et = pow2_rsqrt(limit_scaling(rn));
			E *= Et;
#pragma unroll
			for (int j = 0; j < NVMAX; j++) a[j] *= Et * lane_bcast(Dt, j);
			P *= Dt * Dt;
			q *= Dt;
			D *= Dt;
			const double cm = wave_sum(isvar ? fabs(P) : 0.0) / (double)nv;
			const double qn = limit_scaling(wave_max(isvar ? fabs(q) : 0.0));
			const double ct = pow2_floor(1.0 / limit_scaling(fmax(cm, qn)));
			P *= ct;
			q *= ct;
			cs *= ct;
		}
		publish_rows();
	}

	// S = sum_i w_i a_i a_i' (lower triangle), once
	__device__ __forceinline__ void build_S()
	{
		tv[lane] = w;
		sync();
		if (isvar)
			for (int c = 0; c <= lane; c++) {
				double s = 0.0;
				for (int i = 0; i < m; i++) s += tv[i] * At[lane * RS + i] * At[c * RS + i];
				Sm[lane * NVP + c] = s;
			}
		sync();
	}

	// LDL' of P + sigma I + rho S in LDS (right-looking, one column per step)
	__device__ __forceinline__ bool factor(double rho0, double sigma)
	{
		rho = rho0;
		if (isvar)
			for (int c = 0; c <= lane; c++) Lm[lane * NVP + c] = rho0 * Sm[lane * NVP + c] + (c == lane ? P + sigma : 0.0);
		sync();
		bool ok = true;
		for (int k = 0; k < nv; k++) {
			const double dk = Lm[k * NVP + k];
			ok = ok && (dk > 0.0);
			double ljk = 0.0;
			const bool below = isvar && lane > k;
			if (below) ljk = Lm[lane * NVP + k] / dk;
			sync();
			if (below) Lm[lane * NVP + k] = ljk;
			sync();
			if (below)
				for (int c = k + 1; c <= lane; c++) Lm[lane * NVP + c] -= ljk * Lm[c * NVP + k] * dk;
			sync();
		}
		Dinv_ = isvar ? 1.0 / Lm[lane * NVP + lane] : 0.0;
		return ok;
	}

	// in-place solve of L D L' v = r on the variable lanes
	__device__ __forceinline__ double solve(double r)
	{
		for (int k = 0; k < nv - 1; k++) {
			const double rk = lane_bcast(r, k);
			if (isvar && lane > k) r -= Lm[lane * NVP + k] * rk;
		}
		r *= Dinv_;
		for (int k = nv - 1; k >= 1; k--) {
			const double rk = lane_bcast(r, k);
			if (lane < k) r -= Lm[k * NVP + lane] * rk;
		}
		return r;
	}

	__device__ __forceinline__ void iterate(double sigma, double alpha)
	{
		const double rho_i = w * rho;
		const double t = isrow ? rho_i * z - y : 0.0;
		double r = col_dot(t);
		r += sigma * x - q;
		const double xt = isvar ? solve(r) : 0.0;
		const double zt = row_dot(xt);
		const double oma = 1.0 - alpha;
		if (isvar) {
			const double xn = alpha * xt + oma * x;
			dx = xn - x;
			x = xn;
		}
		if (isrow) {
			const double zr = alpha * zt + oma * z;
			const double zn = fmin(fmax(zr + y / rho_i, l), u);
			dy = rho_i * (zr - zn);
			y += dy;
			z = zn;
		}
	}
};

// One block of 64 threads = one wavefront = one QP.
template <int NVMAX>
__global__ __launch_bounds__(64) void qp_wave_kernel(asif_hip_solver S_, QpArgs a)
{
	using W = AdmmWave<NVMAX>;
	__shared__ double lds[W::kLdsDoubles];
	W s;
	s.At = lds;
	s.Sm = lds + NVMAX * W::RS;
	s.Lm = s.Sm + NVMAX * W::NVP;
	s.tv = s.Lm + NVMAX * W::NVP;
	const int lane = threadIdx.x;
	const int64_t qi = xcd_contiguous_index(blockIdx.x, a.B);
	if (qi >= a.B) return; // wave-uniform (one wave per workgroup)
	const int nv = a.nv, nc = a.nc, m = nc + nv;
	const int64_t ld = a.ld;
	s.lane = lane;
	s.nv = nv;
	s.m = m;
	s.isrow = lane < m;
	s.isvar = lane < nv;
	const bool isgen = lane < nc;

	// ---- form translation (src/qpwrapper_osqp.cpp:263-376): rows [A; I], l = [b; lb], u = [inf | b; ub]
#pragma unroll
	for (int j = 0; j < NVMAX; j++) {
		double v = 0.0;
		if (j < nv) {
			if (isgen) v = a.A[(int64_t)(lane + j * nc) * ld + qi];
			else if (s.isrow) v = (lane - nc == j) ? 1.0 : 0.0;
		}
		s.a[j] = v;
	}
	double lo = -kInfty, hi = kInfty;
	if (isgen) {
		lo = a.b[(int64_t)lane * ld + qi];
		hi = ((a.be_mask >> lane) & 1ull) ? lo : kInfty;
	} else if (s.isrow) {
		lo = a.lb[(int64_t)(lane - nc) * ld + qi];
		hi = a.ub[(int64_t)(lane - nc) * ld + qi];
	}
	s.P = s.isvar ? 2.0 * a.Hd[(int64_t)lane * ld + qi] : 1.0;
	s.q = s.isvar ? a.c[(int64_t)lane * ld + qi] : 0.0;
	s.scale(S_.scaling_iters);
	s.l = lo * s.E;
	s.u = hi * s.E;
	// per-row rho weight (OSQP's rho_vec classes); lanes beyond m carry no row
	if (!s.isrow) s.w = 0.0;
	else if (s.l < -kInfty * kMinScaling && s.u > kInfty * kMinScaling) s.w = 1e-5;
	else if (s.u - s.l < kRhoTol) s.w = kRhoEqOverIneq;
	else s.w = 1.0;
	s.build_S();
	s.x = 0.0;
	s.dx = 0.0;
	s.z = 0.0;
	s.y = 0.0;
	s.dy = 0.0;
	bool fact_ok = s.factor(S_.rho, S_.sigma);
	const double cinv = pow2_inv(s.cs);
	// without an active-set finish a check only tests residuals: no point in checking more often than OSQP-ish
	const int K = S_.check_interval > 10 ? S_.check_interval : 10;
	int status = 0, it = 0;
	while (it < S_.max_iter && status == 0) {
		for (int k = 0; k < K; k++) s.iterate(S_.sigma, S_.alpha);
		it += K;
		const bool last = it >= S_.max_iter;
		// ---- residuals (unscaled for termination, scaled for the rho estimate)
		const double ax = s.row_dot(s.x);
		const double aty = s.col_dot(s.isrow ? s.y : 0.0);
		const double ei = s.isrow ? pow2_inv(s.E) : 0.0;
		const double pri = wave_max(s.isrow ? fabs(ei * (ax - s.z)) : 0.0);
		const double nz = wave_max(fabs(ei * s.z)), nax = wave_max(fabs(ei * ax));
		const double pri_s = wave_max(s.isrow ? fabs(ax - s.z) : 0.0);
		const double nz_s = wave_max(s.isrow ? fabs(s.z) : 0.0), nax_s = wave_max(s.isrow ? fabs(ax) : 0.0);
		const double di = s.isvar ? pow2_inv(s.D) : 0.0;
		const double px = s.isvar ? s.P * s.x : 0.0;
		const double rd = s.isvar ? px + s.q + aty : 0.0;
		const double dua = cinv * wave_max(fabs(di * rd));
		const double nq = wave_max(fabs(di * s.q)), naty = wave_max(fabs(di * aty)), npx = wave_max(fabs(di * px));
		const double dua_s = wave_max(fabs(rd));
		const double nq_s = wave_max(s.isvar ? fabs(s.q) : 0.0), naty_s = wave_max(s.isvar ? fabs(aty) : 0.0);
		const double npx_s = wave_max(fabs(px));
		// ---- primal infeasibility: projected delta_y
		double v = s.isrow ? s.dy : 0.0;
		if (s.u > kInfty * kMinScaling) v = (s.l < -kInfty * kMinScaling) ? 0.0 : fmin(v, 0.0);
		else if (s.l < -kInfty * kMinScaling) v = fmax(v, 0.0);
		const double ndy = wave_max(fabs(s.E * v));
		const double lhs = wave_sum(s.isrow ? s.u * fmax(v, 0.0) + s.l * fmin(v, 0.0) : 0.0);
		const double atdy = s.col_dot(v);
		const double natdy = wave_max(fabs(di * atdy));
		// ---- dual infeasibility: delta_x
		const double ndx = wave_max(s.isvar ? fabs(s.D * s.dx) : 0.0);
		const double qdx = wave_sum(s.isvar ? s.q * s.dx : 0.0);
		const double npdx = wave_max(s.isvar ? fabs(s.P * s.dx * di) : 0.0);
		const double adx = s.row_dot(s.dx) * ei;
		int st = 0;
		for (int approx = 0; approx <= (last ? 1 : 0) && !st; approx++) {
			const double k = approx ? 10.0 : 1.0;
			const double ea = k * S_.eps_abs, er = k * S_.eps_rel, epi = k * S_.eps_prim_inf, edi = k * S_.eps_dual_inf;
			const bool prim_ok = pri < ea + er * fmax(nz, nax);
			const bool dual_ok = dua < ea + er * cinv * fmax(nq, fmax(naty, npx));
			if (prim_ok && dual_ok) st = approx ? kStatusSolvedInaccurate : kStatusSolved;
			else if (!prim_ok && ndy > epi && lhs < -epi * ndy && natdy < epi * ndy)
				st = approx ? kStatusPrimalInfInaccurate : kStatusPrimalInf;
			else if (!dual_ok && ndx > edi && qdx < -s.cs * edi * ndx && npdx < s.cs * edi * ndx) {
				const bool bad = s.isrow && ((s.u < kInfty * kMinScaling && adx > edi * ndx) ||
				                             (s.l > -kInfty * kMinScaling && adx < -edi * ndx));
				if (!__any(bad)) st = approx ? kStatusDualInfInaccurate : kStatusDualInf;
			}
		}
		if (!st && (last || !fact_ok)) st = kStatusMaxIter;
		status = st;
		if (!st && S_.adaptive_rho) {
			const double pr = pri_s / (fmax(nz_s, nax_s) + 1e-10);
			const double dr = dua_s / (fmax(nq_s, fmax(naty_s, npx_s)) + 1e-10);
			double rn = s.rho * sqrt(pr / (dr + 1e-10));
			rn = fmin(fmax(rn, kRhoMin), kRhoMax);
			if (rn > s.rho * S_.adaptive_rho_tolerance || rn < s.rho / S_.adaptive_rho_tolerance)
				fact_ok = s.factor(rn, S_.sigma) && fact_ok;
		}
	}
	if (status == 0) status = kStatusMaxIter;
	if (status == kStatusSolvedInaccurate) status = kStatusSolved;
	if (s.isvar) a.sol[(int64_t)lane * ld + qi] = s.D * s.x;
	if (lane == 0) {
		a.status[qi] = status;
		if (a.iters) a.iters[qi] = it;
	}
}

} // namespace asif
