/*
 * ref_affa_shim.cpp -- thin extern "C" driver around the REFERENCE's libaffa
 * (lib/libaffa/src/ of /root/reference, compiled unmodified by `make -C oracle ref` into oracle/_ref).
 * TEST INFRASTRUCTURE: used to pin oracle/or_affine.c (and, through it, the device affine forms)
 * and to generate tests/golden/affa_*.json.  It contains none of the reference's code; it only
 * calls the AAF class through its public interface.
 *
 * A "program" is a list of register instructions so the very same sequence can be replayed on
 * or_affine.c:   {op, dst, a, b, imm0, imm1}
 */
#include "aa.h"
#include <cmath>
#include <vector>

extern "C" {

enum { OP_CONST = 0, OP_INTERVAL = 1, OP_ADD = 2, OP_SUB = 3, OP_MUL = 4, OP_DIV = 5, OP_INV = 6,
       OP_NEG = 7, OP_SCALE = 8, OP_SIN = 9, OP_COPY = 10 };

struct ref_instr {
	int op, dst, a, b;
	double imm0, imm1;
};

/* Runs the program on nreg AAF registers with a fresh symbol counter. For register r writes
 * center[r], n[r], lo[r], hi[r], and (idx, coeff) pairs into idx[r*cap..], coef[r*cap..]. */
int ref_affa_run(const ref_instr *prog, int nprog, int nreg, int cap, double *center, int *n, double *lo,
                 double *hi, unsigned *idx, double *coef)
{
	AAF::set_default(0);
	std::vector<AAF> R(nreg);
	for (int p = 0; p < nprog; p++) {
		const ref_instr &I = prog[p];
		switch (I.op) {
		case OP_CONST: R[I.dst] = AAF(I.imm0); break;
		case OP_INTERVAL: R[I.dst] = AAF(interval(I.imm0, I.imm1)); break;
		case OP_ADD: R[I.dst] = R[I.a] + R[I.b]; break;
		case OP_SUB: R[I.dst] = R[I.a] - R[I.b]; break;
		case OP_MUL: R[I.dst] = R[I.a] * R[I.b]; break;
		case OP_DIV: R[I.dst] = R[I.a] / R[I.b]; break;
		case OP_INV: R[I.dst] = inv(R[I.a]); break;
		case OP_NEG: R[I.dst] = -R[I.a]; break;
		case OP_SCALE: R[I.dst] = R[I.a] * I.imm0; break;
		case OP_SIN: R[I.dst] = sin(R[I.a]); break;
		case OP_COPY: R[I.dst] = R[I.a]; break;
		default: return -1;
		}
	}
	for (int r = 0; r < nreg; r++) {
		center[r] = R[r].get_center();
		n[r] = (int)R[r].get_length();
		interval iv = R[r].convert();
		lo[r] = iv.left();
		hi[r] = iv.right();
		if (n[r] > cap) return -2;
		for (int k = 0; k < n[r]; k++) {
			idx[r * cap + k] = R[r].get_index(k);
			coef[r * cap + k] = R[r].get_coeff(k);
		}
	}
	return 0;
}

/* Interval Lie derivatives of the robust pendulum exactly as the reference computes them:
 * ASIFrobust::updateConstraints (src/asif_robust.cpp:282-337) with the model callbacks of
 * examples/InvertedPendulum_Robust.cpp:53-70, the templated matmul (include/asif_utils.h:22-62)
 * restated here on AAF because that header cannot be included without <osqp.h>.
 * hp = N pairs {a0,a1}; outputs lo/hi of Lfh[N], Lgh[N] and h[N]. */
int ref_ip_robust_lie(const double *x, double pMin, double pMax, const double *hp, int N, double *h,
                      double *Lfh_lo, double *Lfh_hi, double *Lgh_lo, double *Lgh_hi)
{
	const int nx = 2, nu = 1;
	AAF::set_default(0);
	std::vector<AAF> xI(nx), f(nx), g(nx * nu), DhI(N * nx), Lfh(N), Lgh(N * nu);
	for (int i = 0; i < nx; i++) xI[i] = interval(x[i]);
	std::vector<double> Dh(N * nx);
	for (int i = 0; i < N; i++) {
		h[i] = 1. - hp[2 * i] * x[0] - hp[2 * i + 1] * x[1];
		Dh[i] = -hp[2 * i];
		Dh[i + N] = -hp[2 * i + 1];
	}
	f[0] = xI[1];
	f[1] = sin(xI[0]);
	g[0] = 0.;
	g[1] = interval(pMin, pMax);
	for (int i = 0; i < N * nx; i++) DhI[i] = interval(Dh[i]);
	for (int i = 0; i < N; i++) {
		Lfh[i] = 0.0;
		for (int k = 0; k < nx; k++) Lfh[i] = Lfh[i] + DhI[i + k * N] * f[k];
	}
	for (int i = 0; i < N; i++)
		for (int j = 0; j < nu; j++) {
			Lgh[i + j * N] = 0.0;
			for (int k = 0; k < nx; k++) Lgh[i + j * N] = Lgh[i + j * N] + DhI[i + k * N] * g[k + j * nx];
		}
	for (int i = 0; i < N; i++) {
		interval a = Lfh[i].convert(), b = Lgh[i].convert();
		Lfh_lo[i] = a.left();
		Lfh_hi[i] = a.right();
		Lgh_lo[i] = b.left();
		Lgh_hi[i] = b.right();
	}
	return 0;
}

/* Interval Lie derivatives over the facets of a realizable kernel exactly as the reference computes
 * them: xFaceInt of ASIFrealizable::initialize (src/asif_realizable.cpp:137-157), the interval dynamics
 * of examples/DoubleIntegrator_RealizableSampled.cpp:33-54 (globals m, K, F constructed first), and
 * Lfh/Lgh of updateConstraints (:465-506) for every (facet, active constraint) pair.  nx = 2, nu = 1.
 * out[(i*nA + j)*4 ..] = lo(Lgh), hi(Lgh), lo(Lfh), hi(Lfh). */
int ref_rz_facet_lie(int nFacets, int nA, const double *vertices, const int *facetVertices,
                     const double *facetNormals, const int *facetActive, double mMin, double mMax, double Klo,
                     double Khi, double Flo, double Fhi, double *out)
{
	const int nx = 2, nu = 1;
	AAF::set_default(0);
	const AAF mInt = interval(mMin, mMax), KInt = interval(Klo, Khi), FInt = interval(Flo, Fhi);
	std::vector<std::vector<AAF>> xFace(nFacets, std::vector<AAF>(nx));
	for (int i = 0; i < nFacets; i++) {
		for (int j = 0; j < nx; j++) xFace[i][j] = vertices[facetVertices[i * nx + 0] * nx + j];
		for (int j = 1; j <= nx - 1; j++) {
			AAF lam = AAF(0., 1.);
			const double *vertex = &vertices[facetVertices[i * nx + j] * nx];
			for (int k = 0; k < nx; k++) xFace[i][k] = lam * xFace[i][k] + (1. - lam) * vertex[k];
		}
	}
	for (int i = 0; i < nFacets; i++)
		for (int j = 0; j < nA; j++) {
			const double *normal = &facetNormals[facetActive[i * nA + j] * nx];
			AAF Dh[nx], f[nx], g[nx * nu];
			for (int k = 0; k < nx; k++) Dh[k] = interval(-normal[k]);
			const AAF *x = xFace[i].data();
			f[0] = x[1];
			f[1] = -FInt * x[1] / mInt;
			g[0] = 0.;
			g[1] = KInt / mInt;
			AAF Lfh = 0., Lgh = 0.;
			for (int k = 0; k < nx; k++) Lfh = Lfh + f[k] * Dh[k];
			for (int k = 0; k < nx; k++) Lgh = Lgh + g[k] * Dh[k];
			const interval a = Lgh.convert(), b = Lfh.convert();
			double *o = &out[(i * nA + j) * 4];
			o[0] = a.left();
			o[1] = a.right();
			o[2] = b.left();
			o[3] = b.right();
		}
	return 0;
}

/* Interval Lie derivatives of examples/DoubleIntegrator_Robust.cpp exactly as ASIFrobust::updateConstraints
 * (src/asif_robust.cpp:282-337) computes them for M selected half-planes hp[M][2] at the point state x:
 * globals m, K, F first (:28-37), xInt, dynamics (:51-58), DhInt column-major, all Lfh, then all Lgh
 * (include/asif_utils.h:22-62 restated on AAF).  out[s*4..] = lo(Lgh), hi(Lgh), lo(Lfh), hi(Lfh). */
int ref_di_robust_lie(const double *x, int M, const double *hp, double mMin, double mMax, double Klo, double Khi,
                      double Flo, double Fhi, double *out)
{
	const int nx = 2, nu = 1;
	AAF::set_default(0);
	const AAF mInt = interval(mMin, mMax), KInt = interval(Klo, Khi), FInt = interval(Flo, Fhi);
	std::vector<AAF> xI(nx), f(nx), g(nx * nu), DhI(M * nx), Lfh(M), Lgh(M * nu);
	for (int i = 0; i < nx; i++) xI[i] = interval(x[i]);
	f[0] = xI[1];
	f[1] = -FInt * xI[1] / mInt;
	g[0] = 0.;
	g[1] = KInt / mInt;
	std::vector<double> Dh(M * nx);
	for (int i = 0; i < M; i++) {
		Dh[i] = -hp[2 * i];
		Dh[i + M] = -hp[2 * i + 1];
	}
	for (int i = 0; i < M * nx; i++) DhI[i] = interval(Dh[i]);
	for (int i = 0; i < M; i++) {
		Lfh[i] = 0.0;
		for (int k = 0; k < nx; k++) Lfh[i] = Lfh[i] + DhI[i + k * M] * f[k];
	}
	for (int i = 0; i < M; i++)
		for (int j = 0; j < nu; j++) {
			Lgh[i + j * M] = 0.0;
			for (int k = 0; k < nx; k++) Lgh[i + j * M] = Lgh[i + j * M] + DhI[i + k * M] * g[k + j * nx];
		}
	for (int i = 0; i < M; i++) {
		const interval a = Lgh[i].convert(), b = Lfh[i].convert();
		out[i * 4 + 0] = a.left();
		out[i * 4 + 1] = a.right();
		out[i * 4 + 2] = b.left();
		out[i * 4 + 3] = b.right();
	}
	return 0;
}

/* f, g midpoints of the same dynamics at a point state (src/asif_realizable.cpp:533-553) */
int ref_rz_point_dynamics(const double *x, double mMin, double mMax, double Klo, double Khi, double Flo,
                          double Fhi, double *f, double *g)
{
	AAF::set_default(0);
	const AAF mInt = interval(mMin, mMax), KInt = interval(Klo, Khi), FInt = interval(Flo, Fhi);
	AAF xI[2], fI[2], gI[2];
	for (int i = 0; i < 2; i++) xI[i] = interval(x[i]);
	fI[0] = xI[1];
	fI[1] = -FInt * xI[1] / mInt;
	gI[0] = 0.;
	gI[1] = KInt / mInt;
	for (int i = 0; i < 2; i++) {
		f[i] = fI[i].convert().mid();
		g[i] = gI[i].convert().mid();
	}
	return 0;
}

} /* extern "C" */
