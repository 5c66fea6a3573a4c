// host_gi_driver.cpp -- TEST HARNESS: the product's in-register dual active-set solver (asif_amd/csrc/gi_small.hpp,
// one lane per QP) compiled with g++ and exposed through a C entry, so that tests/test_gi_host.py can compare it
// with the oracle's exact enumeration on the CPU.  Not part of any library the product ships.
#include <stdint.h>
#include "gi_small.hpp"

using namespace asif;

template <int NV, int NC>
static void run(int64_t B, const double *Hd, const double *c, const double *A, const double *b, const double *lb,
                const double *ub, const uint8_t *be, int nc, int max_steps, double *sol, int32_t *status, int32_t *steps)
{
	for (int64_t i = 0; i < B; i++) {
		QpLaneData<NV, NC> qp;
		for (int j = 0; j < NV; j++) {
			qp.Hd[j] = Hd[i * NV + j];
			qp.c[j] = c[i * NV + j];
			qp.lb[j] = lb[i * NV + j];
			qp.ub[j] = ub[i * NV + j];
		}
		for (int r = 0; r < NC; r++) {
			const bool valid = r < nc;
			for (int j = 0; j < NV; j++) qp.A[r][j] = valid ? A[i * nc * NV + r + j * nc] : 0.0; // col-major nc x nv
			qp.b[r] = valid ? b[i * nc + r] : -1e20;
			qp.eq[r] = valid && be && be[r];
		}
		double x[NV];
		int st;
		const int v = GiSmall<NV, NC, 1>::solve(qp, 0, max_steps, x, st);
		for (int j = 0; j < NV; j++) sol[i * NV + j] = x[j];
		status[i] = v;
		steps[i] = st;
	}
}

// AoS per instance like the oracle's or_qp_solve_batch: Hd[nv], c[nv], A[nc*nv] col-major, b[nc], lb[nv], ub[nv].
// status: 1 optimal, 2 infeasible, 0 undecided.  Returns 0, or -1 for a shape without an instantiation.
extern "C" int gi_host_solve_batch(int nv, int nc, int64_t B, const double *Hd, const double *c, const double *A,
                                   const double *b, const double *lb, const double *ub, const uint8_t *be,
                                   int max_steps, double *sol, int32_t *status, int32_t *steps)
{
#define SHAPE(NV_, NC_) \
	if (nv == NV_ && nc <= NC_) { \
		run<NV_, NC_>(B, Hd, c, A, b, lb, ub, be, nc, max_steps, sol, status, steps); \
		return 0; \
	}
	SHAPE(1, 8)
	SHAPE(2, 4)
	SHAPE(2, 18)
	SHAPE(2, 48)
	SHAPE(3, 17)
	SHAPE(3, 41)
	SHAPE(3, 64)
#undef SHAPE
	return -1;
}
