// host_alm_driver.cpp -- TEST HARNESS: asif_amd/host/qp_alm_host.cpp (the wave kernels' method on the calling thread,
// behind ASIF::QPWrapperHost for shapes beyond nv = 3) exposed through a C entry so that tests/test_alm_host.py can
// compare it with the oracle's exact solver on the CPU.  Not part of any library the product ships.
#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace ASIF {
namespace hostqp {
int solve_alm(int nv, int nc, bool diag, const double *H, const double *c, const double *A, const double *b,
              const double *lb, const double *ub, const bool *be, double eps_rel, int max_newton, double *sol, int *newton_out,
              double *warm_x, double *warm_y, bool warm_in);
}
} // namespace ASIF

// AoS per instance like or_qp_solve_batch.  diag != 0: Hd[nv] per instance; else H[nv*nv] column-major per instance.
extern "C" int alm_host_solve_batch(int nv, int nc, int64_t B, int diag, const double *H, const double *c, const double *A,
                                    const double *b, const double *lb, const double *ub, const uint8_t *be, double eps_rel,
                                    int max_newton, double *sol, int32_t *status, int32_t *newton, double *warm_x,
                                    double *warm_y, int warm_in)
{
	std::vector<double> Hf((size_t)nv * nv, 0.0);
	std::vector<uint8_t> beb(nc > 0 ? nc : 1, 0);
	for (int r = 0; r < nc; r++) beb[r] = be ? (be[r] != 0) : 0;
	for (int64_t i = 0; i < B; i++) {
		const double *Hi;
		if (diag) {
			for (int j = 0; j < nv; j++) Hf[j + (size_t)j * nv] = H[i * nv + j];
			Hi = Hf.data();
		} else {
			Hi = H + i * nv * nv;
		}
		int nw = 0;
		status[i] = ASIF::hostqp::solve_alm(nv, nc, diag != 0, Hi, c + i * nv, A + i * (size_t)nc * nv, b + i * nc, lb + i * nv,
		                                    ub + i * nv, reinterpret_cast<const bool *>(beb.data()), eps_rel, max_newton,
		                                    sol + i * nv, &nw, warm_x ? warm_x + i * nv : nullptr,
		                                    warm_y ? warm_y + i * (size_t)(nc + nv) : nullptr, warm_in != 0);
		newton[i] = nw;
	}
	return 0;
}
