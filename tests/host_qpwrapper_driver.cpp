// host_qpwrapper_driver.cpp -- TEST DRIVER for ASIF::QPWrapperHip (asif_amd/host): the plug-in contract of
// include/qpwrapper_abstract.h:16-51 as QPWrapperOsqp keeps it (src/qpwrapper_osqp.cpp:55-261).
// Prints one "name value..." line per check; tests/test_gpu_host_cpp.py compares.
#include <chrono>
#include <cstdio>
#include <vector>
#include "qpwrapper_hip.h"

using ASIF::QPWrapperHip;

int main()
{
	// 1. full cost matrix (diagonalCost = false): min x'Hx + c'x, H = [[2,1],[1,3]], c = (-2,-6), x1+x2 >= 1, 0<=x<=0.9
	{
		QPWrapperHip w(2, 1, false);
		const double H[4] = {2, 99 /*below the diagonal: not read*/, 1, 3}, c[2] = {-2, -6}, A[2] = {1, 1}, b[1] = {1};
		const double lb[2] = {0, 0}, ub[2] = {0.9, 0.9};
		const int r = w.initialize(H, c, A, b, lb, ub);
		const int st = w.solve();
		double x[2];
		w.getSolution(x);
		std::printf("dense %d %d %.15g %.15g\n", r, st, x[0], x[1]);
	}
	// 2. the first solve inside initialize() meets an infeasible problem: set-up succeeds (the reference ignores
	//    that solve, src/asif.cpp:101-105), solve() reports the verdict, an update makes it feasible
	{
		QPWrapperHip w(2, 2, true);
		const double H[4] = {1, 0, 0, 1}, c[2] = {0, 0};
		double A[4] = {1, -1, 0, 0}; // rows x0 >= 1 and -x0 >= 1
		const double b[2] = {1, 1}, lb[2] = {-5, -5}, ub[2] = {5, 5};
		const int r = w.initialize(H, c, A, b, lb, ub);
		const int st = w.solve();
		const double b2[2] = {1, -3};
		w.updateb(b2);
		const int st2 = w.solve();
		double x[2];
		w.getSolution(x);
		std::printf("infeasible_init %d %d %d %d %.15g %.15g\n", r, st, w.lastError(), st2, x[0], x[1]);
	}
	// 3. the largest shape the reference's classes construct: 86 x 65 (ASIFrealizable on the 10 Hz kernel)
	{
		const unsigned nv = 86, nc = 65;
		QPWrapperHip w(nv, nc, true);
		std::vector<double> H(nv * nv, 0.0), c(nv, -2.0), A(nc * nv, 0.0), b(nc, -1.0), lb(nv, 0.0), ub(nv, 0.25);
		for (unsigned j = 0; j < nv; j++) H[j + j * nv] = 1.0;
		for (unsigned i = 0; i < nc; i++) A[i + i * nc] = 1.0; // x_i >= -1: inactive
		const int r = w.initialize(H.data(), c.data(), A.data(), b.data(), lb.data(), ub.data());
		const int st = w.solve();
		std::vector<double> x(nv);
		w.getSolution(x.data());
		double mn = x[0], mx = x[0];
		for (double v : x) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
		std::printf("big %d %d %.15g %.15g\n", r, st, mn, mx);
	}
	// 3b. what one agent's control step costs at the plug-in: update + solve + read back of the 2 x 4 explicit-class QP
	{
		QPWrapperHip w(2, 4, true);
		const double H[4] = {1, 0, 0, 10}, c[2] = {-1, -20}, lb[2] = {-1, 1}, ub[2] = {1, 1};
		double A[8] = {0.5, -0.5, 1, -1, 1, 1, 1, 1}, b[4] = {-1, -1, -2, -2};
		w.initialize(H, c, A, b, lb, ub);
		double x[2];
		for (int k = 0; k < 20; k++) w.solve();
		const auto t0 = std::chrono::steady_clock::now();
		const int n = 200;
		int st = 0;
		for (int k = 0; k < n; k++) {
			b[0] = -1.0 - 1e-3 * k;
			w.updateb(b);
			st = w.solve();
			w.getSolution(x);
		}
		const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
		std::printf("latency_us %d %.2f\n", st, us);
	}
	// 4. a shape beyond the kernels is a set-up error, not a solver verdict
	{
		QPWrapperHip w(140, 4, true);
		std::vector<double> H(140 * 140, 0.0), c(140, 0.0), A(4 * 140, 0.0), b(4, -1.0), lb(140, 0.0), ub(140, 1.0);
		const int r = w.initialize(H.data(), c.data(), A.data(), b.data(), lb.data(), ub.data());
		std::printf("toobig %d %d %d\n", r, w.solve(), w.lastError());
	}
	return 0;
}
