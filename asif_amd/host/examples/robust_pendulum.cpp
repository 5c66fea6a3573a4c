// robust_pendulum.cpp -- ASIF::ASIFrobust on the model of examples/InvertedPendulum_Robust.cpp:53-70
// (ROBUST flavour: g[1] in [0.8, 1.2]) with the box half-planes of SURVEY 8(d): single-agent filter()
// (affine-arithmetic rows on the host, the full 18-variable QP on the GPU's wave-per-QP kernel) next to
// filterBatch() on the same seeded states.
//   usage: robust_pendulum [--solver host] --loop STEPS [P [UDES]]   (the example's main loop)   |   robust_pendulum N    prints  i,uAct,relax,rc,uActBatch,rcBatch, then "A,<i>,<216 row entries>" lines
#include <asif++.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <algorithm>
#include <cstring>
#include <vector>

static const double pMin = 0.8, pMax = 1.2;
static std::vector<std::vector<double>> SafetySetData;

static void safetySet(const double *x, double *h, double *Dh)
{
	for (uint32_t i = 0; i < SafetySetData.size(); i++) {
		h[i] = 1. - SafetySetData[i][0] * x[0] - SafetySetData[i][1] * x[1];
		Dh[i] = -SafetySetData[i][0];
		Dh[i + SafetySetData.size()] = -SafetySetData[i][1];
	}
}
static void dynamics(const interval_t *x, interval_t *f, interval_t *g)
{
	f[0] = x[1];
	f[1] = sin(x[0]);
	g[0] = 0.;
	g[1] = interval(pMin, pMax);
}
static double rng(uint64_t seed, uint64_t i, uint64_t j)
{
	uint64_t z = (seed << 32) + (i * 16 + j);
	z += 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	z = z ^ (z >> 31);
	return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// reaches the solver the class made for itself (timing mode below: Newton steps of the last solve)
struct RobustProbe : ASIF::ASIFrobust {
	using ASIF::ASIFrobust::ASIFrobust;
	ASIF::QPWrapperAbstract *solver() { return QPsolver_; }
};

int main(int argc, char **argv)
{
	QPSOLVER solver = QPSOLVER::HIP; // `--solver host`: single-agent filter() only, the QP on the calling thread, no device
	for (int i = 1; i + 1 < argc; i++)
		if (!std::strcmp(argv[i], "--solver")) {
			if (!std::strcmp(argv[i + 1], "host")) solver = QPSOLVER::HOST;
			for (int j = i; j + 2 < argc; j++) argv[j] = argv[j + 2];
			argc -= 2;
			break;
		}
	const bool host = solver == QPSOLVER::HOST;
	const long N = argc > 1 ? std::atol(argv[1]) : 16;
	const double a = 1.0 / M_PI;
	SafetySetData = {{a, 0}, {-a, 0}, {0, a}, {0, -a}};
	const double lb[1] = {-1.5}, ub[1] = {1.5};
	ASIF::ASIFrobust::Options opts; // examples/InvertedPendulum_Robust.cpp:120-121
	opts.relaxCost = 50.0;
	opts.relaxLb = 5.0;
	RobustProbe flt(2, 1, (uint32_t)SafetySetData.size(), safetySet, dynamics, (uint32_t)-1, solver);
	if (flt.initialize(lb, ub, opts) != 1) return 3;
	asif_hip_options md;
	asif_hip_default_options(ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST, ASIF_HIP_ROBUST, &md); // same half-planes, pMin, pMax
	if (!host && flt.bindDeviceModel(ASIF_HIP_MODEL_INVERTED_PENDULUM_ROBUST, md) != 0) return 3;
	if (argc > 2 && !std::strcmp(argv[1], "--loop")) {
		// the example's own loop (examples/InvertedPendulum_Robust.cpp:134-175, ROBUST flavour): from (0.5, 0) with
		// uDes = 0 by default; UDES = +-1.5 drives the velocity into its half-plane, where the filter takes the input
		// back -- the position rows have Lgh = 0 and can only be relaxed); the plant's input gain p is one of the
		// example's three values pMin + i (pMax - pMin) / 2.  Prints step,x0,x1,uAct,relax,rc with x the state handed to
		// filter().
		const long steps = std::atol(argv[2]);
		const double p = argc > 3 ? std::atof(argv[3]) : 1.0, dt = 0.01;
		double x[2] = {0.5, 0.0};
		const double ud[1] = {argc > 4 ? std::atof(argv[4]) : 0.0};
		std::printf("step,x0,x1,uAct,relax,rc\n");
		for (long i = 0; i < steps; i++) {
			double ua[1] = {0.0}, rl = 0.0;
			const int32_t rc = flt.filter(x, ud, ua, rl);
			std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%d\n", i, x[0], x[1], ua[0], rl, rc);
			const double f[2] = {x[1], std::sin(x[0])}, g[2] = {0., p};
			for (int k = 0; k < 2; k++) x[k] += dt * (f[k] + g[k] * ua[0]);
		}
		return 0;
	}
	if (argc > 2 && !std::strcmp(argv[1], "--loop-time")) {
		// the same loop, timed: one line with the cost of a filter() call (the lifted 18 x 12 problem through the class's
		// solver: QPWrapperHip, warm-started from the previous control step with ASIF_HIP_QP_WARM=1, or QPWrapperHost)
		const long steps = std::atol(argv[2]);
		const double p = argc > 3 ? std::atof(argv[3]) : 1.0, dt = 0.01;
		double x[2] = {0.5, 0.0};
		const double ud[1] = {argc > 4 ? std::atof(argv[4]) : 0.0};
		ASIF::QPWrapperHip *hipSolver = dynamic_cast<ASIF::QPWrapperHip *>(flt.solver());
		std::vector<double> us(steps), usOk;
		long newton = 0, maxNewton = 0, ok = 0, newtonOk = 0;
		for (long i = 0; i < steps; i++) {
			double ua[1] = {0.0}, rl = 0.0;
			const auto t0 = std::chrono::steady_clock::now();
			const int32_t rc = flt.filter(x, ud, ua, rl);
			us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
			ok += rc == 1;
			if (rc == 1) usOk.push_back(us[i]);
			if (hipSolver && rc == 1) newtonOk += hipSolver->lastIterations();
			if (hipSolver) {
				newton += hipSolver->lastIterations();
				maxNewton = std::max<long>(maxNewton, hipSolver->lastIterations());
			}
			const double f[2] = {x[1], std::sin(x[0])}, g[2] = {0., p};
			for (int k = 0; k < 2; k++) x[k] += dt * (f[k] + g[k] * ua[0]);
		}
		double mean = 0.0;
		for (double v : us) mean += v / (double)steps;
		std::sort(us.begin(), us.end());
		std::sort(usOk.begin(), usOk.end());
		// the steps whose problem had a solution are the ones a warm start can speak for (after any other verdict the
		// next start is cold, in OSQP too)
		std::printf("{\"steps\": %ld, \"rc_ok\": %ld, \"us_median_rc_ok\": %.3f, \"newton_mean_rc_ok\": %.3f, "
		            "\"us_median\": %.3f, \"us_mean\": %.3f, \"us_p99\": %.3f, "
		            "\"newton_mean\": %.3f, \"newton_max\": %ld, \"warm_start\": %d, \"final_x\": [%.17g, %.17g]}\n",
		            steps, ok, usOk.empty() ? 0.0 : usOk[usOk.size() / 2], ok ? (double)newtonOk / (double)ok : 0.0,
		            us[steps / 2], mean, us[(size_t)(0.99 * (double)(steps - 1))], (double)newton / (double)steps,
		            maxNewton, hipSolver ? (int)hipSolver->warmStart : -1, x[0], x[1]);
		return 0;
	}
	std::vector<double> bx(2 * N), bu(N), ba(N, 0.0), br(N, 0.0);
	std::vector<int32_t> brc(N, 0);
	for (long i = 0; i < N; i++) {
		bx[i] = -3.0 + 6.0 * rng(4, i, 0);
		bx[N + i] = -3.0 + 6.0 * rng(4, i, 1);
		bu[i] = -1.5 + 3.0 * rng(4, i, 2);
	}
	if (!host && flt.filterBatch(N, bx.data(), bu.data(), ba.data(), br.data(), brc.data()) != 0) return 4;
	std::printf("i,uAct,relax,rc,uActBatch,rcBatch\n");
	for (long i = 0; i < N; i++) {
		const double x[2] = {bx[i], bx[N + i]}, ud[1] = {bu[i]};
		double ua[1] = {0.0}, rl = 0.0;
		const int32_t rc = flt.filter(x, ud, ua, rl);
		std::printf("%ld,%.17g,%.17g,%d,%.17g,%d\n", i, ua[0], rl, rc, ba[i], brc[i]);
		std::printf("A,%ld", i);
		for (uint32_t e = 0; e < flt.nc() * flt.nv(); e++) std::printf(",%.17g", flt.rowsA()[e]);
		std::printf("\n");
	}
	return 0;
}
