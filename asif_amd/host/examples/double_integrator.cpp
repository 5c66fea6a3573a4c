// double_integrator.cpp -- C1 of BASELINE.json: the closed loop of the reference's
// examples/DoubleIntegrator.cpp (:63-116) on this build's ASIF::ASIF class, single agent, the QP of
// every control step solved by the GPU plug-in.  Prints one CSV row per step
//   t,x,v,uDes,uAct,relax,rc
// for the first `steps` steps (default 2500 = up to the updateOptions call at t > 2.5 s, after which
// the reference's behaviour depends on OSQP internals: lb_relax = 6 > ub_relax = 5).
// --solver host: the QP on the calling thread (ASIF::QPWrapperHost, `QPSOLVER::HOST`) -- BASELINE config 1 as written,
// "CPU path, no GPU"; the default is the GPU plug-in (QPWrapperHip).
// With --batch B it also runs B copies of the current state through filterBatch() at every 100th step
// and checks they agree with the single-agent answer.  With --time every filter() call is clocked on the host and a
// one-line JSON summary (median / mean / p99 microseconds per call) goes to stderr -- bench.py's `c1` entry.
#include <asif++.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static const uint32_t nx = 2, nu = 1, npSS = 4;
static const double lb[nu] = {-1.0}, ub[nu] = {1.0};

// model callbacks = the reference example's (examples/DoubleIntegrator.cpp:24-61)
static void safetySet(const double *x, double *h, double *Dh)
{
	const double brake = (x[1] * x[1]) / 2.0;
	const bool fwd = x[1] > 0;
	h[0] = fwd ? (1.0 - x[0] - brake) : (-x[0] + 1.0);
	h[1] = fwd ? (x[0] + 1.0) : (x[0] + 1.0 - brake);
	h[2] = x[1] + 1.0;
	h[3] = -x[1] + 1.0;
	Dh[0] = -1.0; Dh[4] = fwd ? -x[1] : 0.0;
	Dh[1] = 1.0;  Dh[5] = fwd ? 0.0 : -x[1];
	Dh[2] = 0.0;  Dh[6] = 1.0;
	Dh[3] = 0.0;  Dh[7] = -1.0;
}
static void dynamics(const double *x, double *f, double *g)
{
	f[0] = x[1];
	f[1] = 0.0;
	g[0] = 0.0;
	g[1] = 1.0;
}

int main(int argc, char **argv)
{
	int steps = 2500;
	long batch = 0;
	bool timing = false;
	QPSOLVER solver = QPSOLVER::HIP;
	for (int i = 1; i < argc; i++) {
		if (!std::strcmp(argv[i], "--solver") && i + 1 < argc) {
			solver = !std::strcmp(argv[++i], "host") ? QPSOLVER::HOST : QPSOLVER::HIP;
			continue;
		}
		if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
		else if (!std::strcmp(argv[i], "--time")) timing = true;
		else if (!std::strcmp(argv[i], "--batch") && i + 1 < argc) batch = std::atol(argv[++i]);
	}
	ASIF::ASIF *asif = new ASIF::ASIF(nx, nu, npSS, safetySet, dynamics, npSS, solver);
	const int32_t ir = asif->initialize(lb, ub);
	if (ir != 1) {
		std::fprintf(stderr, "initialize failed: %d (%s)\n", ir, asif_hip_error_string(ir));
		return 2;
	}
	if (batch > 0 && asif->bindDeviceModel(ASIF_HIP_MODEL_DOUBLE_INTEGRATOR) != 0) {
		std::fprintf(stderr, "bindDeviceModel failed\n");
		return 2;
	}
	const double dt = 0.001;
	double xNow[2] = {0.0, 0.0}, uDesNow[1] = {1.0}, uActNow[1] = {0.0}, tNow = 0.0, relax = 0.0;
	int bad = 0;
	std::vector<double> us;
	us.reserve(steps);
	std::printf("t,x,v,uDes,uAct,relax,rc\n");
	for (int k = 0; k < steps; k++) {
		const auto c0 = std::chrono::steady_clock::now();
		const int32_t rc = asif->filter(xNow, uDesNow, uActNow, relax);
		us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count());
		if (batch > 0 && k % 100 == 0) {
			std::vector<double> bx(2 * batch), bu(batch, uDesNow[0]), ba(batch, 0.0), br(batch, 0.0);
			std::vector<int32_t> brc(batch, 0);
			for (long i = 0; i < batch; i++) {
				bx[i] = xNow[0];
				bx[batch + i] = xNow[1];
			}
			if (asif->filterBatch(batch, bx.data(), bu.data(), ba.data(), br.data(), brc.data()) != 0) bad++;
			for (long i = 0; i < batch; i++)
				if (brc[i] != rc || std::fabs(ba[i] - uActNow[0]) > 1e-12) bad++;
		}
		double f[2], g[2];
		dynamics(xNow, f, g);
		for (uint32_t i = 0; i < nx; i++) xNow[i] += dt * (f[i] + uActNow[0] * g[i]);
		tNow += dt;
		std::printf("%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d\n", tNow, xNow[0], xNow[1], uDesNow[0], uActNow[0], relax, rc);
	}
	delete asif;
	if (timing && !us.empty()) {
		double sum = 0.0;
		for (double v : us) sum += v;
		std::sort(us.begin(), us.end());
		std::fprintf(stderr, "{\"median_us\": %.3f, \"mean_us\": %.3f, \"p99_us\": %.3f, \"min_us\": %.3f, \"calls\": %zu}\n",
		             us[us.size() / 2], sum / us.size(), us[(us.size() * 99) / 100], us[0], us.size());
	}
	if (bad) {
		std::fprintf(stderr, "filterBatch disagreed with filter() %d times\n", bad);
		return 1;
	}
	return 0;
}
