// explicit_variants.cpp -- class ASIF::ASIF beyond the shipped example: (a) a model with two inputs
// (ASIF_HIP_MODEL_PLANAR_TWO_INPUT), (b) the double integrator with npSSmax = 2 of its 4 safety functions kept per
// call (src/asif.cpp:250-268).  For each: construct, initialize(), bindDeviceModel(), filter() per state against
// filterBatch() over the same states; then updateOptions() with another relaxation cost / bound and again.
// filter() assembles the rows on the host and solves on the GPU through QPWrapperHip; filterBatch() runs the fused
// kernel of the bound model -- they must describe the same filter before AND after updateOptions().
// Prints one CSV row per state and phase:  case,phase,i,rcSingle,rcBatch,uSingle...,uBatch...
#include <asif++.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// ---- (a) the synthetic two-input model of asif_amd/csrc/models.hpp (PlanarTwoInput), as host callbacks
static void p2SafetySet(const double *x, double *h, double *Dh)
{
	const double a[5][2] = {{1., 0.}, {-1., 0.}, {0., 1.}, {0., -1.}, {0.6, 0.8}}, r[5] = {1., 1., 1., 1., 1.2};
	for (int i = 0; i < 5; i++) {
		h[i] = r[i] - a[i][0] * x[0] - a[i][1] * x[1];
		Dh[i] = -a[i][0];
		Dh[i + 5] = -a[i][1];
	}
}
static void p2Dynamics(const double *x, double *f, double *g)
{
	f[0] = -0.5 * x[0] + 0.2 * x[1];
	f[1] = 0.1 * x[0] + -0.3 * x[1];
	g[0] = 1.0; g[1] = 0.0;
	g[2] = 0.3; g[3] = 1.0;
}
// ---- (b) examples/DoubleIntegrator.cpp:24-61
static void diSafetySet(const double *x, double *h, double *Dh)
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
static void diDynamics(const double *x, double *f, double *g)
{
	f[0] = x[1];
	f[1] = 0.0;
	g[0] = 0.0;
	g[1] = 1.0;
}

static double unit(uint64_t k) // splitmix64 -> [0,1)
{
	uint64_t z = k + 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	z ^= z >> 31;
	return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// `--solver host`: the objects are built with QPSOLVER::HOST, nothing is bound to a device, and every line carries the
// state and the desired input after the single-agent answer (case,phase,i,rc,u...,relax,x...,uDes...) for
// tests/test_host_solver_cpp.py to put through the oracle
static QPSOLVER g_solver = QPSOLVER::HIP;

static int run_case(const char *name, ASIF::ASIF &flt, int model, uint32_t nx, uint32_t nu, long n, double scale)
{
	const bool host = g_solver == QPSOLVER::HOST;
	std::vector<double> lb(nu, -1.0), ub(nu, 1.0);
	int32_t r = flt.initialize(lb.data(), ub.data());
	if (r != 1) { std::fprintf(stderr, "%s: initialize %d\n", name, r); return 2; }
	r = host ? 0 : flt.bindDeviceModel(model);
	if (r != 0) { std::fprintf(stderr, "%s: bindDeviceModel %d (%s)\n", name, r, asif_hip_error_string(r)); return 2; }
	std::vector<double> x(nx * n), ud(nu * n);
	for (long i = 0; i < n; i++) {
		for (uint32_t k = 0; k < nx; k++) x[k * n + i] = scale * (2.0 * unit(i * 16 + k) - 1.0);
		for (uint32_t j = 0; j < nu; j++) ud[j * n + i] = 1.5 * (2.0 * unit(i * 16 + 8 + j) - 1.0);
	}
	for (int phase = 0; phase < 2; phase++) {
		if (phase == 1) {
			ASIF::ASIF::Options o;
			o.relaxCost = 20.0;
			o.relaxLb = 2.0;
			// updateOptions() moves only the LOWER bound of the pinned relaxation variable (src/asif.cpp:227-228, the
			// upper one stays where initialize() put it), while the batched path pins it at the new value like a fresh
			// object.  To compare the two on the same filter the object is re-initialised with the new options first
			// (both bounds move) and THEN updateOptions() pushes them to the bound device model -- the step under test:
			// it must rebuild the device options for the model and row budget this object was bound with.
			r = flt.initialize(lb.data(), ub.data(), o);
			if (r != 1) return 2;
			r = flt.updateOptions(o);
			if (r != 1) { std::fprintf(stderr, "%s: updateOptions %d\n", name, r); return 2; }
		}
		std::vector<double> ua(nu * n, 7.0), rl(n, -7.0);
		std::vector<int32_t> rc(n, 0);
		r = host ? 0 : flt.filterBatch(n, x.data(), ud.data(), ua.data(), rl.data(), rc.data());
		if (r != 0) { std::fprintf(stderr, "%s: filterBatch %d (%s)\n", name, r, asif_hip_error_string(r)); return 2; }
		for (long i = 0; i < n; i++) {
			double xs[4], us[2], u1[2] = {7.0, 7.0}, relax = -7.0;
			for (uint32_t k = 0; k < nx; k++) xs[k] = x[k * n + i];
			for (uint32_t j = 0; j < nu; j++) us[j] = ud[j * n + i];
			const int32_t r1 = flt.filter(xs, us, u1, relax);
			if (host) {
				std::printf("%s,%d,%ld,%d", name, phase, i, r1);
				for (uint32_t j = 0; j < nu; j++) std::printf(",%.17g", u1[j]);
				std::printf(",%.17g", relax);
				for (uint32_t k = 0; k < nx; k++) std::printf(",%.17g", xs[k]);
				for (uint32_t j = 0; j < nu; j++) std::printf(",%.17g", us[j]);
				std::printf("\n");
				continue;
			}
			std::printf("%s,%d,%ld,%d,%d", name, phase, i, r1, rc[i]);
			for (uint32_t j = 0; j < nu; j++) std::printf(",%.17g", u1[j]);
			for (uint32_t j = 0; j < nu; j++) std::printf(",%.17g", ua[j * n + i]);
			std::printf(",%.17g,%.17g\n", relax, rl[i]);
		}
	}
	return 0;
}

int main(int argc, char **argv)
{
	for (int i = 1; i + 1 < argc; i++)
		if (!std::strcmp(argv[i], "--solver")) {
			if (!std::strcmp(argv[i + 1], "host")) g_solver = QPSOLVER::HOST;
			for (int j = i; j + 2 < argc; j++) argv[j] = argv[j + 2];
			argc -= 2;
			break;
		}
	const long n = argc > 1 ? std::atol(argv[1]) : 64;
	std::printf("case,phase,i,rcSingle,rcBatch,u...\n");
	{
		ASIF::ASIF flt(2, 2, 5, p2SafetySet, p2Dynamics, -1, g_solver);
		if (int r = run_case("planar2", flt, ASIF_HIP_MODEL_PLANAR_TWO_INPUT, 2, 2, n, 1.3)) return r;
	}
	{
		ASIF::ASIF flt(2, 1, 4, diSafetySet, diDynamics, /*npSSmax=*/2, g_solver);
		if (int r = run_case("di_keep2", flt, ASIF_HIP_MODEL_DOUBLE_INTEGRATOR, 2, 1, n, 1.2)) return r;
	}
	return 0;
}
