// backup_filters.cpp -- single-agent ASIFimplicit / ASIFimplicitTB (host callbacks + GPU QP plug-in)
// next to filterBatch() (everything on the GPU) on the same seeded states.  The model callbacks are the
// reference examples' (examples/InvertedPendulum_Implicit.cpp:31-80, examples/segway_implicit_tb.cpp:27-212);
// the compiled device functors are reused as host functions so both paths see the same model.
//   usage: backup_filters implicit|tb|tbdi N      prints  i,uAct,relax0,relax1,rc,uActBatch,rcBatch
//          backup_filters implicit-out N          prints  Lfh_out_[41], Lgh_out_[41][0], Dh_out_[41][2] after each filter()
//          backup_filters implicit-loop|dii-loop|tbip-loop STEPS [RUN]   the main() loops of InvertedPendulum_Implicit.cpp,
//                                                 DoubleIntegrator_implicit.cpp, InvertedPendulum_ImplicitTB.cpp
//          backup_filters tb-loop STEPS [PUSH]    the closed loop of examples/segway_implicit_tb.cpp:236-275 (pitch rate PUSH at t = 0)
//          backup_filters tbdi-loop STEPS         the closed loop of examples/DoubleIntegrator_implicit_tb.cpp:105-160
//                                                 (fused-gradient constructor, updateOptions at half time); prints
//                                                 i,x0,x1,uAct,relax,TTS,rc,updated with the state the filter was called on
#include <asif++.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "models.hpp"

static double rng(uint64_t seed, uint64_t i, uint64_t j) // splitmix64(seed*2^32 + i*16 + j)
{
	uint64_t z = (seed << 32) + (i * 16 + j);
	z += 0x9e3779b97f4a7c15ULL;
	z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
	z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
	z = z ^ (z >> 31);
	return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

static const asif::DevOptions kNoOpts = {};

template <class M>
struct HostModel {
	static void safetySet(const double *x, double *h, double *Dh)
	{
		double xs[M::NX], hh[M::NPSS], D[M::NPSS * M::NX];
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::safetySet(kNoOpts, xs, hh, D);
		for (int i = 0; i < M::NPSS; i++) h[i] = hh[i];
		for (int i = 0; i < M::NPSS * M::NX; i++) Dh[i] = D[i];
	}
	static void backupSet3(const double *x, double *h, double *Dh)
	{
		double DDh[M::NX * M::NX];
		backupSet4(x, h, Dh, DDh);
	}
	static void backupSet4(const double *x, double *h, double *Dh, double *DDh)
	{
		double xs[M::NX], D[M::NX], DD[M::NX * M::NX], hv;
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::backupSet(kNoOpts, xs, hv, D, DD);
		h[0] = hv;
		for (int i = 0; i < M::NX; i++) Dh[i] = D[i];
		for (int i = 0; i < M::NX * M::NX; i++) DDh[i] = DD[i];
	}
	static void dynamics(const double *x, double *f, double *g)
	{
		double xs[M::NX], ff[M::NX], gg[M::NX];
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::dynamics(kNoOpts, xs, ff, gg);
		for (int i = 0; i < M::NX; i++) { f[i] = ff[i]; g[i] = gg[i]; }
	}
	static void gradients(const double *x, double *Df, double *Dg)
	{
		double xs[M::NX], ff[M::NX], gg[M::NX], A[M::NX * M::NX], Bm[M::NX * M::NX];
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::dynamicsAndGradients(kNoOpts, xs, ff, gg, A, Bm);
		for (int i = 0; i < M::NX * M::NX; i++) { Df[i] = A[i]; Dg[i] = Bm[i]; }
	}
	// the fused callback of the second constructor (include/asif_implicit_tb.h:74-84): f, g and d(f + g u)/dx
	static void dynamicsWithGradient(const double *x, const double *u, double *f, double *g, double *d_fcl_dx)
	{
		static_assert(M::NU == 1, "single-input models");
		double Df[M::NX * M::NX], Dg[M::NX * M::NX];
		dynamics(x, f, g);
		gradients(x, Df, Dg);
		for (int i = 0; i < M::NX * M::NX; i++) d_fcl_dx[i] = Df[i] + Dg[i] * u[0];
	}
	static void controller(const double *x, double *u, double *Du)
	{
		double xs[M::NX], uu[1], D[M::NX];
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::backupController(kNoOpts, xs, uu, D);
		u[0] = uu[0];
		for (int i = 0; i < M::NX; i++) Du[i] = D[i];
	}
};

static QPSOLVER g_solver = QPSOLVER::HIP; // `--solver host` anywhere on the command line: the single-agent QPs on the CPU

int main(int argc, char **argv)
{
	for (int i = 1; i + 1 < argc; i++)
		if (!std::strcmp(argv[i], "--solver")) {
			if (!std::strcmp(argv[i + 1], "host")) g_solver = QPSOLVER::HOST;
			for (int j = i; j + 2 < argc; j++) argv[j] = argv[j + 2];
			argc -= 2;
			break;
		}
	if (argc < 3) return 2;
	const bool tb = !std::strncmp(argv[1], "tb", 2) && std::strncmp(argv[1], "tbdi", 4);
	const long N = std::atol(argv[2]);
	// ---- the closed loops of the three remaining backup-trajectory examples, each as its main() runs it (plant Euler at
	// 1 ms, uDes constant); prints  i,x0,x1,uAct,relax0,relax1,rc,updated  with the state the filter was called on
	if (!std::strcmp(argv[1], "implicit-loop") || !std::strcmp(argv[1], "dii-loop") || !std::strcmp(argv[1], "tbip-loop")) {
		const int run = argc > 3 ? std::atoi(argv[3]) : 0;
		const double dt = 0.001, tEnd = dt * (double)N;
		std::printf("i,x0,x1,uAct,relax0,relax1,rc,updated\n");
		auto plant = [&](auto dyn, double (&x)[2], double ua) {
			double f[2], g[2];
			dyn(x, f, g);
			for (int k = 0; k < 2; k++) x[k] += dt * (f[k] + g[k] * ua);
		};
		if (!std::strcmp(argv[1], "implicit-loop")) { // examples/InvertedPendulum_Implicit.cpp:84-140, run `run` of its ten
			typedef HostModel<asif::InvertedPendulum> H;
			const double lb[1] = {-1.5}, ub[1] = {1.5}, ud[1] = {0.0};
			ASIF::ASIFimplicit::Options opts;
			opts.backTrajHorizon = 5.0;
			opts.backTrajDt = 0.001;
			opts.relaxReachLb = 5.0;
			opts.relaxSafeLb = 10.0;
			ASIF::ASIFimplicit flt(2, 1, 4, 1, 10, H::safetySet, H::backupSet3, H::dynamics, H::gradients, H::controller, g_solver);
			if (flt.initialize(lb, ub, opts) != 1) return 3;
			double x[2] = {0.1 + (double)run * 0.29, 0.0};
			for (long i = 0; i < N; i++) {
				double ua[1] = {0.0}, rl[2] = {0.0, 0.0};
				const int32_t rc = flt.filter(x, ud, ua, rl);
				std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%d,0\n", i, x[0], x[1], ua[0], rl[0], rl[1], rc);
				plant(H::dynamics, x, ua[0]);
			}
		} else if (!std::strcmp(argv[1], "dii-loop")) { // examples/DoubleIntegrator_implicit.cpp:87-150: fused-gradient constructor,
			typedef HostModel<asif::DoubleIntegratorImplicit> H; // updateOptions(backTrajHorizon = 5) at half time
			const double lb[1] = {-1.0}, ub[1] = {1.0}, ud[1] = {1.0};
			ASIF::ASIFimplicit::Options opts;
			opts.backTrajHorizon = 2.0;
			opts.backTrajDt = 0.01;
			opts.relaxReachLb = 5.0;
			opts.relaxSafeLb = 10.0;
			ASIF::ASIFimplicit flt(2, 1, 4, 1, 4, H::safetySet, H::backupSet3, H::dynamicsWithGradient, H::controller, g_solver);
			if (flt.initialize(lb, ub, opts) != 1) return 3;
			opts.backTrajHorizon = 5.0;
			double x[2] = {0.0, 0.0}, t = 0.0;
			bool updated = false;
			for (long i = 0; i < N; i++) {
				if (!updated && t > tEnd / 2) {
					updated = true;
					if (flt.updateOptions(opts) != 1) return 5;
				}
				double ua[1] = {0.0}, rl[2] = {0.0, 0.0};
				const int32_t rc = flt.filter(x, ud, ua, rl);
				std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%d\n", i, x[0], x[1], ua[0], rl[0], rl[1], rc, (int)updated);
				plant(H::dynamics, x, ua[0]);
				t += dt;
			}
		} else { // examples/InvertedPendulum_ImplicitTB.cpp:100-175, run `run` of its two
			typedef HostModel<asif::InvertedPendulumTB> H;
			const double lb[1] = {-1.5}, ub[1] = {1.5}, ud[1] = {0.0};
			ASIF::ASIFimplicitTB::Options opts;
			opts.backTrajHorizon = 11.0;
			opts.backTrajDt = 0.001;
			opts.relaxCost = 10.;
			opts.relaxSafeLb = 10.0;
			opts.relaxTTS = 30.0;
			opts.relaxMinOrtho = 60.0;
			opts.backTrajMinOrtho = 0.001;
			ASIF::ASIFimplicitTB flt(2, 1, 4, 4, H::safetySet, H::backupSet4, H::dynamics, H::gradients, H::controller, g_solver);
			if (flt.initialize(lb, ub, opts) != 1) return 3;
			double x[2] = {-0.1 + (double)run * 0.2, 0.0};
			for (long i = 0; i < N; i++) {
				double ua[1] = {0.0}, rl = 0.0;
				const int32_t rc = flt.filter(x, ud, ua, rl);
				std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%d,0\n", i, x[0], x[1], ua[0], rl, flt.TTS_, rc);
				plant(H::dynamics, x, ua[0]);
			}
		}
		return 0;
	}
	if (!std::strncmp(argv[1], "tbdi", 4)) {
		typedef HostModel<asif::DoubleIntegratorTB> H;
		const double lb[1] = {-1.0}, ub[1] = {1.0};
		ASIF::ASIFimplicitTB::Options opts; // examples/DoubleIntegrator_implicit_tb.cpp:107-112
		opts.backTrajHorizon = 2.0;
		opts.backTrajDt = 0.001;
		opts.relaxSafeLb = 10.0;
		opts.relaxTTS = 5.0;
		opts.relaxMinOrtho = 5.0;
		ASIF::ASIFimplicitTB flt(2, 1, 4, 4, H::safetySet, H::backupSet4, H::dynamicsWithGradient, H::controller, g_solver);
		// (under `--solver host` the loops run without a device: nothing is bound and the batched last line is left out)
		const bool noDevice = g_solver == QPSOLVER::HOST;
		if (flt.initialize(lb, ub, opts) != 1 || (!noDevice && flt.bindDeviceModel(ASIF_HIP_MODEL_DOUBLE_INTEGRATOR_TB) != 0)) return 3;
		if (!std::strcmp(argv[1], "tbdi-loop")) {
			std::printf("i,x0,x1,uAct,relax,TTS,rc,updated\n");
			const double dt = 0.001, tEnd = dt * (double)N; // the example runs N = 5000
			double x[2] = {0.1, 0.1}, t = 0.0;
			const double ud[1] = {0.9};
			bool updated = false;
			for (long i = 0; i < N; i++) {
				if (!updated && t > tEnd / 2) {
					opts.backTrajHorizon = 7.0;
					updated = true;
					if (flt.updateOptions(opts) != 1) return 5;
				}
				double ua[1] = {0.0}, rl = 0.0;
				const int32_t rc = flt.filter(x, ud, ua, rl);
				std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%d\n", i, x[0], x[1], ua[0], rl, flt.TTS_, rc, (int)updated);
				double f[2], g[2];
				H::dynamics(x, f, g);
				for (int k = 0; k < 2; k++) x[k] += dt * (f[k] + g[k] * ua[0]);
				t += dt;
			}
			if (noDevice) return 0;
			// the batched path after updateOptions(): the last state, as a batch of one
			double ua[1] = {0.0}, rl[1] = {0.0};
			int32_t rc = 0;
			if (flt.filterBatch(1, x, ud, ua, rl, &rc) != 0) return 4;
			std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%d\n", N, x[0], x[1], ua[0], rl[0], 0.0, rc, 2);
			return 0;
		}
		std::printf("i,uAct,relax0,relax1,rc,uActBatch,rcBatch\n");
		std::vector<double> bx(2 * N), bu(N), ba(N, 0.0), br(N, 0.0);
		std::vector<int32_t> brc(N, 0);
		for (long i = 0; i < N; i++) {
			bx[i] = -0.04 + 0.08 * rng(12, i, 0);
			bx[N + i] = -0.04 + 0.08 * rng(12, i, 1);
			bu[i] = -1.5 + 3.0 * rng(12, i, 2);
		}
		if (flt.filterBatch(N, bx.data(), bu.data(), ba.data(), br.data(), brc.data()) != 0) return 4;
		for (long i = 0; i < N; i++) {
			const double x[2] = {bx[i], bx[N + i]}, ud[1] = {bu[i]};
			double ua[1] = {0.0}, rl = 0.0;
			const int32_t rc = flt.filter(x, ud, ua, rl);
			std::printf("%ld,%.17g,%.17g,%.17g,%d,%.17g,%d\n", i, ua[0], rl, flt.TTS_, rc, ba[i], brc[i]);
		}
		return 0;
	}
	if (std::strcmp(argv[1], "tb-loop")) std::printf("i,uAct,relax0,relax1,rc,uActBatch,rcBatch\n");
	if (!tb) {
		typedef HostModel<asif::InvertedPendulum> H;
		const double lb[1] = {-1.5}, ub[1] = {1.5};
		ASIF::ASIFimplicit::Options opts; // examples/InvertedPendulum_Implicit.cpp:93-97
		opts.backTrajHorizon = 5.0;
		opts.backTrajDt = 0.001;
		opts.relaxReachLb = 5.0;
		opts.relaxSafeLb = 10.0;
		ASIF::ASIFimplicit flt(2, 1, 4, 1, 10, H::safetySet, H::backupSet3, H::dynamics, H::gradients, H::controller, g_solver);
		if (flt.initialize(lb, ub, opts) != 1 || flt.bindDeviceModel(ASIF_HIP_MODEL_INVERTED_PENDULUM) != 0) return 3;
		std::vector<double> bx(2 * N), bu(N), ba(N, 0.0), br(2 * N, 0.0);
		std::vector<int32_t> brc(N, 0);
		for (long i = 0; i < N; i++) {
			bx[i] = -1.5 + 3.0 * rng(2, i, 0);
			bx[N + i] = -1.5 + 3.0 * rng(2, i, 1);
			bu[i] = -1.5 + 3.0 * rng(2, i, 2);
		}
		if (!std::strcmp(argv[1], "implicit-out")) {
			// the class's public Lfh_out_ / Lgh_out_ / Dh_out_ (include/asif_implicit.h:122-124) after each filter():
			// one line of 41 + 41 + 82 numbers per instance
			for (long i = 0; i < N; i++) {
				const double x[2] = {bx[i], bx[N + i]}, ud[1] = {bu[i]};
				double ua[1] = {0.0};
				(void)flt.filter(x, ud, ua);
				if (flt.Lfh_out_.size() != 41 || flt.Lgh_out_.size() != 41 || flt.Dh_out_.size() != 41) return 6;
				for (int r = 0; r < 41; r++) std::printf("%.17g,", flt.Lfh_out_[r]);
				for (int r = 0; r < 41; r++) std::printf("%.17g,", flt.Lgh_out_[r][0]);
				for (int r = 0; r < 41; r++) std::printf("%.17g,%.17g%s", flt.Dh_out_[r][0], flt.Dh_out_[r][1], r == 40 ? "\n" : ",");
			}
			return 0;
		}
		if (flt.filterBatch(N, bx.data(), bu.data(), ba.data(), br.data(), brc.data()) != 0) return 4;
		for (long i = 0; i < N; i++) {
			const double x[2] = {bx[i], bx[N + i]}, ud[1] = {bu[i]};
			double ua[1] = {0.0}, rl[2] = {0.0, 0.0};
			const int32_t rc = flt.filter(x, ud, ua, rl);
			std::printf("%ld,%.17g,%.17g,%.17g,%d,%.17g,%d\n", i, ua[0], rl[0], rl[1], rc, ba[i], brc[i]);
		}
	} else {
		typedef HostModel<asif::Segway> H;
		const double lb[1] = {-20.0}, ub[1] = {20.0};
		ASIF::ASIFimplicitTB::Options opts; // examples/segway_implicit_tb.cpp:223-230
		opts.backTrajHorizon = 3.0;
		opts.backTrajDt = 0.01;
		opts.relaxCost = 10;
		opts.relaxSafeLb = 2.0;
		opts.relaxTTS = 30.0;
		opts.relaxMinOrtho = 60.0;
		opts.backTrajMinOrtho = 0.001;
		ASIF::ASIFimplicitTB flt(4, 1, 4, 4, H::safetySet, H::backupSet4, H::dynamics, H::gradients, H::controller, g_solver);
		if (flt.initialize(lb, ub, opts) != 1 || (g_solver != QPSOLVER::HOST && flt.bindDeviceModel(ASIF_HIP_MODEL_SEGWAY) != 0)) return 3;
		if (!std::strcmp(argv[1], "tb-loop")) {
			// the closed loop of examples/segway_implicit_tb.cpp:236-275: from rest with uDes = 0, plant Euler at 1 ms,
			// updateOptions(backTrajHorizon = 6) once t > tEnd / 2.  x0 is given a push (argv[3], pitch rate) so that the
			// loop leaves the backup set and the filter has something to do; 0 reproduces the example literally.
			const double dt = 0.001, tEnd = dt * (double)N; // the example runs to 10.99 s
			double x[4] = {0.0, 0.0, 0.0, argc > 3 ? std::atof(argv[3]) : 0.0}, t = 0.0;
			const double ud[1] = {0.0};
			bool updated = false;
			std::printf("i,x0,x1,x2,x3,uAct,relax,TTS,rc,updated\n");
			for (long i = 0; i < N; i++) {
				if (!updated && t > tEnd / 2) {
					opts.backTrajHorizon = 6.0;
					updated = true;
					if (flt.updateOptions(opts) != 1) return 5;
				}
				double ua[1] = {0.0}, rl = 0.0;
				const int32_t rc = flt.filter(x, ud, ua, rl);
				std::printf("%ld,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%d\n", i, x[0], x[1], x[2], x[3], ua[0], rl, flt.TTS_, rc,
				            (int)updated);
				double f[4], g[4];
				H::dynamics(x, f, g);
				for (int k = 0; k < 4; k++) x[k] += dt * (f[k] + g[k] * ua[0]);
				t += dt;
			}
			return 0;
		}
		const double xb[4] = {3.0, 3.0, M_PI / 6, M_PI};
		std::vector<double> bx(4 * N), bu(N), ba(N, 0.0), br(N, 0.0);
		std::vector<int32_t> brc(N, 0);
		for (long i = 0; i < N; i++) {
			for (int j = 0; j < 4; j++) bx[j * N + i] = 0.05 * xb[j] * (2.0 * rng(3, i, j) - 1.0);
			bu[i] = -5.0 + 10.0 * rng(3, i, 4);
		}
		if (flt.filterBatch(N, bx.data(), bu.data(), ba.data(), br.data(), brc.data()) != 0) return 4;
		for (long i = 0; i < N; i++) {
			const double x[4] = {bx[i], bx[N + i], bx[2 * N + i], bx[3 * N + i]}, ud[1] = {bu[i]};
			double ua[1] = {0.0}, rl = 0.0;
			const int32_t rc = flt.filter(x, ud, ua, rl);
			std::printf("%ld,%.17g,%.17g,%.17g,%d,%.17g,%d\n", i, ua[0], rl, flt.TTS_, rc, ba[i], brc[i]);
		}
	}
	return 0;
}
