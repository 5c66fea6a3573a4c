// backup_filters.cpp -- single-agent ASIFimplicit / ASIFimplicitTB (host callbacks + GPU QP plug-in)
// next to filterBatch() (everything on the GPU) on the same seeded states.  The model callbacks are the
// reference examples' (examples/InvertedPendulum_Implicit.cpp:31-80, examples/segway_implicit_tb.cpp:27-212);
// the compiled device functors are reused as host functions so both paths see the same model.
//   usage: backup_filters implicit|tb N      prints  i,uAct,relax0,relax1,rc,uActBatch,rcBatch
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
	static void controller(const double *x, double *u, double *Du)
	{
		double xs[M::NX], uu[1], D[M::NX];
		for (int i = 0; i < M::NX; i++) xs[i] = x[i];
		M::backupController(kNoOpts, xs, uu, D);
		u[0] = uu[0];
		for (int i = 0; i < M::NX; i++) Du[i] = D[i];
	}
};

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	const bool tb = !std::strcmp(argv[1], "tb");
	const long N = std::atol(argv[2]);
	std::printf("i,uAct,relax0,relax1,rc,uActBatch,rcBatch\n");
	if (!tb) {
		typedef HostModel<asif::InvertedPendulum> H;
		const double lb[1] = {-1.5}, ub[1] = {1.5};
		ASIF::ASIFimplicit::Options opts; // examples/InvertedPendulum_Implicit.cpp:93-97
		opts.backTrajHorizon = 5.0;
		opts.backTrajDt = 0.001;
		opts.relaxReachLb = 5.0;
		opts.relaxSafeLb = 10.0;
		ASIF::ASIFimplicit flt(2, 1, 4, 1, 10, H::safetySet, H::backupSet3, H::dynamics, H::gradients, H::controller);
		if (flt.initialize(lb, ub, opts) != 1 || flt.bindDeviceModel(ASIF_HIP_MODEL_INVERTED_PENDULUM) != 0) return 3;
		std::vector<double> bx(2 * N), bu(N), ba(N, 0.0), br(2 * N, 0.0);
		std::vector<int32_t> brc(N, 0);
		for (long i = 0; i < N; i++) {
			bx[i] = -1.5 + 3.0 * rng(2, i, 0);
			bx[N + i] = -1.5 + 3.0 * rng(2, i, 1);
			bu[i] = -1.5 + 3.0 * rng(2, i, 2);
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
		ASIF::ASIFimplicitTB flt(4, 1, 4, 4, H::safetySet, H::backupSet4, H::dynamics, H::gradients, H::controller);
		if (flt.initialize(lb, ub, opts) != 1 || flt.bindDeviceModel(ASIF_HIP_MODEL_SEGWAY) != 0) return 3;
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
