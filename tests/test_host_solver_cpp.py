"""`QPSOLVER::HOST` (ASIF::QPWrapperHost, asif_amd/host/qpwrapper_host.cpp): the single-agent filter() of the mirror's
classes with the QP solved on the calling thread by the product's dual active-set method (asif_amd/csrc/gi_small.hpp,
host build) -- BASELINE config 1 as written ("single agent, CPU path, no GPU").  No device anywhere in these programs:
they run here, in the GPU-less container, and go through the same step-by-step checks against the oracle as their
QPWrapperHip twins on the GPU box (tests/test_gpu_host_cpp.py): on the state the program was in, every step's input is
the exact optimum of the QP the reference assembles, return codes identical, the plant step the example's.

The host solver is selected by name only.  That the default solver still fails loudly without a device is
tests/test_reference_examples_link.py::test_linked_example_fails_loudly_without_a_gpu and tests/test_capi_host.py."""
import os
import subprocess

import numpy as np
import pytest

import test_gpu_host_cpp as gpu_twin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "asif_amd", "host")


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__
    if not os.path.exists(os.path.join(HOST, "double_integrator")) or not os.path.exists(os.path.join(HOST, "backup_filters")):
        __graft_entry__.build()


def test_c1_double_integrator_closed_loop_on_the_host_solver(oracle):
    gpu_twin.check_double_integrator_closed_loop(oracle, "host")


@pytest.mark.parametrize("kind,cfg,steps,run", [("implicit-loop", 3, 120, 5), ("dii-loop", 9, 1200, 0), ("tbip-loop", 8, 60, 1)])
def test_backup_trajectory_example_loops_on_the_host_solver(oracle, kind, cfg, steps, run):
    gpu_twin.check_example_loop_step_by_step(oracle, kind, cfg, steps, run, "host")


def test_segway_tb_closed_loop_on_the_host_solver(oracle):
    gpu_twin.check_segway_tb_closed_loop(oracle, "host")


def test_double_integrator_tb_closed_loop_on_the_host_solver(oracle):
    gpu_twin.check_double_integrator_tb_closed_loop(oracle, "host", steps=1200)


@pytest.mark.parametrize("plain", [False, True])
def test_implicit_rb_class_and_learned_residual_on_the_host_solver(oracle, plain):
    """ASIF::ASIFimplicitRB (held backup input, interval margins on host AAF operands, the two residual networks) and
    ASIF::ASIFimplicit with use_learning, single agent, 3 x 41 QP on QPWrapperHost: exact optimum, rc and the class's
    public diagnostics (Dh_index_, learning_data_.Lfh_diff / Lgh_diff) against the oracle -- no device."""
    gpu_twin.check_implicit_rb_class(oracle, plain, "host")


def test_robust_class_on_the_host_solver(oracle):
    """ASIF::ASIFrobust, single agent: affine-arithmetic rows bit-identical to the oracle's, the full 18 x 12 problem --
    multipliers and all, as src/asif_robust.cpp hands it to its solver -- on QPWrapperHost's Newton stage
    (asif_amd/host/qp_alm_host.cpp): every return code, input and relaxation against the exact optimum."""
    gpu_twin.check_robust_class(oracle, "host")


@pytest.mark.parametrize("p,ud,steps", [(0.8, 0.0, 280), (1.0, 1.5, 155), (1.2, -1.5, 210)])
def test_robust_pendulum_closed_loop_on_the_host_solver(oracle, p, ud, steps):
    gpu_twin.check_robust_pendulum_closed_loop(oracle, p, ud, steps, "host")


def test_robust_class_on_shipped_data_on_the_host_solver(oracle, tmp_path):
    """examples/DoubleIntegrator_Robust.cpp's filter (22 x 15, the shipped half-planes): infeasible instances included."""
    gpu_twin.check_robust_class_on_shipped_data(oracle, tmp_path, "host")


def test_double_integrator_robust_closed_loop_on_the_host_solver(oracle, tmp_path):
    gpu_twin.check_double_integrator_robust_closed_loop(oracle, tmp_path, "host")


def test_realizable_class_on_the_host_solver(oracle, tmp_path):
    """ASIF::ASIFrealizable, single agent: the facet tests (2 x 5) on the active-set stage, the lifted 38 x 29 problem on
    the Newton stage; critical-facet counts, rows, return codes 1 / -2 and inputs against the oracle."""
    gpu_twin.check_realizable_class(oracle, tmp_path, "host")


def test_realizable_sampled_closed_loop_on_the_host_solver(oracle, tmp_path):
    gpu_twin.check_realizable_sampled_closed_loop(oracle, tmp_path, "host")


def test_explicit_class_two_inputs_and_reduced_row_budget_on_the_host_solver(oracle):
    """class ASIF beyond the shipped example, on QPWrapperHost: a model with two inputs (nv = 3, five rows) and the double
    integrator with npSSmax = 2 of its 4 rows (src/asif.cpp:250-268), before and after initialize(options) +
    updateOptions(): rc and uAct of every call against the oracle's exact optimum on the same state; uAct and relax
    untouched where the QP is infeasible (src/asif.cpp:208-209)."""
    exe = os.path.join(HOST, "explicit_variants")
    n = 96
    out = subprocess.run([exe, str(n), "--solver", "host"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l.split(",") for l in out.stdout.strip().split("\n")[1:]]
    failed = 0
    for name, cfg, nu, keep in (("planar2", 11, 2, 0), ("di_keep2", 2, 1, 2)):
        model, variant = oracle.CONFIGS[cfg]
        for phase in (0, 1):
            rows = np.array([[float(v) for v in l[2:]] for l in lines if l[0] == name and int(l[1]) == phase])
            assert rows.shape == (n, 2 + nu + 1 + 2 + nu), (name, phase, rows.shape)
            rc1, u1, relax = rows[:, 1].astype(int), rows[:, 2:2 + nu], rows[:, 2 + nu]
            x, ud = np.ascontiguousarray(rows[:, 3 + nu:5 + nu]), np.ascontiguousarray(rows[:, 5 + nu:])
            o = oracle.default_options(model, variant)
            if keep:
                o.npSSmax = keep
            if phase == 1:
                o.relaxCost, o.relaxLb = 20.0, 2.0
            ua, rl, rc = oracle.filter_batch(model, variant, o, x, ud, oracle.SOLVER_EXACT, uact_init=np.full((n, nu), 7.0))
            assert np.array_equal(rc1, rc), (name, phase, np.where(rc1 != rc)[0][:8])
            ok = rc == 1
            assert ok.sum() >= n // 4
            failed += int((~ok).sum())
            assert np.abs(u1[ok] - ua[ok]).max() <= 1e-6, (name, phase)
            assert np.all(u1[~ok] == 7.0) and np.all(relax[~ok] == -7.0)
            assert np.abs(relax[ok] - (5.0 if phase == 0 else 2.0)).max() <= 1e-9  # the pinned relaxation variable
    assert failed >= 1  # the infeasible branch was exercised somewhere


def test_host_solver_is_opt_in_and_bounded_by_shape(tmp_path):
    """makeQPWrapper: HOST gives QPWrapperHost up to 128 x 128 (the robust and realizable classes' lifted problems and
    full cost matrices on its Newton stage), QPWrapperHip beyond that and under the default name; QPWrapperHost refuses
    larger shapes."""
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include <asif++.h>
#include <cstdio>
#include <vector>
int main() {
	using namespace ASIF;
	QPWrapperAbstract *a = makeQPWrapper(QPSOLVER::HOST, 2, 4, true), *b = makeQPWrapper(QPSOLVER::HOST, 18, 12, true),
	                  *c = makeQPWrapper(QPSOLVER::HOST, 3, 41, false), *d = makeQPWrapper(QPSOLVER::OSQP, 2, 4, true),
	                  *e = makeQPWrapper(QPSOLVER::HOST, 129, 4, true), *f = makeQPWrapper(QPSOLVER::OSQP, 18, 12, true);
	std::printf("%d %d %d %d %d %d\n", dynamic_cast<QPWrapperHost *>(a) != nullptr, dynamic_cast<QPWrapperHost *>(b) != nullptr,
	            dynamic_cast<QPWrapperHost *>(c) != nullptr, dynamic_cast<QPWrapperHip *>(d) != nullptr,
	            dynamic_cast<QPWrapperHip *>(e) != nullptr, dynamic_cast<QPWrapperHip *>(f) != nullptr);
	QPWrapperHost big(129, 1, true);
	std::vector<double> H(129 * 129, 0.0), z(129, 0.0);
	std::printf("%d\n", big.initialize(H.data(), z.data(), z.data(), z.data(), z.data(), z.data()));
	// full cost matrix on the Newton stage: min x'Hx + c'x, H = [[2,1],[1,3]] (upper triangle read), x1 + x2 >= 1, 0 <= x <= 0.9
	{
		QPWrapperHost w(2, 1, false);
		const double Hf[4] = {2, 99, 1, 3}, cf[2] = {-2, -6}, Af[2] = {1, 1}, bf[1] = {1}, l[2] = {0, 0}, u[2] = {0.9, 0.9};
		double x[2];
		const int r = w.initialize(Hf, cf, Af, bf, l, u), st = w.solve();
		w.getSolution(x);
		std::printf("%d %d %.15g %.15g\n", r, st, x[0], x[1]);
	}
	// a 2 x 4 problem of class ASIF: min (u - 1)^2 + 50 (d - 5)^2, u <= 0.25 by the first row, d pinned at 5
	const double H2[4] = {1, 0, 0, 50}, c2[2] = {-2, -500}, A2[8] = {-1, 0, 0, 0, 0, 0, 0, 0}, b2[4] = {-0.25, -1e20, -1e20, -1e20},
	             lb2[2] = {-1, 5}, ub2[2] = {1, 5};
	double sol[2];
	const int r0 = a->initialize(H2, c2, A2, b2, lb2, ub2), r1 = a->solve();
	a->getSolution(sol);
	std::printf("%d %d %.17g %.17g\n", r0, r1, sol[0], sol[1]);
	const double b3[4] = {2.0, -1e20, -1e20, -1e20}; // -u >= 2 against u >= -1: infeasible
	a->updateb(b3);
	std::printf("%d\n", a->solve());
	return 0;
}
""")
    exe = tmp_path / "t"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(HOST, "include"), "-I", os.path.join(ROOT, "include"),
                           str(src), os.path.join(HOST, "libasif_host.a"), "-L" + os.path.join(ROOT, "asif_amd"), "-lasif_hip",
                           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "asif_amd"),
                           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.split("\n")
    assert out[0] == "1 1 1 1 1 1"
    assert int(out[1]) < 0  # ASIF_HIP_EUNSUPPORTED
    r, st, x0, x1 = out[2].split()  # unconstrained minimiser (0, 1) clipped: x1 = 0.9, then 4 x0 + 2 x1 - 2 = 0 -> 0.05; row: 0.1
    assert (int(r), int(st)) == (0, 1) and abs(float(x0) - 0.1) <= 1e-7 and abs(float(x1) - 0.9) <= 1e-7
    r0, r1, u, d = out[3].split()
    assert (int(r0), int(r1)) == (0, 1) and float(u) == 0.25 and float(d) == 5.0
    assert int(out[4]) == -3  # OSQP's primal-infeasible value, raw (src/qpwrapper_osqp.cpp:225-238)
