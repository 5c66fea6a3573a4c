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
    """makeQPWrapper: HOST gives QPWrapperHost for nv <= 3 with a diagonal cost, QPWrapperHip for anything else (the
    robust and realizable classes' lifted problems) and under the default name; QPWrapperHost refuses other shapes."""
    src = tmp_path / "t.cpp"
    src.write_text(r"""
#include <asif++.h>
#include <cstdio>
int main() {
	using namespace ASIF;
	QPWrapperAbstract *a = makeQPWrapper(QPSOLVER::HOST, 2, 4, true), *b = makeQPWrapper(QPSOLVER::HOST, 18, 12, true),
	                  *c = makeQPWrapper(QPSOLVER::HOST, 3, 41, false), *d = makeQPWrapper(QPSOLVER::OSQP, 2, 4, true);
	std::printf("%d %d %d %d\n", dynamic_cast<QPWrapperHost *>(a) != nullptr, dynamic_cast<QPWrapperHip *>(b) != nullptr,
	            dynamic_cast<QPWrapperHip *>(c) != nullptr, dynamic_cast<QPWrapperHip *>(d) != nullptr);
	QPWrapperHost big(4, 4, true);
	const double H[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, z[16] = {0};
	std::printf("%d\n", big.initialize(H, z, z, z, z, z));
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
    assert out[0] == "1 1 1 1"
    assert int(out[1]) < 0  # ASIF_HIP_EUNSUPPORTED
    r0, r1, u, d = out[2].split()
    assert (int(r0), int(r1)) == (0, 1) and float(u) == 0.25 and float(d) == 5.0
    assert int(out[3]) == -3  # OSQP's primal-infeasible value, raw (src/qpwrapper_osqp.cpp:225-238)
