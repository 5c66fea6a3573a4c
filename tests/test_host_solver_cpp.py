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
