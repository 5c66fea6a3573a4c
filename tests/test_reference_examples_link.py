"""The drop-in claim, tested: the reference's own example programs compile and link UNCHANGED with this repo's
`asif_amd/host/include` put in the place of the reference's `include/` directory and `libasif_host.a` +
`libasif_hip.so` in the place of `libasif++` + OSQP (north star: "keeping the ASIF::filter() / qpwrapper_abstract C++
API so it drops into the existing examples").

The sources are read where they lie under /root/reference at test time and the executables go to a temporary
directory: nothing of the reference is copied into the repo or travels to the GPU box (the test skips where the
reference is absent).  Of the reference's include/ directory only its DATA headers (the shipped kernels and half-plane
sets) and the timer its examples print with are put on the include path, by name, so that no API header of the
reference can stand in for a missing one of the mirror.

Out: `DoubleIntegrator_implicit`, `DoubleIntegrator_implicit_tb`, `segway_implicit_tb` include a `customTimer.h` the
reference itself does not ship (`examples/segway_implicit_tb.cpp:9`): they do not compile in the reference tree either.
Their `main()` loops run through the mirror's classes in tests/test_gpu_host_cpp.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HOST = os.path.join(ROOT, "asif_amd", "host")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

# the reference's data headers + the timer class its examples use; everything else must come from the mirror
DATA_HEADERS = ["CyberTimer.hpp", "KernelData_70-135kg.h", "KernelData_70-75kg.h", "RealizableKernelData_100Hz.h",
                "RealizableKernelData_100Hz_50pt.h", "RealizableKernelData_10Hz.h", "RealizableKernelData_10Hz_50pt.h"]

EXAMPLES = ["DoubleIntegrator", "DoubleIntegrator_Robust", "DoubleIntegrator_RealizableSampled",
            "InvertedPendulum_Implicit", "InvertedPendulum_ImplicitTB", "InvertedPendulum_Robust",
            "InvertedPendulum_Realizable", "InvertedPendulum_RealizableSampled"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples")),
                                reason="the reference tree is not on this machine")


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    import __graft_entry__
    if not (os.path.exists(os.path.join(HOST, "libasif_host.a"))
            and os.path.exists(os.path.join(ROOT, "asif_amd", "libasif_hip.so"))):
        __graft_entry__.build()
    d = tmp_path_factory.mktemp("refex")
    data = d / "reference_data_headers"
    data.mkdir()
    for h in DATA_HEADERS:
        os.symlink(os.path.join(REF, "include", h), data / h)
    return d, data


def _compile(tree, name, std):
    d, data = tree
    exe = d / f"{name}.{std.replace('+', 'p')}"
    cmd = ["g++", f"-std={std}", "-O0", "-Wall",
           "-I", os.path.join(HOST, "include"), "-I", os.path.join(ROOT, "include"), "-I", str(data),
           os.path.join(REF, "examples", name + ".cpp"), os.path.join(HOST, "libasif_host.a"),
           "-L" + os.path.join(ROOT, "asif_amd"), "-lasif_hip", "-L" + os.path.join(ROCM, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "asif_amd"), "-Wl,-rpath," + os.path.join(ROCM, "lib"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r, exe


@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_example_links_against_the_mirror(tree, name):
    # gnu++17 = g++ 11's default, what the reference's CMake (no CMAKE_CXX_STANDARD) gives it here
    r, exe = _compile(tree, name, "gnu++17")
    errors = [ln for ln in r.stderr.splitlines() if "error" in ln or "undefined reference" in ln]
    assert r.returncode == 0, "\n".join(errors[:12]) or r.stderr[-2000:]
    # the program's filter class is the mirror's, out of libasif_host.a, and its solver is the device one
    syms = subprocess.run(["nm", "-C", "--defined-only", str(exe)], capture_output=True, text=True).stdout
    assert "ASIF::QPWrapperHip::solve()" in syms
    assert "ASIF::ASIF" in syms
    needed = subprocess.run(["readelf", "-d", str(exe)], capture_output=True, text=True).stdout
    assert "libasif_hip.so" in needed
    assert "osqp" not in needed.lower()


def test_reference_example_compiles_as_cxx11(tree):
    """The mirror's headers stay within C++11, the language level of the reference's sources."""
    for name in ("DoubleIntegrator_Robust", "InvertedPendulum_Implicit"):
        r, _ = _compile(tree, name, "c++11")
        assert r.returncode == 0, r.stderr[-2000:]


def test_no_reference_api_header_was_needed(tree):
    """Every header an example names apart from the data headers resolves inside asif_amd/host/include."""
    mirror = set(os.listdir(os.path.join(HOST, "include")))
    for h in ("asif++.h", "asif.h", "asif_utils.h", "asif_robust.h", "asif_realizable.h", "asif_implicit.h",
              "asif_implicit_robust.h", "asif_implicit_tb.h", "asif_learning_utils.h", "qpwrapper_abstract.h",
              "qpwrappers.h", "aa.h"):
        assert h in mirror, h
    d, data = tree
    assert sorted(os.listdir(data)) == sorted(DATA_HEADERS)


def test_linked_example_fails_loudly_without_a_gpu(tree):
    """No CPU fallback behind the default solver: on a machine without a gfx950 device every filter() of the linked
    reference program reports failure (examples/DoubleIntegrator.cpp:93-94 prints it) instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r, exe = _compile(tree, "DoubleIntegrator", "gnu++17")
    assert r.returncode == 0
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout
    lines = out.splitlines()
    assert sum(1 for ln in lines if "ASIF failed" in ln) >= 1000
    shutil.rmtree(str(tree[0]), ignore_errors=True)
