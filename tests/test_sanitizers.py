"""The suite's host-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5, row
"host tests under -fsanitize=address,undefined"): `make -C tests san` builds the CPU oracle (oracle/*.c), the product's
dual active-set solver compiled for the host (asif_amd/csrc/gi_small.hpp behind tests/host_gi_driver.cpp), QPWrapperHost's
Newton stage (asif_amd/host/qp_alm_host.cpp), the
multi-device ownership rules (multi_own.hpp), the mirror's host affine arithmetic (asif_affine.h) and the mirror's
classes themselves (asif_amd/host/*.cpp behind the example programs, run with `--solver host`) with
-fsanitize=address,undefined -fno-sanitize-recover=all, and the existing host cases run on those builds in a child
python with the sanitizer runtimes preloaded.  CPU build only: never on the GPU box (not a gpu test; GPU
AddressSanitizer is not available on the pool).

First run of this file found one: tests/oracle_lib.rb_last_learning handed the oracle a one-double buffer for an
OR_MAX_NU = 2 array (8 bytes past a numpy allocation on every call)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_DIR = os.path.join(ROOT, "tests", "_san")

# the host cases: everything that loads native code of this repo without a GPU, sized for minutes under the sanitizers
CASES = [
    "tests/test_gi_host.py",
    "tests/test_alm_host.py",
    "tests/test_host_affine.py",
    "tests/test_capi_host.py::test_create_multi_releases_everything_exactly_once",
    "tests/test_oracle_qp.py",
    "tests/test_oracle_affine.py",
    "tests/test_oracle_robust_data.py",
    "tests/test_oracle_implicit_rb.py",
    "tests/test_oracle_dopri.py",
    "tests/test_oracle_realizable.py",
    "tests/test_oracle_assembly.py",
]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(os.environ.get("ASIF_SAN_DIR") is not None, reason="already inside the sanitizer run")
def test_host_native_code_is_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("this gcc ships no sanitizer runtimes")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests"), "-s", "san"])
    env = dict(os.environ)
    env.update({
        "ASIF_SAN_DIR": SAN_DIR,
        "LD_PRELOAD": asan + ":" + ubsan,
        # python itself is not leak-clean; everything else aborts the child at the first report
        "ASAN_OPTIONS": f"detect_leaks=0:abort_on_error=0:exitcode=23:log_path={tmp_path}/asan",
        "UBSAN_OPTIONS": f"print_stacktrace=1:halt_on_error=1:exitcode=24:log_path={tmp_path}/ubsan",
    })
    # (the exhaustive enumeration that checks the 3 x 41 stress families is minutes of sanitized oracle time for no new
    # code path: the same families run at 3 x 17 and the config QPs at 3 x 41)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        "-k", "not (stress_families and 3-41)"] + CASES,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    reports = sorted(f for f in os.listdir(tmp_path) if f.startswith(("asan", "ubsan")))
    text = "".join(open(os.path.join(tmp_path, f)).read()[:4000] for f in reports)
    assert not reports, "sanitizer report:\n" + text
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    # the mirror's classes (asif_amd/host/*.cpp: the host Euler loop with its sensitivity, the critical-sample selection,
    # the TB rows) and QPWrapperHost behind the example programs, `--solver host`: no device anywhere in the run
    env2 = {k: v for k, v in env.items() if k != "LD_PRELOAD"}  # executables link their runtimes themselves
    runs = [["double_integrator_san", "--solver", "host", "--steps", "600"],
            ["backup_filters_san", "implicit-loop", "12", "5", "--solver", "host"],
            ["backup_filters_san", "dii-loop", "400", "--solver", "host"],
            ["backup_filters_san", "tbip-loop", "6", "1", "--solver", "host"],
            ["backup_filters_san", "tb-loop", "300", "0.5", "--solver", "host"],
            ["backup_filters_san", "tbdi-loop", "60", "--solver", "host"],
            ["implicit_rb_san", "6", "--solver", "host"],            # ASIFimplicitRB: host AAF margins, held input, networks
            ["implicit_rb_san", "6", "plain", "--solver", "host"],
            ["explicit_variants_san", "48", "--solver", "host"],     # class ASIF: two inputs, npSSmax < npSS, updateOptions
            ["robust_pendulum_san", "--solver", "host", "24"],       # ASIFrobust: host AAF rows, 18 x 12 on the Newton stage
            ["robust_pendulum_san", "--solver", "host", "--loop", "60", "1.0", "1.5"]]
    # the two classes built from data files (the shipped half-planes; the 100 Hz kernel): files written from the fixtures
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import test_gpu_host_cpp as twin
    hp = oracle_lib.load_halfplanes()
    hfile, kfile = str(tmp_path / "hp.txt"), str(tmp_path / "kernel.txt")
    with open(hfile, "w") as f:
        f.write(f"{hp.shape[0]}\n" + "".join(f"{float(a[0])!r} {float(a[1])!r}\n" for a in hp))
    twin._write_kernel(oracle_lib.load_kernel("100Hz"), kfile)
    runs += [["di_robust_san", "--solver", "host", hfile, "--loop", "120"],      # 22 x 15, infeasible steps included
             ["realizable_di_san", "--solver", "host", kfile, "--loop", "600"]]  # facet tests (2 x 5) + 38 x 29
    for cmd in runs:
        p = subprocess.run([os.path.join(SAN_DIR, cmd[0])] + cmd[1:], env=env2, capture_output=True, text=True, timeout=900)
        reports = sorted(f for f in os.listdir(tmp_path) if f.startswith(("asan", "ubsan")))
        assert not reports, "sanitizer report in " + " ".join(cmd) + ":\n" + open(os.path.join(tmp_path, reports[0])).read()[:4000]
        assert p.returncode == 0 and len(p.stdout.strip().split("\n")) >= 6, (cmd, p.returncode, p.stderr[-1000:])
