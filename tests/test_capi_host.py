"""The C-ABI library without a GPU: it loads, exports every function include/asif_hip.h declares,
its host-side option/dimension logic matches the oracle's restatement of the reference constructors,
and it FAILS LOUDLY (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import __graft_entry__  # builds in-tree if needed
    from asif_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        __graft_entry__.build()
    capi.load()
    return capi


def test_exports_every_declared_symbol(capi):
    header = open(os.path.join(ROOT, "include", "asif_hip.h")).read()
    declared = set(re.findall(r"\b(asif_hip_[a-z_]+)\s*\(", header))
    assert len(declared) >= 12
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(capi.EXPORTS)
    assert lib.asif_hip_version() >= 100
    assert b"gfx950" in lib.asif_hip_error_string(-2)


def test_default_options_match_oracle(capi, oracle):
    for cfg, (model, variant, _) in capi.CONFIGS.items():
        o = capi.default_options(model, variant)
        ref = oracle.default_options(*oracle.CONFIGS[cfg])
        for name, _ in capi.Options._fields_:
            a, b = getattr(o, name), getattr(ref, name)
            if hasattr(a, "__len__"):
                assert list(a) == list(b), (cfg, name)
            else:
                assert a == b, (cfg, name)


def test_model_and_variant_ids_match_the_header(capi):
    """asif_amd/capi.py restates the enums of include/asif_hip.h by value: a model added on one side only would
    silently bind another model."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "asif_hip.h")).read()
    ids = {m.group(1): int(m.group(2)) for m in re.finditer(r"ASIF_HIP_MODEL_([A-Z_0-9]+)\s*=\s*(\d+)", hdr)}
    assert ids["DOUBLE_INTEGRATOR"] == capi.MODEL_DOUBLE_INTEGRATOR and ids["INVERTED_PENDULUM"] == capi.MODEL_INVERTED_PENDULUM
    assert ids["SEGWAY"] == capi.MODEL_SEGWAY and ids["INVERTED_PENDULUM_ROBUST"] == capi.MODEL_INVERTED_PENDULUM_ROBUST
    assert ids["INVERTED_PENDULUM_TB"] == capi.MODEL_INVERTED_PENDULUM_TB == capi.CONFIGS[8][0]
    assert ids["DOUBLE_INTEGRATOR_IMPLICIT"] == capi.MODEL_DOUBLE_INTEGRATOR_IMPLICIT == capi.CONFIGS[9][0]
    assert ids["PLANAR_TWO_INPUT"] == capi.MODEL_PLANAR_TWO_INPUT == capi.CONFIGS[11][0]
    assert ids["DOUBLE_INTEGRATOR_TB"] == capi.MODEL_DOUBLE_INTEGRATOR_TB == capi.CONFIGS[12][0]
    assert sorted(ids.values()) == list(range(len(ids)))  # dense, no duplicates


def test_struct_layouts_match_header(capi):
    # sizes the C compiler gives the same structs (gcc on the public header)
    import subprocess, tempfile
    src = '#include "asif_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu\\n",sizeof(asif_hip_options),sizeof(asif_hip_solver),sizeof(asif_hip_dims),sizeof(asif_hip_realizable_options),sizeof(asif_hip_robust_data_options));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "s.c")
        open(p, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), p, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(capi.Options), C.sizeof(capi.Solver), C.sizeof(capi.Dims),
                     C.sizeof(capi.RealizableOptions), C.sizeof(capi.RobustDataOptions)]


def test_no_gpu_means_loud_failure(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.AsifHipError):
        capi.Filter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT)
    lib = capi.load()
    s = capi.default_solver()
    r = lib.asif_hip_qp_solve_batch(0, C.byref(s), 4, 4, 2, 4, *([C.c_void_p(8)] * 6), None, C.c_void_p(8),
                                    C.c_void_p(8), None, None)
    assert r == -2  # ASIF_HIP_ENODEVICE


def test_bad_arguments_are_rejected(capi):
    lib = capi.load()
    assert lib.asif_hip_default_options(99, 0, C.byref(capi.Options())) == -1
    assert lib.asif_hip_default_solver(None) == -1
    h = C.c_void_p()
    assert lib.asif_hip_create(C.byref(h), 0, 3, None, None, 0) != 0  # DoubleIntegrator has no robust variant
    assert not h.value


def test_partition_arithmetic(capi):
    """asif_hip_partition: contiguous blocks, remainder to the first ones, every instance exactly once."""
    for B in (0, 1, 7, 8, 9, 65536, 262144, 262147):
        for n in (1, 2, 3, 4, 8):
            cuts = [capi.partition(B, n, r) for r in range(n)]
            assert cuts[0][0] == 0 and sum(c for _, c in cuts) == B
            for r in range(1, n):
                assert cuts[r][0] == cuts[r - 1][0] + cuts[r - 1][1]
            sizes = [c for _, c in cuts]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert capi.partition(262144, 8, 3) == (3 * 32768, 32768)  # BASELINE.json config 4
    with pytest.raises(capi.AsifHipError):
        capi.partition(10, 2, 2)


def test_multi_create_without_gpu_fails_loudly(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.AsifHipError):
        capi.MultiFilter(capi.MODEL_DOUBLE_INTEGRATOR, capi.EXPLICIT, [0, 1])


def test_create_multi_releases_everything_exactly_once(tmp_path):
    """asif_hip_create_multi's ownership rules (asif_amd/csrc/multi_own.hpp) on counting stand-ins: whichever step
    fails -- the k-th handle, or the k-th stream after every handle exists -- each handle and stream made so far is
    released once and only once (round 2 destroyed the handles twice when a stream could not be created)."""
    import subprocess
    so = str(tmp_path / "libmulti_own.so")
    if os.environ.get("ASIF_SAN_DIR"):  # tests/test_sanitizers.py: the build of `make -C tests san`
        so = os.path.join(os.environ["ASIF_SAN_DIR"], "libmulti_own_san.so")
    else:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "asif_amd", "csrc"),
                               os.path.join(ROOT, "tests", "host_multi_own_driver.cpp"), "-o", so])
    lib = C.CDLL(so)
    n = 4
    for fail_create, fail_stream in [(-1, -1)] + [(k, -1) for k in range(n)] + [(-1, k) for k in range(n)]:
        destroyed = (C.c_int32 * n)()
        unmade = (C.c_int32 * n)()
        lh, ls = C.c_int32(), C.c_int32()
        r = lib.multi_own_scenario(n, fail_create, fail_stream, destroyed, unmade, C.byref(lh), C.byref(ls))
        if fail_create < 0 and fail_stream < 0:
            assert r == 0 and lh.value == n and ls.value == n
            assert list(destroyed) == [0] * n and list(unmade) == [0] * n
            continue
        assert r == (77 if fail_create >= 0 else 88)
        assert lh.value == 0 and ls.value == 0
        made_h = fail_create if fail_create >= 0 else n
        made_s = fail_stream if fail_stream >= 0 else 0
        assert list(destroyed) == [1] * made_h + [0] * (n - made_h)
        assert list(unmade) == [1] * made_s + [0] * (n - made_s)
