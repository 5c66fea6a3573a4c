"""Several devices behind one call (asif_hip_filter_batch_host_multi): blocks of the batch on their own
handles, host threads and streams.  On the one-GPU box the device list names device 0 several times -- the code path
is the one an 8-GPU node takes (separate handles, threads, streams, staging, offsets); the result must be bitwise
the single-handle result.  An N-GPU run proper needs an N-GPU node: unmeasured until the driver has one."""
import numpy as np
import pytest
import torch

import gpu_util
from asif_amd import workloads

pytestmark = pytest.mark.gpu


def _host_single(hip, cfg, x, udes, uact0, relax0):
    model, variant, _ = hip.CONFIGS[cfg]
    flt = hip.Filter(model, variant)
    d = flt.dims
    B = x.shape[1]
    uact = np.full((d.nu, B), uact0)
    relax = np.full((d.nrelax, B), relax0)
    rc = np.zeros(B, dtype=np.int32)
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    hip.check(flt.lib.asif_hip_filter_batch_host(flt.handle, C.c_int64(B), p(x), p(udes), p(uact), p(relax), p(rc)))
    flt.close()
    return uact, relax, rc


# (4, 262144, 8): BASELINE configs[3] itself -- 262 144 segway agents in 8 blocks of 32 768, the split an 8-GPU node
# takes (here: eight handles, host threads and streams on device 0)
@pytest.mark.parametrize("cfg,B,ndev", [(2, 65536, 2), (2, 10007, 3), (4, 4099, 4), (3, 130, 2), (4, 262144, 8)])
def test_multi_equals_single_bitwise(hip, cfg, B, ndev):
    x, udes = workloads.make_batch(cfg, B)
    x, udes = np.ascontiguousarray(x), np.ascontiguousarray(udes)
    ua1, rl1, rc1 = _host_single(hip, cfg, x, udes, 7.0, -7.0)
    model, variant, _ = hip.CONFIGS[cfg]
    m = hip.MultiFilter(model, variant, [0] * ndev)
    d = m.dims
    ua = np.full((d.nu, B), 7.0)
    rl = np.full((d.nrelax, B), -7.0)
    rc = np.zeros(B, dtype=np.int32)
    m.filter_host(x, udes, ua, rl, rc)
    m.close()
    assert np.array_equal(rc, rc1) and np.array_equal(ua, ua1) and np.array_equal(rl, rl1)
    assert len(np.unique(rc)) >= 2  # more than one branch exercised, untouched slots included above


def test_multi_matches_device_pointer_path(hip, oracle):
    B = 8192
    out = gpu_util.run_filter(2, B)
    m = hip.MultiFilter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, [0, 0])
    ua = np.zeros((1, B))
    rl = np.zeros((1, B))
    rc = np.zeros(B, dtype=np.int32)
    m.filter_host(np.ascontiguousarray(out["x"]), np.ascontiguousarray(out["udes"]), ua, rl, rc)
    m.close()
    assert np.array_equal(rc, out["rc"]) and np.array_equal(ua, out["uact"])


def test_more_blocks_than_instances(hip):
    x, udes = workloads.make_batch(2, 3)
    m = hip.MultiFilter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, [0] * 5)
    ua, rl, rc = np.zeros((1, 3)), np.zeros((1, 3)), np.zeros(3, dtype=np.int32)
    m.filter_host(np.ascontiguousarray(x), np.ascontiguousarray(udes), ua, rl, rc)
    m.close()
    assert set(rc.tolist()) <= {1, -1} and np.all(rc != 0)


@pytest.mark.parametrize("cfg,B,ndev", [(2, 10007, 1), (2, 10007, 3), (4, 4099, 2), (3, 130, 2)])
def test_page_locked_buffers_take_the_zero_copy_path(hip, cfg, B, ndev):
    """Page-locked (device-addressable) host buffers are read and written in place by the kernels instead of being
    staged; interior pointers of one allocation (the blocks of the multi-device entry) included.  Same results, bitwise,
    as pageable buffers through the staged path; untouched slots keep the caller's values."""
    x, udes = workloads.make_batch(cfg, B)
    x, udes = np.ascontiguousarray(x), np.ascontiguousarray(udes)
    ua1, rl1, rc1 = _host_single(hip, cfg, x, udes, 7.0, -7.0)
    model, variant, _ = hip.CONFIGS[cfg]
    m = hip.MultiFilter(model, variant, [0] * ndev)
    d = m.dims
    tx, tu = torch.from_numpy(x).pin_memory(), torch.from_numpy(udes).pin_memory()
    ua = torch.full((d.nu, B), 7.0, dtype=torch.float64).pin_memory()
    rl = torch.full((d.nrelax, B), -7.0, dtype=torch.float64).pin_memory()
    rc = torch.zeros(B, dtype=torch.int32).pin_memory()
    m.filter_host(tx.numpy(), tu.numpy(), ua.numpy(), rl.numpy(), rc.numpy())
    m.close()
    assert np.array_equal(rc.numpy(), rc1) and np.array_equal(ua.numpy(), ua1) and np.array_equal(rl.numpy(), rl1)


def test_multi_update_options_is_all_or_nothing(hip):
    """asif_hip_multi_update_options checks the options against every handle before any handle takes them:
    refused options leave all blocks of the batch on the old ones; accepted options reach all of them."""
    import ctypes as C
    B = 3000
    x, udes = workloads.make_batch(2, B)
    x, udes = np.ascontiguousarray(x), np.ascontiguousarray(udes)
    m = hip.MultiFilter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, [0, 0, 0])

    def run():
        ua, rl, rc = np.full((1, B), 7.0), np.full((1, B), -7.0), np.zeros(B, dtype=np.int32)
        m.filter_host(x, udes, ua, rl, rc)
        return ua, rl, rc

    before = run()
    bad = hip.default_options(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    bad.relaxLb = 2.0
    bad.integrator = 2  # no such integrator: EINVAL
    assert m.lib.asif_hip_multi_update_options(m.handle, C.byref(bad)) < 0
    after_bad = run()
    assert all(np.array_equal(a, b) for a, b in zip(before, after_bad))
    good = hip.default_options(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT)
    good.relaxLb = 2.0
    assert m.lib.asif_hip_multi_update_options(m.handle, C.byref(good)) == 0
    ua, rl, rc = run()
    ok = rc == 1
    assert ok.sum() > 100 and np.all(rl[0, ok] == 2.0)  # the pinned relaxation variable of every block moved
    single = hip.Filter(hip.MODEL_DOUBLE_INTEGRATOR, hip.EXPLICIT, options=good)
    single.close()
    m.close()
