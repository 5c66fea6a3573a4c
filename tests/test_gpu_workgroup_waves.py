"""Launch geometry must not show in the results.  Kernels whose waves are independent run one wave per workgroup on
small launches and four from two waves per CU on (launchers.hpp: waves_per_workgroup; a developer switch forces either),
and the trajectory kernels choose between LDS layouts and pass forms by batch size: rows, diagnostics, uAct, relax and rc
of the same instances are the same BITS whichever was taken -- compared across processes (the switch is read once per
process) on ragged batches, and between a batch and the prefix of a larger one that takes the other path."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gpu_util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(cfg, B, env, integrator=0):
    code = ("import sys, json, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import gpu_util\n"
            "from asif_amd import capi\n"
            "o = capi.default_options(*capi.CONFIGS[%d][:2]); o.integrator = %d\n" % (ROOT, ROOT, cfg, integrator) +
            "r = gpu_util.run_assemble(%d, %d, options=o); f = gpu_util.run_filter(%d, %d, options=o, uact_init=7.0, relax_init=-7.0)\n"
            "print(json.dumps({k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in "
            "(('A', r['A']), ('b', r['b']), ('code', r['code']), ('diag', r['diag']), ('uact', f['uact']), "
            "('relax', f['relax']), ('rc', f['rc']))}))\n" % (cfg, B, cfg, B))
    o = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert o.returncode == 0, o.stderr[-1500:]
    return json.loads(o.stdout.strip().split("\n")[-1])


@pytest.mark.parametrize("cfg,B", [(3, 1000), (9, 4099), (12, 4099), (8, 300), (4, 4099), (5, 4099), (10, 300), (11, 4099)])
def test_one_and_four_waves_per_workgroup_give_the_same_bits(hip, cfg, B):
    one = _digests(cfg, B, {"ASIF_HIP_WG_WAVES": "1"})
    four = _digests(cfg, B, {"ASIF_HIP_WG_WAVES": "4"})
    two = _digests(cfg, B, {"ASIF_HIP_WG_WAVES": "2"})
    assert one == four == two


@pytest.mark.parametrize("cfg,B", [(3, 300), (12, 1000)])
def test_the_adaptive_integrator_too(hip, cfg, B):
    """integrator = 1 (dopri5 dense output, ASIFimplicit and ASIFimplicitTB): one and four waves per workgroup."""
    assert _digests(cfg, B, {"ASIF_HIP_WG_WAVES": "1"}, integrator=1) == _digests(cfg, B, {"ASIF_HIP_WG_WAVES": "4"}, integrator=1)


@pytest.mark.parametrize("cfg,B", [(3, 1000), (10, 300), (8, 700)])
def test_blocks_without_the_bevel_code_give_the_same_bits(hip, cfg, B):
    """The pendulum's rows kernels (ASIFimplicit, ASIFimplicitRB, ASIFimplicitTB) run the blocks of Euler steps that start clear of both bevels of the soft saturation
    on a step without the bevel code, checks afterwards that no lane met one and repeats the block otherwise
    (k_implicit.hip, k_tb.hip, bevel_rate in models.hpp); ASIF_HIP_BEVEL_FREE=0 keeps every block on the full step.  On the
    TB pass lanes that have reached the backup set sit out, and a repeated block must leave them as they stopped."""
    off = _digests(cfg, B, {"ASIF_HIP_BEVEL_FREE": "0"})
    assert off == _digests(cfg, B, {})
    # "2": the prediction without its margin -- a block is taken for clear whenever no lane is IN a bevel at its start, many
    # meet one a few steps later and are repeated from their first sample
    assert off == _digests(cfg, B, {"ASIF_HIP_BEVEL_FREE": "2"})


@pytest.mark.parametrize("cfg,n,big", [(9, 8192, 200000), (12, 4096, 70000), (3, 2048, 40000)])
def test_a_batch_equals_the_prefix_of_a_larger_one(hip, cfg, n, big):
    """n instances alone (few waves: one-wave workgroups, checkpoints in their own LDS region) and as the first n of a
    launch large enough for four-wave workgroups and the spilling layout."""
    small, large = gpu_util.run_filter(cfg, n, uact_init=3.0), gpu_util.run_filter(cfg, big, uact_init=3.0)
    for k in ("uact", "relax", "rc"):
        assert np.array_equal(small[k], large[k][..., :n], equal_nan=True), k
    rs, rl = gpu_util.run_assemble(cfg, n), gpu_util.run_assemble(cfg, big)
    for k in ("A", "b", "code", "diag"):
        assert np.array_equal(rs[k], rl[k][..., :n], equal_nan=True), k

