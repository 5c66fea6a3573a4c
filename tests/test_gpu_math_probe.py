"""The device math the trajectory kernels use in place of library calls, tested on the device itself
(asif_hip_math_probe) against 80-bit references: sincos_fast (models.hpp: range reduction + fdlibm-style kernels,
the quadrant signs as sign-bit XORs), tanh_abs_accurate (absolute accuracy, the segway's friction term), rcp_newton,
and the bevel's sqrt / divide without the IEEE sequences' rescaling steps -- those two must be BITWISE the IEEE
results on their stated operand ranges."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LD = np.longdouble
EPS = 2.0 ** -53


def _ulp(v):
    return np.spacing(np.abs(v))


def test_sincos_fast_accuracy_over_its_range(hip):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-4, 4, 400000), rng.uniform(-1e5, 1e5, 400000), rng.normal(0, 1e-3, 100000),
                        np.array([0.0, -0.0, np.pi / 4, -np.pi / 4, np.pi / 2, np.pi, 1e5, -1e5, 1e-300])])
    s, c = hip.math_probe(0, x)
    rs, rc = np.sin(x.astype(LD)), np.cos(x.astype(LD))
    # absolute error: below 1.5 ulp of 1 over the whole range (the results live in [-1, 1]); relative to the result's
    # own ulp, within 1.5 ulp for the arguments the filters' states produce (|x| <= 4: half an ulp of the reduced
    # argument plus the kernels' own error) and 2 ulp out to 1e5, away from the zeros of sin / cos
    assert np.abs(s.astype(LD) - rs).max() <= 1.5 * EPS and np.abs(c.astype(LD) - rc).max() <= 1.5 * EPS
    for res, ref in ((s, rs), (c, rc)):
        far = np.abs(ref) > 1e-3
        rel = np.abs(res.astype(LD) - ref) / _ulp(ref.astype(np.float64))
        assert rel[far & (np.abs(x) <= 4)].max() <= 1.5
        assert rel[far].max() <= 2.0
    assert np.all(s[x == 0.0] == 0.0) and np.all(c[x == 0.0] == 1.0)  # (the sign of a zero argument is not kept)


def test_sincos_checked_version_covers_the_whole_line(hip):
    x = np.array([1e5 * (1 + 2.0 ** -40), -3e7, 1e15, 1e300, np.inf, -np.inf, np.nan])
    s, c = hip.math_probe(1, x)
    fin = np.isfinite(x)
    assert np.abs(s[fin].astype(LD) - np.sin(x[fin].astype(LD))).max() <= 2 * EPS
    assert np.abs(c[fin].astype(LD) - np.cos(x[fin].astype(LD))).max() <= 2 * EPS
    assert np.all(np.isnan(s[~fin])) and np.all(np.isnan(c[~fin]))
    # and the unpoliced fast path maps NaN / inf to NaN as well (the kernels rely on it)
    s0, c0 = hip.math_probe(0, x[~fin])
    assert np.all(np.isnan(s0)) and np.all(np.isnan(c0))


def test_tanh_absolute_accuracy(hip):
    rng = np.random.default_rng(2)
    y = np.concatenate([rng.uniform(-30, 30, 600000), rng.normal(0, 1, 300000), rng.normal(0, 1e-3, 100000),
                        np.array([0.0, -0.0, 1e-300, 18.7, 19.1, 25.0, -25.0, 26.0, 1e3, -1e9, np.inf, -np.inf])])
    t, _ = hip.math_probe(2, y)
    ref = np.tanh(y.astype(LD))
    assert np.abs(t.astype(LD) - ref).max() <= 4e-16
    assert t[y == np.inf][0] == 1.0 and t[y == -np.inf][0] == -1.0 and np.all(np.abs(t) <= 1.0)
    assert np.signbit(t[-11]) and t[-11] == 0.0 and t[-12] == 0.0  # tanh(-0) = -0, tanh(0) = 0
    tn, _ = hip.math_probe(2, np.array([np.nan]))
    assert np.isnan(tn[0])


def test_rcp_newton_within_an_ulp(hip):
    rng = np.random.default_rng(3)
    d = np.concatenate([rng.uniform(0.1, 100, 500000), -rng.uniform(0.1, 100, 500000), np.exp(rng.uniform(-200, 200, 100000))])
    r, _ = hip.math_probe(3, d)
    q = 1.0 / d.astype(LD)
    assert (np.abs(r.astype(LD) - q) / _ulp((1.0 / d))).max() <= 1.0


def test_bevel_sqrt_and_divide_are_the_ieee_results_on_their_ranges(hip):
    rng = np.random.default_rng(4)
    # sqrt: x = 0 or x >= 2^-767 (backup_traj.hpp)
    x = np.concatenate([np.exp(rng.uniform(np.log(2.0 ** -767), np.log(2.0 ** 1000), 400000)), rng.uniform(0, 2, 400000),
                        np.array([0.0, 1.0, 4.0, 2.0 ** -767, np.inf])])
    s, _ = hip.math_probe(4, x)
    assert np.array_equal(s, np.sqrt(x))
    sn, _ = hip.math_probe(4, np.array([-1.0, np.nan]))
    assert np.all(np.isnan(sn))
    # a / b: both magnitudes within 2^+-255 (a may be 0)
    a = np.concatenate([rng.choice([-1, 1], 500000) * np.exp(rng.uniform(np.log(2.0 ** -255), np.log(2.0 ** 255), 500000)),
                        rng.uniform(-2, 2, 300000), np.zeros(10)])
    b = np.concatenate([np.exp(rng.uniform(np.log(2.0 ** -255), np.log(2.0 ** 255), 500000)), rng.uniform(1e-9, 2, 300000),
                        rng.uniform(0.1, 2, 10)])
    q, _ = hip.math_probe(5, a, b)
    assert np.array_equal(q, a / b)


def test_bevel_arc_pair_is_the_ieee_pair_over_the_arc(hip):
    """backup_traj.hpp bevel_arc: sqrt(d) and n / sqrt(d) from ONE v_rsq (the division reuses the square root's
    reciprocal estimate) must be bitwise what sqrt and divide give, for every satSharpness the options admit
    (0.01 <= r <= 2): d in [r^2 / 2, r^2], |n| <= r."""
    rng = np.random.default_rng(6)
    r = np.concatenate([np.full(400000, 0.1), rng.uniform(0.01, 2.0, 1200000), np.full(200000, 0.01), np.full(200000, 2.0)])
    t = rng.uniform(0.0, 1.0, r.size)
    t[:1000] = 0.0
    t[1000:2000] = 1.0
    d = r * r * (0.5 + 0.5 * t)
    n = np.sqrt(np.maximum(r * r - d, 0.0)) * rng.choice([-1.0, 1.0], r.size)  # as in the kernel: n^2 + d = r^2
    n[2000:200000] = rng.uniform(-1.0, 1.0, 198000) * r[2000:200000]              # and anything else within |n| <= r
    n[200000:200100] = 0.0
    n[200100:200200] = np.spacing(1.0) * r[200100:200200]                           # one ulp inside the clamp threshold
    sq, q = hip.math_probe(7, d, n)
    ref = np.sqrt(d)
    assert np.array_equal(sq, ref)
    assert np.array_equal(q, n / ref)


def test_carried_sincos_over_one_block(hip):
    """kTrigCarried (models.hpp): sin / cos evaluated at the block's first sample and rotated by the angle's increments
    for the 15 steps of a block: at the accumulated angle, within 1.5 + 15/2 ulp of 1 by construction, a few ulp in
    practice; increments up to kTrigCarryMaxStep."""
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-4, 4, 400000)
    d = np.concatenate([rng.uniform(-0.05, 0.05, 200000), rng.normal(0, 2e-3, 200000)])
    s, c = hip.math_probe(6, x0, d)
    x = x0.copy()
    for _ in range(15):
        x = x + d  # the angle the loop holds (accumulated in double, as the Euler state is)
    es = np.abs(s.astype(LD) - np.sin(x.astype(LD)))
    ec = np.abs(c.astype(LD) - np.cos(x.astype(LD)))
    assert max(es.max(), ec.max()) <= 9.5 * EPS
    assert np.sqrt(np.mean(es.astype(np.float64) ** 2)) <= 2.0 * EPS  # typical error: an ulp or two
