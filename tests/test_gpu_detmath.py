"""The numeric contract (include/vxrt_detmath.h) promises bit-identical results on host and device.
This runs every primitive on the GPU (vxrt_detmath_probe) and on the CPU (oracle build of the same
header, g++) over wide random inputs and edge values and compares bit patterns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EDGE = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 3.0, 1e-38, 1e-45, -1e-45, 1e38, 3.4e38, np.inf, -np.inf, np.nan,
                 0.05, 400.0, 1.32, 6.2831855, 1.5707964, 88.0, -87.0, -104.0, 1e-5, 64.00001], np.float32)


def inputs(lo, hi, seed, n=1 << 20):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(lo, hi, n).astype(np.float32), EDGE])


@pytest.mark.parametrize("fn,lo,hi", [("sin", -50, 50), ("cos", -50, 50), ("tan", -1.5, 1.5), ("exp", -110, 90),
                                      ("log", 0, 1e6), ("sqrt", 0, 1e6)])
def test_unary_bit_equal(O, H, fn, lo, hi):
    x = inputs(lo, hi, 1)
    dev, cpu = H.detmath_probe(fn, x), O.detmath(fn, x)
    same = (dev.view(np.uint32) == cpu.view(np.uint32)) | (np.isnan(dev) & np.isnan(cpu))
    bad = np.flatnonzero(~same)
    assert len(bad) == 0, [(float(x[i]), float(dev[i]), float(cpu[i])) for i in bad[:5]]


def test_div_pow_bit_equal(O, H):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-100, 100, 1 << 20).astype(np.float32), EDGE, np.repeat(EDGE, len(EDGE))])
    y = np.concatenate([np.exp(rng.uniform(-20, 20, 1 << 20)).astype(np.float32) * rng.choice([-1, 1], 1 << 20).astype(np.float32),
                        EDGE[::-1], np.tile(EDGE, len(EDGE))])
    for fn in ("div",):
        dev, cpu = H.detmath_probe(fn, x, y), O.detmath(fn, x, y)
        m = ~(np.isnan(dev) & np.isnan(cpu))
        assert np.array_equal(dev[m].view(np.uint32), cpu[m].view(np.uint32)), fn
    xp = np.concatenate([rng.uniform(0, 1.2, 1 << 20).astype(np.float32), EDGE[EDGE >= 0]])
    yp = np.concatenate([np.full(1 << 20, 399.99997, np.float32), np.full((EDGE >= 0).sum(), 2.5, np.float32)])
    dev, cpu = H.detmath_probe("pow", xp, yp), O.detmath("pow", xp, yp)
    m = ~(np.isnan(dev) & np.isnan(cpu))
    assert np.array_equal(dev[m].view(np.uint32), cpu[m].view(np.uint32))


SPECIAL = np.array([0.0, -0.0, 0.25, 0.5, 0.75, 0.125, 1 - 2.0 ** -24, 1.0, -1.0, 2.0, -2.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40,
                    1e-45, -1e-45, 3.0, -0.5], np.float32)


@pytest.mark.parametrize("fn", ["hemi_y", "hemi_z", "mul", "sub", "flip", "min", "max", "max0", "sign", "clamp", "min0"])
def test_signs_of_zero_and_special_operands(O, H, fn):
    """Bit equality INCLUDING the sign of zero on every pair of special operands.  (A bounce direction (-1, +0, -0) and one
    (-1, -0, -0) walk different octants: a kernel compiler that folds min(0.0, v) into v_min_f32 changed 16 pixels of a frame
    whose noise table holds exact zeros — found by tests/test_gpu_stress.py at scale 12, fixed with vx_min0.)"""
    x, y = [a.ravel() for a in np.meshgrid(SPECIAL, SPECIAL)]
    dev, cpu = H.detmath_probe(fn, x, y), O.detmath(fn, x, y)
    bad = np.flatnonzero((dev.view(np.uint32) != cpu.view(np.uint32)) & ~(np.isnan(dev) & np.isnan(cpu)))
    assert len(bad) == 0, [(float(x[i]), float(y[i]), float(dev[i]), float(cpu[i])) for i in bad[:5]]


def test_min0_on_device_over_every_exponent(H):
    """vx_min0 on the device against the GLSL wording evaluated in numpy: every sign x exponent with the extreme and some random
    mantissas (zeros, denormals, infinities, quiet and signalling NaN patterns included), bit for bit."""
    rng = np.random.default_rng(5)
    mant = np.concatenate([[0, 1, 2, 0x3fffff, 0x400000, 0x400001, 0x7ffffe, 0x7fffff], rng.integers(0, 1 << 23, 56)]).astype(np.uint32)
    se = (np.arange(512, dtype=np.uint32) << 23)
    bits = (se[:, None] | mant[None, :]).ravel()
    v = bits.view(np.float32)
    dev = H.detmath_probe("min0", v, np.ones_like(v))                      # vx_min0(v) * 1.0f
    with np.errstate(invalid="ignore"):
        want = np.where(v < 0, v, np.float32(0.0)).astype(np.float32)      # min(0.0, v): "y if y < x, otherwise x" with x = 0.0
    nan = np.isnan(v)
    assert not np.isnan(dev[nan]).any() and (dev[nan].view(np.uint32) == 0).all()      # NaN < 0 is false: +0.0
    assert np.array_equal(dev[~nan].view(np.uint32), want[~nan].view(np.uint32))
