"""CPU: accuracy of the numeric contract (include/vxrt_detmath.h) against numpy binary64, so that the
pinned built-ins are also GOOD built-ins (a few ulp), not merely reproducible ones."""
import numpy as np


def ulp_err(got, ref):
    ref32 = ref.astype(np.float32)
    return np.abs(got.astype(np.float64) - ref) / np.spacing(np.abs(ref32)).astype(np.float64)


def test_sin_cos_tan(O):
    x = np.random.default_rng(0).uniform(-100, 100, 400000).astype(np.float32)
    assert np.abs(O.detmath("sin", x) - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(O.detmath("cos", x) - np.cos(x.astype(np.float64))).max() < 2e-7
    x = np.random.default_rng(1).uniform(-1.4, 1.4, 100000).astype(np.float32)
    assert ulp_err(O.detmath("tan", x), np.tan(x.astype(np.float64))).max() < 6
    assert np.isnan(O.detmath("sin", np.array([np.inf, np.nan, 1e9], np.float32))).all()


def test_exp_log_pow(O):
    x = np.random.default_rng(2).uniform(-87, 88, 400000).astype(np.float32)
    assert ulp_err(O.detmath("exp", x), np.exp(x.astype(np.float64))).max() < 3
    assert O.detmath("exp", np.array([-200, -88, 0, 89, 200], np.float32)).tolist() == [0, 0, 1, np.inf, np.inf]
    x = np.exp(np.random.default_rng(3).uniform(-85, 85, 400000)).astype(np.float32)
    assert ulp_err(O.detmath("log", x), np.log(x.astype(np.float64))).max() < 3
    edge = O.detmath("log", np.array([0, 1, np.inf, -1, 1e-45], np.float32))
    assert edge[0] == -np.inf and edge[1] == 0 and edge[2] == np.inf and np.isnan(edge[3]) and abs(edge[4] - np.log(1.4e-45)) < 1e-3
    x = np.random.default_rng(4).uniform(0.5, 1, 200000).astype(np.float32)
    y = np.full_like(x, 399.99997)
    got, ref = O.detmath("pow", x, y).astype(np.float64), np.power(x.astype(np.float64), y.astype(np.float64))
    m = ref > 1e-30
    assert (np.abs(got - ref)[m] / ref[m]).max() < 1e-4           # exponent 400 amplifies log's ulp
    assert O.detmath("pow", np.array([0, 0, 1], np.float32), np.array([400, 0, 400], np.float32)).tolist() == [0, 1, 1]


def test_fused_exp_is_within_one_ulp_of_the_unfused_form(O):
    """round 4 rewrote vx_exp's kernel with fused multiply-adds and regenerated the self-golden frames in the same commit, so
    "bit-exact against the oracle" held by construction.  What justifies the regenerated goldens is checked here: over the denoiser's
    argument range (the exact tap's exp(-factor_range - factor_distance), a number in [-87.3, 0]) and over vx_pow's (y * log(x) for the
    sun disc: x in (0, 1], y = 400 -> [-87.3, 0] as well; emitters and parameters outside the defaults: up to +88.7) the two forms
    differ by at most one unit in the last place, the fused form is never further from the true value than the unfused one by more than
    that, and both flush to +0 / overflow to inf at the same arguments.  Neither form is pinned by the reference (U6: parity unpinned)."""
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-87.3, 0, 1_500_000), rng.uniform(0, 88.72, 500_000), -np.exp(rng.uniform(-30, 4.4, 500_000)),
                        np.array([-87.3, -87.29999, 0.0, -0.0, 88.72, 88.71999, -1e-30, 1e-30])]).astype(np.float32)
    fused, unfused = O.detmath("exp", x), O.detmath("exp_unfused", x)
    ulps = np.abs(fused.view(np.int32).astype(np.int64) - unfused.view(np.int32).astype(np.int64))
    assert ulps.max() <= 1, ulps.max()
    assert 0.02 < (ulps == 1).mean() < 0.5                       # they DO differ (so the goldens had to move), by one ulp, on a minority
    true = np.exp(x.astype(np.float64))
    m = np.isfinite(fused) & (true < 3.0e38)
    assert ulp_err(fused[m], true[m]).max() <= ulp_err(unfused[m], true[m]).max() + 1e-9 < 3
    edge = np.array([-200, -87.31, -87.3, 88.72, 88.73, 200, np.nan], np.float32)
    a, b = O.detmath("exp", edge), O.detmath("exp_unfused", edge)
    assert np.array_equal(a, b, equal_nan=True) and a[0] == 0 and a[1] == 0 and a[2] > 0 and np.isinf(a[4]) and np.isnan(a[6])


def test_glsl_min_max_sign_semantics(O):
    # min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x  -> a NaN in the SECOND operand is ignored, in the first it wins.
    # checked through pow/exp/log being NaN-propagating and through the traversal tests; here: f2i definedness via tan/div
    z = O.detmath("div", np.array([1, -1, 0, 1], np.float32), np.array([0, 0, 0, np.inf], np.float32))
    assert z[0] == np.inf and z[1] == -np.inf and np.isnan(z[2]) and z[3] == 0


def test_min0_equals_glsl_min_for_every_binary32(tmp_path):
    """vx_min0(v) — the integer-compare form the kernels use for min(0.0, v) because the gfx950 backend folds the literal select into
    v_min_f32 (-0 for v = -0) — against the literal GLSL wording, for all 2^32 bit patterns, bit for bit (gcc, -O2)."""
    import os
    import subprocess
    from conftest import ROOT
    src = tmp_path / "min0.c"
    src.write_text('''
#include <stdio.h>
#include "vxrt_detmath.h"
int main(void) {
    unsigned long long bad = 0;
    uint32_t b = 0;
    do {
        float v = vx_u2f(b);
        uint32_t want = vx_f2u(vx_min(0.0f, v)), got = vx_f2u(vx_min0(v));
        if (want != got) { if (bad < 4) printf("%08x: want %08x got %08x\\n", b, want, got); bad++; }
    } while (++b != 0);
    printf("bad %llu\\n", bad);
    return bad != 0;
}
''')
    exe = tmp_path / "min0"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("bad 0"), out.stdout


def test_sun_power_is_exactly_zero_below_the_kernels_bound(O):
    """The tracer skips pow(x, 1/sun_size^2) (voxels.comp:378-381) for x below TraceArgs::sun_zero_below =
    float(exp(-88 / y) * (1 - 1e-4)) (csrc/api_trace.hip frame_constants, csrc/trace_common.h sun_power_of) because
    vx_pow(x, y) = vx_exp(y * vx_log(x)) is +0 there.  Checked on the contract's own vx_pow: the 2^21 binary32 values just below
    the bound, a log-uniform sample of everything below it, zero and the denormals — for the default exponent 400 and others."""
    rng = np.random.default_rng(3)
    for y in (400.0, 1.0 / (0.05 * 0.05), 1.5, 10.0, 123.456, 1e4, 9.9e5):
        y32 = np.float32(y)
        bound = np.float32(np.exp(-88.0 / float(y32)) * (1.0 - 1e-4))
        assert 0 < bound < 1
        top = (bound.view(np.uint32) - np.arange(1, 1 << 21, dtype=np.uint32)).view(np.float32)    # the floats just below the bound
        sample = np.exp(rng.uniform(np.log(1e-38), np.log(float(bound)), 1 << 20)).astype(np.float32)
        sample = sample[sample < bound]
        small = np.concatenate([[0.0], np.float32(1e-45) * np.arange(1, 1000, dtype=np.float32)]).astype(np.float32)
        for x in (top, sample, small):
            p = O.detmath("pow", x, np.full_like(x, y32))
            assert (p.view(np.uint32) == 0).all(), (y, float(x[(p != 0) | np.signbit(p)][0]))
        # and the bound is not absurdly conservative: two units of the exponent above it the power is positive
        above = np.float32(np.exp(-86.0 / float(y32)))
        assert O.detmath("pow", np.array([above]), np.array([y32]))[0] > 0
