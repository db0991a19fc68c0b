"""The arithmetic of the reference-precision encoder (csrc/rr_ce_h2.hip) restated on the CPU.
(1) Its fp32-grade GELU (h2_gelu2): the coefficients in the kernel source are the ones tools/fit_gelu_f32.py produces, and
    the formula -- evaluated in float32 with the kernel's operation order -- stays within fp32 rounding of the erf form the
    reference computes (torch's `gelu` inside BertIntermediate, app/app_product_search.py:277-278).
(2) Its products: every fp32 operand as two fp16 numbers, three products per fp32 product."""
import importlib.util
import pathlib
import re

import numpy as np
from scipy.special import erf

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _tool():
    spec = importlib.util.spec_from_file_location("fit_gelu_f32", ROOT / "tools" / "fit_gelu_f32.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_coefficients_are_the_fitted_ones_and_the_formula_is_fp32_grade():
    tool = _tool()
    src = (ROOT / "review-recommender_amd" / "csrc" / "rr_ce_h2.hip").read_text()
    body = src[src.index("h2_f32x2 h2_gelu2(h2_f32x2 x)"):]
    body = body[:body.index("#undef H2_C2")]
    in_kernel = [np.float32(v) for v in re.findall(r"H2_C2\((-?\d\.\d+e[+-]\d+)f\)", body)][:8]      # P's coefficients, high to low
    assert len(in_kernel) == 8
    coef_u, fit_err = tool.fit(7)
    coef_t = tool.powers_of_t(coef_u)               # low to high
    assert fit_err < 2e-8
    want = [np.float32(c) for c in coef_t[::-1]]
    for a, b in zip(in_kernel, want):
        assert abs(float(a) - float(b)) <= 2e-6 * max(1.0, abs(float(b))), (in_kernel, want)
    assert "4.25f" in body and abs(tool.T_MAX - 4.25) < 1e-12
    # the formula with the KERNEL's constants, in float32, against the float64 erf form
    x = np.concatenate([np.linspace(-10, 10, 400_001), np.linspace(-0.02, 0.02, 4001), [0.0, -0.0, 4.25 * np.sqrt(2), 30.0, -30.0]])
    x = tool.f32(x)
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    got = tool.gelu_kernel(x, [float(c) for c in in_kernel[::-1]])
    err = np.abs(got - ref)
    assert err.max() < 3e-7 and (err / np.maximum(1.0, np.abs(x))).max() < 1.2e-7, (err.max(), x[err.argmax()])
    assert got[np.abs(x) == 0].tolist() == [0.0, 0.0] or np.all(got[np.abs(x) == 0] == 0)


def test_fp16_pair_products_are_as_exact_as_an_fp32_gemm():
    """The arithmetic of csrc/rr_ce_h2.hip restated in numpy: x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) * 2048),
    x y = hi hi + (hi lo + lo hi) / 2048.  On the encoder's operand statistics the three-product sum sits closer to the float64
    product than a float32 GEMM does, for K = 384 and K = 1536, and values in fp16's subnormal range keep 22 bits too (numpy's
    float16 rounds subnormals as the hardware's conversions do)."""
    rng = np.random.default_rng(0)

    def split(x):
        hi = x.astype(np.float16).astype(np.float32)
        lo = ((x - hi) * np.float32(2048)).astype(np.float16).astype(np.float32)
        return hi.astype(np.float64), lo.astype(np.float64)

    for k in (384, 1536):
        a = rng.standard_normal((256, k)).astype(np.float32)
        a[rng.random(a.shape) < 0.05] *= np.float32(1e-6)               # some entries in and below the subnormal range
        w = (rng.standard_normal((128, k)) * 0.05).astype(np.float32)
        exact = a.astype(np.float64) @ w.astype(np.float64).T
        ah, al = split(a)
        wh, wl = split(w)
        pairs = ah @ wh.T + (ah @ wl.T + al @ wh.T) / 2048.0
        scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
        e_pairs = np.abs(pairs - exact) / scale
        e_f32 = np.abs((a @ w.T).astype(np.float64) - exact) / scale
        assert e_pairs.max() < 2.0e-7, (k, e_pairs.max())                   # 3 x 2^-22 = 7e-7 is the proven bound
        assert np.sqrt((e_pairs ** 2).mean()) < np.sqrt((e_f32 ** 2).mean()), k
    # the representation itself: 22 bits, down through the subnormal range
    x = (10.0 ** rng.uniform(-9, 4, 200_000) * rng.choice([-1, 1], 200_000)).astype(np.float32)
    hi, lo = split(x)
    rel = np.abs(hi + lo / 2048.0 - x.astype(np.float64)) / np.abs(x)
    assert rel[np.abs(x) >= 2.0 ** -14].max() <= 2.0 ** -22
    assert (np.abs(hi + lo / 2048.0 - x.astype(np.float64))[np.abs(x) < 2.0 ** -14]).max() <= 2.0 ** -36      # lo's half ulp there
