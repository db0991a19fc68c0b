"""The fp32-grade GELU of the reference-precision encoder (csrc/rr_ce_h2.hip: h2_gelu2) on the CPU: the coefficients in the
kernel source are the ones tools/fit_gelu_f32.py produces, and the formula -- evaluated in float32 with the kernel's operation
order -- stays within fp32 rounding of the erf form the reference computes (torch's `gelu`, app/app_product_search.py:277-278
through BertIntermediate)."""
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
