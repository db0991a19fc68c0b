"""fp32-grade GELU for the reference-precision encoder (csrc/rr_ce_h2.hip: h2_gelu): one formula, no branch, one v_exp_f32.

    gelu(x) = x Phi(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2),      erfc(t) = 2^q(t),  q(0) = 0

q(t) = t * P(t), P a polynomial fitted here: weighted least squares on Chebyshev nodes, the weights iterated (Lawson) towards
the minimax of the ABSOLUTE error of |x| / 2 * erfc (the quantity that enters gelu); beyond T_MAX the argument is clamped
(erfc(T_MAX) ~ 1e-9).  The script evaluates the formula in float32 with the kernel's operation order (fused multiply-adds
emulated in float64 -> float32) and prints the maximum error against scipy's float64 erf and against torch's fp32 gelu.
    python tools/fit_gelu_f32.py [degree]"""
import sys

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

T_MAX = 4.25
DEG = 7                                                  # degree of P (q has degree DEG + 1); the kernel's


def fit(deg):
    n = 4001
    k = np.arange(n)
    u = np.cos(np.pi * (k + 0.5) / n)                    # Chebyshev nodes on [-1, 1]
    t = (u + 1) * 0.5 * T_MAX
    t = np.maximum(t, 1e-9)
    target = np.log2(erfc(t)) / t                         # P(t)
    # d(|x|/2 erfc) = |x|/2 * erfc * ln2 * t * dP, |x| = t sqrt2
    sens = t * np.sqrt(2) / 2 * erfc(t) * np.log(2) * t
    w = sens.copy()
    for _ in range(60):
        c = C.chebfit(u, target, deg, w=w)
        err = np.abs(C.chebval(u, c) - target) * sens
        w = w * (0.05 + err / err.max()) ** 0.5
        w /= w.max()
    return C.cheb2poly(c), float(err.max())


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def fma(a, b, c):
    return f32(a * b + c)                                 # float64 product and sum are exact enough for one fp32 rounding


def powers_of_t(coef_u):
    """P given in powers of u = 2 t / T_MAX - 1 (the fit's variable)  ->  in powers of t (what the kernel evaluates: measured
    as accurate in fp32 as the u form, one operation less)"""
    from numpy.polynomial import polynomial as Pn
    out = np.zeros(len(coef_u))
    base = np.array([-1.0, 2.0 / T_MAX])
    for k, c in enumerate(coef_u):
        out[:k + 1] += c * Pn.polypow(base, k)
    return out


def gelu_kernel(x, coef_t):
    """the kernel's operation order, in fp32"""
    x = f32(x)
    a = np.abs(x)
    t = np.minimum(f32(a * f32(np.float32(0.7071067811865476))), f32(T_MAX))
    p = np.full_like(t, f32(coef_t[-1]))
    for k in range(len(coef_t) - 2, -1, -1):
        p = fma(p, t, f32(coef_t[k]))
    q = f32(p * t)
    e = f32(np.exp2(q))                                   # v_exp_f32: 1 ulp
    return fma(f32(-0.5 * a), e, np.maximum(x, 0.0))


if __name__ == "__main__":
    from scipy.special import erf
    for deg in ([int(sys.argv[1])] if len(sys.argv) > 1 else [5, 6, 7, 8, 9]):
        coef_u, fit_err = fit(deg)
        coef_t = powers_of_t(coef_u)
        x = np.concatenate([np.linspace(-10, 10, 4_000_001), np.linspace(-0.02, 0.02, 40001)])
        x = f32(x)
        ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
        g = gelu_kernel(x, coef_t)
        err = np.abs(g - ref)
        scale = np.maximum(1.0, np.abs(x))
        line = f"deg {deg}: fit err {fit_err:.2e}; fp32 formula max |err| {err.max():.2e} at x = {x[err.argmax()]:.4f}; max err / max(1,|x|) {np.max(err / scale):.2e}"
        try:
            import torch
            tg = torch.nn.functional.gelu(torch.from_numpy(x.astype(np.float32))).double().numpy()
            line += f"; torch fp32 gelu itself: {np.abs(tg - ref).max():.2e}"
        except Exception:
            pass
        print(line)
        if len(sys.argv) > 1:
            print("coefficients of P in powers of t, low to high (T_MAX = %.2f):" % T_MAX)
            print(", ".join("%.9ef" % np.float32(c) for c in coef_t))
