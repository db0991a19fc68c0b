import numpy as np
from scipy.special import erf
from numpy.polynomial import chebyshev as C
def fit(X, deg):
    a = np.linspace(0, X, 40001)
    t = 2*a/X - 1
    tgt = 0.5*erf(a/np.sqrt(2))
    w = np.maximum(a, 0.05)        # error in gelu = |x| * dh
    c = C.chebfit(t, tgt, deg, w=w)
    return C.cheb2poly(c)
def gelu_approx(x, p, X):
    x = x.astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(X)).astype(np.float32)
    t = (a*np.float32(2/X) - np.float32(1)).astype(np.float32)
    acc = np.full_like(t, np.float32(p[-1]))
    for k in range(len(p)-2, -1, -1):
        acc = (acc*t + np.float32(p[k])).astype(np.float32)
    h = np.copysign(acc, x).astype(np.float32)
    return (x*h + np.float32(0.5)*x).astype(np.float32)
x = np.concatenate([np.linspace(-12, 12, 2_000_001), np.linspace(-0.01,0.01,20001)])
ref = 0.5*x*(1+erf(x/np.sqrt(2)))
for X in (4.0, 4.25, 4.5, 5.0):
    for deg in (8,9,10,11,12,13):
        p = fit(X, deg)
        g = gelu_approx(x, p, X)
        err = np.abs(g-ref)
        print(X, deg, "max abs err %.2e" % err.max(), "at x=%.3f" % x[err.argmax()], " abs err near 0: %.2e" % err[np.abs(x)<0.01].max(), "p0=%.2e"%p.sum() if False else "")
print("---- weighted at the clamp")
def fit2(X, deg, wx):
    a = np.concatenate([np.linspace(0, X, 40001), np.full(400, X)])
    t = 2*a/X - 1
    tgt = 0.5*erf(a/np.sqrt(2)); tgt[-400:] = 0.5
    w = np.maximum(a, 0.05); w[-400:] = wx
    c = C.chebfit(t, tgt, deg, w=w)
    return C.cheb2poly(c)
best=None
for X in (4.25, 4.5, 4.75, 5.0):
    for deg in (9,10,11,12):
        for wx in (12, 40, 120):
            p = fit2(X, deg, wx)
            g = gelu_approx(x, p, X)
            err = np.abs(g-ref)
            print(X, deg, wx, "max abs err %.2e" % err.max(), "at x=%.3f" % x[err.argmax()], " |x|<=4: %.2e" % err[np.abs(x)<=4].max())
print("==== chosen")
for (X,deg) in ((4.5,10),(4.75,11)):
    p = fit2(X, deg, 40)
    g = gelu_approx(x, p, X)
    print(X, deg, "max abs err %.3e" % np.abs(g-ref).max())
    print(", ".join("%.9ef" % np.float32(c) for c in p))
