#!/usr/bin/env python3
"""One-off fit of the GELU constants of superslam_amd/csrc/lg_ffn.h (kGeluQ0..4): gelu(y) = y sigmoid(y Q(y^2)), Q fitted by
Lawson-reweighted least squares to the minimax relative error against y Phi(y) (scipy).  The result is pasted into the kernel source;
tests/test_lightglue_known_answers.py re-checks the pasted constants against erf."""
import numpy as np
from scipy.special import ndtr, log_ndtr
from scipy.optimize import least_squares
def gelu_exact(v): return v * ndtr(v)
def approx(c, v):
    q = np.polyval(c[::-1], v * v)
    with np.errstate(over='ignore'):
        return v / (1.0 + np.exp(-v * q))
yy = np.concatenate([np.linspace(-12, 12, 60001)])
ge = gelu_exact(yy)
den = np.maximum(np.abs(ge), 5e-2)
best = None
for nterm in (4, 5):
    y = np.linspace(1e-3, 7, 4000); s = y*y
    g = (log_ndtr(y) - log_ndtr(-y)) / y
    c = np.linalg.lstsq(np.vander(s, nterm, increasing=True), g, rcond=None)[0]
    w = np.ones_like(yy)
    for itr in range(60):   # Lawson-style reweighting towards minimax of the relative error
        r = least_squares(lambda c: w * (approx(c, yy) - ge) / den, c, method="lm", xtol=1e-15, ftol=1e-15, max_nfev=2000)
        c = r.x
        e = np.abs(approx(c, yy) - ge) / den
        w = w * (0.5 + e / e.max()); w /= w.mean()
    e = approx(c, yy) - ge
    ss = np.linspace(0, 4000, 400001)
    qmin = np.polyval(c[::-1], ss).min()
    print(nterm, "max abs %.3e max rel(den>=0.05) %.3e  Qmin %.3f" % (np.abs(e).max(), (np.abs(e)/den).max(), qmin))
    print("   coefs:", ", ".join("%.9e" % k for k in c))
    print("   coefs * -log2(e):", ", ".join("%.9ef" % (-k*1.4426950408889634) for k in c))
