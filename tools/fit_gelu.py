#!/usr/bin/env python3
"""Fit and check the polynomial of the fused erf-GELU epilogue (uspace_amd/csrc/common.h, US_GELU_C*).

    gelu(v) = max(v, 0) - |v| w(|v|),   w(x) = Phi(-x) = exp2(P(x)),   x = min(|v|, 8)

`python tools/fit_gelu.py 6` fits P of that degree (reweighted least squares driven towards the minimax of the error of |v| w,
weighted x / (x + 1/4)) and prints the coefficients with their fp32 error against the exact erfc form;
`python tools/fit_gelu.py --check` reads the coefficients out of common.h and prints the same error figures
(tests/test_host_logic.py holds them to the bounds the header states).  CPU only (numpy + scipy)."""
import os
import re
import sys

import numpy as np
from scipy.optimize import least_squares
from scipy.special import erfc

X_MAX = 8.0
HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "uspace_amd", "csrc", "common.h")


def fit(deg, a=0.25, iters=300):
    x = np.linspace(0, X_MAX, 40001)
    w = 0.5 * erfc(x / np.sqrt(2))
    V = np.vander(x, deg + 1, increasing=True)
    scale = (x + 1e-9) / (x + a) * (1 + a)
    w0 = x * w + 1e-12
    c = np.linalg.lstsq(V * w0[:, None], np.log2(w) * w0, rcond=None)[0]
    res = lambda cc: (np.exp2(V @ cc) - w) * scale
    best, wts = (np.inf, None), np.ones_like(x)
    for _ in range(iters):
        c = least_squares(lambda cc: res(cc) * wts, c, xtol=1e-15, ftol=1e-15, gtol=1e-15).x
        r = np.abs(res(c))
        if r.max() < best[0]:
            best = (r.max(), c.copy())
        wts = np.clip(wts * (1 + 0.5 * (r / r.max() - 0.5)), 1e-3, None)
        wts /= wts.mean()
    return best[1]


def header_coefficients():
    text = open(HEADER).read()
    found = dict(re.findall(r"#define US_GELU_C(\d)\s+\(?(-?[0-9.]+e[-+]\d+)f\)?", text))
    return [float(found[str(k)]) for k in range(len(found))]


def errors(c):
    """(max absolute error, max error / |v|) of the fp32 Horner + fma evaluation over [-12, 12] and 3 sigma normal draws."""
    v = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).standard_normal(1000000) * 3]).astype(np.float32)
    x = np.minimum(np.abs(v), np.float32(X_MAX))
    q = np.full_like(x, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):      # one rounding per step, as v_fma_f32
        q = (q.astype(np.float64) * x + np.float64(np.float32(c[k]))).astype(np.float32)
    g = (-np.abs(v).astype(np.float64) * np.exp2(q).astype(np.float64) + np.maximum(v, 0)).astype(np.float32)
    v64 = v.astype(np.float64)
    e = np.abs(g - v64 * 0.5 * erfc(-v64 / np.sqrt(2)))
    return float(e.max()), float((e / np.maximum(np.abs(v64), 1e-30)).max())


if __name__ == "__main__":
    if sys.argv[1:] == ["--check"]:
        c = header_coefficients()
        print(f"degree {len(c) - 1} from {os.path.relpath(HEADER)}: max |err| {errors(c)[0]:.3e}, max |err| / |v| {errors(c)[1]:.3e}")
    else:
        for deg in map(int, sys.argv[1:] or ["6"]):
            c = fit(deg)
            ea, er = errors(c)
            print(f"degree {deg}: max |err| {ea:.3e}, max |err| / |v| {er:.3e}")
            for k, ck in enumerate(c):
                print(f"#define US_GELU_C{k} ({ck:.9e}f)")
