"""numpy restatement of the ODE integrators the reference selects in torchdiffeq.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the integrator arithmetic of the reference lives in the third-party package
``torchdiffeq`` (PyPI, version not pinned: /root/reference/README.md:121 lists it bare; call
sites flow_matching.py:118,140,163,172 and flow_matching_t2i.py:115,135,158,167), which is
neither vendored in /root/reference nor installed here, and the reference holds no test or
golden vector at that boundary.  What follows restates the published algorithms (explicit
Euler / midpoint / 3-8-rule RK4 on a fixed grid; Dormand-Prince 5(4), Bogacki-Shampine 3(2)
and Heun-Euler 2(1) with the standard step-size controller: RMS error norm against
atol + rtol*max(|y0|,|y1|), factor clamp(0.9 * err^(-1/order), 0.2, 10), Hairer's initial
step, dense output evaluated at the end point) and is checked by self-consistency tests
(convergence order, closed-form fields, encode/decode round trip), not by cross-parity.

State is float32 (like the reference's latents), time is float64.
"""
import math

import numpy as np

TABLEAUX = {
    "dopri5": dict(
        order=5,
        alpha=[1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0],
        beta=[[1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
              [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
              [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
              [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
        b5=[35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0],
        b4=[1951 / 21600, 0.0, 22642 / 50085, 451 / 720, -12231 / 42400, 649 / 6300, 1 / 60],
        mid=[6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]),
    "bosh3": dict(
        order=3, alpha=[0.5, 0.75, 1.0], beta=[[0.5], [0.0, 0.75], [2 / 9, 1 / 3, 4 / 9]],
        b5=[2 / 9, 1 / 3, 4 / 9, 0.0], b4=[7 / 24, 1 / 4, 1 / 3, 1 / 8], mid=[0.0, 0.5, 0.0, 0.0]),
    "adaptive_heun": dict(
        order=2, alpha=[1.0], beta=[[1.0]], b5=[0.5, 0.5], b4=[1.0, 0.0], mid=[0.5, 0.0]),
}


def _lin(y, ks, cs):
    out = y.astype(np.float32).copy()
    for k, c in zip(ks, cs):
        out += np.float32(c) * k
    return out


def _rms(v):
    return float(np.sqrt(np.mean(np.square(v.astype(np.float32)), dtype=np.float32)))


def grid_points(t0, t1, step):
    a, b, h = np.float32(t0), np.float32(t1), np.float32(step)
    n = int(math.ceil(float((b - a) / h + np.float32(1))))
    g = (np.arange(n, dtype=np.float32) * h + a).astype(np.float32)
    g[-1] = b
    return g.astype(np.float64)


def solve(f, y0, t0, t1, method="dopri5", rtol=1e-5, atol=1e-5, step_size=None, n_steps=None, counters=None):
    """y(t1) for dy/dt = f(t, y);  t1 < t0 integrates backwards (f is always given the true time)."""
    counters = counters if counters is not None else {}
    counters.setdefault("nfe", 0)
    y = np.asarray(y0, np.float32)
    flip = t1 < t0
    if flip:
        t0, t1 = -t0, -t1

    def g(s, yy):
        counters["nfe"] += 1
        return (-f(-s, yy) if flip else f(s, yy)).astype(np.float32)

    if t0 == t1:
        return y
    if method in ("euler", "midpoint", "rk4"):
        pts = np.linspace(t0, t1, n_steps + 1) if n_steps else grid_points(t0, t1, step_size)
        for a, b in zip(pts[:-1], pts[1:]):
            h = b - a
            k1 = g(a, y)
            if method == "euler":
                y = _lin(y, [k1], [h])
            elif method == "midpoint":
                y = _lin(y, [g(a + h / 2, _lin(y, [k1], [h / 2]))], [h])
            else:
                k2 = g(a + h / 3, _lin(y, [k1], [h / 3]))
                k3 = g(a + 2 * h / 3, _lin(y, [k2, k1], [h, -h / 3]))
                k4 = g(b, _lin(y, [k1, k2, k3], [h, -h, h]))
                y = _lin(y, [k1, k2, k3, k4], [h / 8, 3 * h / 8, 3 * h / 8, h / 8])
        return y
    if method not in TABLEAUX:
        raise NotImplementedError(method)
    T = TABLEAUX[method]
    order = T["order"]
    err_c = [p - q for p, q in zip(T["b5"], T["b4"])]

    def attempt(t, h, yy, f0):
        ks = [f0]
        yi = yy
        for al, row in zip(T["alpha"], T["beta"]):
            yi = _lin(yy, ks, [h * b for b in row])
            ks.append(g(t + h if al == 1.0 else t + al * h, yi))
        fsal = T["b5"][-1] == 0.0 and list(T["b5"][:-1]) == list(T["beta"][-1])
        y1 = yi if fsal else _lin(yy, ks, [h * b for b in T["b5"]])
        return y1, ks

    f0 = g(t0, y)
    if n_steps:
        h = (t1 - t0) / n_steps
        t = t0
        for k in range(n_steps):
            tn = t1 if k == n_steps - 1 else t0 + (k + 1) * h
            y, ks = attempt(t, tn - t, y, f0)
            f0 = ks[-1]
            t = tn
        return y

    scale0 = np.float32(atol) + np.float32(rtol) * np.abs(y)
    d0, d1 = _rms(y / scale0), _rms(f0 / scale0)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = g(t0 + h0, _lin(y, [f0], [h0]))
    d2 = _rms((f1 - f0) / scale0) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / order)   # torchdiffeq passes order - 1 to its heuristic
    h = min(100 * h0, h1)
    t = t0
    counters["accepted"] = counters["rejected"] = 0
    while True:
        y1, ks = attempt(t, h, y, f0)
        err = _lin(np.zeros_like(y), ks, [h * c for c in err_c])
        tol = np.float32(atol) + np.float32(rtol) * np.maximum(np.abs(y), np.abs(y1))
        ratio = _rms(err / tol)
        if ratio == 0.0:
            h_next = h * 10.0
        else:
            lo = 1.0 if ratio < 1.0 else 0.2
            h_next = h * min(10.0, max(0.9 / ratio ** (1.0 / order), lo))
        if ratio <= 1.0:
            counters["accepted"] += 1
            if t + h >= t1:
                if t + h == t1:
                    return y1
                ym = _lin(y, ks, [h * c for c in T["mid"]])
                fa, fb = ks[0], ks[-1]
                x = (t1 - t) / h
                a = 2 * h * (fb - fa) - 8 * (y1 + y) + 16 * ym
                b = h * (5 * fa - 3 * fb) + 18 * y + 14 * y1 - 32 * ym
                c = h * (fb - 4 * fa) - 11 * y - 5 * y1 + 16 * ym
                d = h * fa
                return (y + x * (d + x * (c + x * (b + x * a)))).astype(np.float32)
            t, y, f0 = t + h, y1, ks[-1]
        else:
            counters["rejected"] += 1
        h = h_next
