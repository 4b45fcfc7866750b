"""ODE integrators for flow-matching sampling, stated from the published methods.

The reference delegates to ``torchdiffeq.odeint_adjoint`` (flow_matching.py:8,118,140,163,172),
an un-vendored, un-pinned dependency that is absent from the reference tree; this module
restates the algorithms it selects there (fixed grid: euler / midpoint / rk4; adaptive:
dopri5 / bosh3 / adaptive_heun) -- parity with torchdiffeq's controller is UNPINNED
(DESIGN.md §oracle).  Only the final state is returned, which is all the reference uses
(``odeint(...)[-1]``).

State arithmetic is delegated to an ``ops`` object.  The product path uses ``HipStateOps``
(uspace_ode_combine / uspace_ode_error_norm kernels on fp32 device tensors); there is no
other implementation in this package.
"""
import math

import numpy as np

# ---------------------------------------------------------------------------------------------
# Butcher tableaux (Dormand & Prince 1980; Bogacki & Shampine 1989; Heun-Euler)
# ---------------------------------------------------------------------------------------------
_DOPRI5 = dict(
    order=5,
    alpha=[1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0],
    beta=[
        [1 / 5],
        [3 / 40, 9 / 40],
        [44 / 45, -56 / 15, 32 / 9],
        [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
        [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
        [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
    ],
    c_sol=[35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0],
    c_err=[35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
           -2187 / 6784 + 12231 / 42400, 11 / 84 - 649 / 6300, -1 / 60],
    c_mid=[6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2],
)
_BOSH3 = dict(
    order=3,
    alpha=[1 / 2, 3 / 4, 1.0],
    beta=[[1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]],
    c_sol=[2 / 9, 1 / 3, 4 / 9, 0.0],
    c_err=[2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
    c_mid=[0.0, 0.5, 0.0, 0.0],
)
_HEUN = dict(
    order=2,
    alpha=[1.0],
    beta=[[1.0]],
    c_sol=[0.5, 0.5],
    c_err=[0.5, -0.5],
    c_mid=[0.5, 0.0],
)
ADAPTIVE = {"dopri5": _DOPRI5, "bosh3": _BOSH3, "adaptive_heun": _HEUN}
FIXED = ("euler", "midpoint", "rk4")

SAFETY, IFACTOR, DFACTOR = 0.9, 10.0, 0.2


class Stats:
    """Counters of one solve (NFE is what the benchmark quotes)."""

    def __init__(self):
        self.nfe = 0
        self.accepted = 0
        self.rejected = 0


def allreduce_mean_square(pair, group=None):
    """Global mean of squares from each rank's ``pair`` = [sum of squares, element count] (a 2-element fp64 tensor on the
    rank's device): ONE small all-reduce and ONE read-back; returns a host float.  It is the only collective of an
    error-controlled solve run with a process group (option B): every rank then takes the same accept /
    reject decisions and step sizes, those of a solve over the whole batch.  ``group`` None or True = the default group."""
    import torch.distributed as dist

    dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=None if group is True else group)
    tot, n = pair.tolist()
    return tot / n if n > 0 else 0.0


class HipStateOps:
    """fp32 device-tensor arithmetic through libuspace_hip.so.

    ``group``: a torch.distributed process group (or True for the default group).  With it the RMS norms that steer an
    adaptive solve are taken over the batch of ALL ranks: each rank's raw fp32 sum of squares (second output of
    uspace_ode_error_norm) and element count are all-reduced in fp64, so every rank makes the same decisions and the
    sharded solve follows the step sequence of the unsharded one up to the rounding of that sum (the single-GPU kernel adds
    the same squares in one fp32 reduction tree, the sharded form adds per-rank fp32 sums in fp64: a ratio within ~1e-7 of
    1.0 can fall on the other side).  Without a group (default) every rank controls its own steps, as the reference does
    under ``accelerate launch`` (each rank solves its own mini-batch, tools/utils_uvit.py:269-277)."""

    def __init__(self, like, group=None):
        import torch

        from . import _hip

        _hip.require_device(like, "ODE state")
        self._hip = _hip
        self._torch = torch
        self._scratch = torch.empty(1024, dtype=torch.float32, device=like.device)
        self._result = torch.zeros(2, dtype=torch.float32, device=like.device)      # [rms, sum of squares]
        self.group = group
        self._pair = torch.zeros(2, dtype=torch.float64, device=like.device) if group is not None else None

    def prepare(self, y):
        return y.detach().to(self._torch.float32).contiguous()

    def combine(self, y, ks, coefs):
        """y + sum_i coefs[i] * ks[i] into a fresh tensor."""
        out = self._torch.empty_like(y)
        if y.numel() == 0:
            return out
        return self._hip.ode_combine(out, y, ks, coefs)

    def scaled_norm(self, y0, y1, ks, coefs, rtol, atol):
        """sqrt(mean((sum_i c_i k_i / (atol + rtol*max(|y0|,|y1|)))^2)) as a host float (one sync)."""
        n = y0.numel()
        if n > 0:
            self._hip.ode_error_norm(y0, y1, ks, coefs, rtol, atol, self._scratch, self._result)
        if self.group is None:
            return float(self._result[0].item()) if n > 0 else 0.0
        # (sum of squares, count) of this rank -> global RMS.  Both slots are rewritten before every all-reduce: it leaves
        # the GLOBAL sum and count in them.
        if n > 0:
            self._pair[0].copy_(self._result[1])
        else:
            self._pair[0].zero_()
        self._pair[1].fill_(float(n))
        ms = allreduce_mean_square(self._pair, self.group)
        return math.sqrt(ms)


def _call(func, t, y, sign, stats):
    stats.nfe += 1
    f = func(sign * t, y)
    return f, sign


def fixed_grid(t0, t1, step_size):
    """Grid of the fixed-step solvers: t0 + k*step (fp32 arithmetic like the reference's
    ``torch.arange(niters) * step_size + t0`` on fp32 tensors), last point clipped to t1."""
    t0f, t1f, hf = np.float32(t0), np.float32(t1), np.float32(step_size)
    niters = int(math.ceil(float((t1f - t0f) / hf + np.float32(1.0))))
    grid = (np.arange(niters, dtype=np.float32) * hf + t0f).astype(np.float32)
    grid[-1] = t1f
    return [float(v) for v in grid]


def _fixed_step(func, method, ops, t0, t1, y, sign, stats):
    dt = t1 - t0
    k1, _ = _call(func, t0, y, sign, stats)
    if method == "euler":
        return ops.combine(y, [k1], [sign * dt])
    if method == "midpoint":
        ym = ops.combine(y, [k1], [sign * 0.5 * dt])
        k2, _ = _call(func, t0 + 0.5 * dt, ym, sign, stats)
        return ops.combine(y, [k2], [sign * dt])
    if method == "rk4":  # 3/8 rule (the fixed-grid rk4 variant torchdiffeq ships)
        y2 = ops.combine(y, [k1], [sign * dt / 3])
        k2, _ = _call(func, t0 + dt / 3, y2, sign, stats)
        y3 = ops.combine(y, [k2, k1], [sign * dt, -sign * dt / 3])
        k3, _ = _call(func, t0 + 2 * dt / 3, y3, sign, stats)
        y4 = ops.combine(y, [k1, k2, k3], [sign * dt, -sign * dt, sign * dt])
        k4, _ = _call(func, t1, y4, sign, stats)
        return ops.combine(y, [k1, k2, k3, k4], [sign * dt / 8, sign * 3 * dt / 8, sign * 3 * dt / 8, sign * dt / 8])
    raise NotImplementedError(f"unknown fixed solver {method}")


def _neg(ops, f):
    # reversed time: the solver integrates dy/ds = -f(-s, y)
    return ops.combine(f, [f], [-2.0])


def odeint(func, y0, t0, t1, *, method="dopri5", rtol=1e-5, atol=1e-5, step_size=None, n_steps=None,
           ops=None, stats=None, max_num_steps=100000):
    """Integrate dy/dt = func(t, y) from t0 to t1 and return y(t1).

    func(t: float, y) -> dy/dt (same type as y).  Reverse time (t1 < t0) integrates -func(-s, y)
    over s in [-t0, -t1], so func always sees the true time.
      method in FIXED      -> step_size grid (fixed_grid) or ``n_steps`` equal steps
      method in ADAPTIVE   -> error-controlled steps (rtol/atol), or -- with ``n_steps`` --
                              exactly n_steps equal steps of the same tableau with FSAL and no
                              rejection ("dopri5-50" of BASELINE.md: 1 + 6*n NFE)
    """
    stats = stats if stats is not None else Stats()
    grouped = getattr(ops, "group", None) is not None
    if hasattr(y0, "numel") and y0.numel() == 0 and not (grouped and method in ADAPTIVE and n_steps is None):
        # empty batch: nothing to integrate (torchdiffeq returns y0's shape).  A rank with an empty shard of a group-controlled
        # adaptive solve still walks the control loop: it owes the other ranks its (zero) share of every norm.
        if method not in FIXED and method not in ADAPTIVE:
            raise NotImplementedError(f"method={method}")
        return y0.clone()
    if ops is None:
        ops = HipStateOps(y0)
    y = ops.prepare(y0)
    sign = 1.0
    if t1 < t0:
        sign, t0, t1 = -1.0, -t0, -t1
    if t0 == t1:
        return y

    if method in FIXED:
        if n_steps is not None:
            grid = [t0 + (t1 - t0) * k / n_steps for k in range(n_steps)] + [t1]
        else:
            if step_size is None:
                raise ValueError("fixed solvers need step_size or n_steps")
            grid = fixed_grid(t0, t1, step_size)
        for a, b in zip(grid[:-1], grid[1:]):
            y = _fixed_step(func, method, ops, a, b, y, sign, stats)
        return y

    if method not in ADAPTIVE:
        raise NotImplementedError(f"unknown solver {method}")
    tab = ADAPTIVE[method]

    def f_eval(t, yy):
        f, _ = _call(func, t, yy, sign, stats)
        return f if sign == 1.0 else _neg(ops, f)

    f0 = f_eval(t0, y)
    if n_steps is not None:                       # fixed-size steps of the embedded method (FSAL)
        h = (t1 - t0) / n_steps
        t = t0
        for k in range(n_steps):
            tn = t1 if k == n_steps - 1 else t0 + (k + 1) * h
            y, f0 = _adaptive_try(func, tab, ops, t, tn - t, y, f0, sign, stats)[:2]
            t = tn
        return y

    # ---- adaptive: initial step (Hairer-Norsett-Wanner II.4), then accept/reject with a PI-free controller
    order = tab["order"]
    d0 = ops.scaled_norm(y, y, [y], [1.0], rtol, atol)
    d1 = ops.scaled_norm(y, y, [f0], [1.0], rtol, atol)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    y_probe = ops.combine(y, [f0], [h0])
    f_probe = f_eval(t0 + h0, y_probe)
    d2 = ops.scaled_norm(y, y, [f_probe, f0], [1.0, -1.0], rtol, atol) / h0
    # torchdiffeq hands `order - 1` to its initial-step heuristic (rk_common.py: _select_initial_step(..., self.order - 1,
    # ...)), whose exponent is 1 / (that + 1): 1/5 for dopri5, not Hairer's 1/(order + 1)
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / order)
    dt = min(100 * h0, h1)

    t = t0
    steps = 0
    while True:
        # step until the accepted interval [t, t+dt] covers t1, then evaluate the dense output at t1
        y1, f1, ks, ratio = _adaptive_try(func, tab, ops, t, dt, y, f0, sign, stats, rtol, atol)
        steps += 1
        if steps > max_num_steps:
            raise RuntimeError("max_num_steps exceeded")
        accept = ratio <= 1.0
        dt_next = _next_step(dt, ratio, order)
        if accept:
            stats.accepted += 1
            if t + dt >= t1:
                if t + dt == t1:
                    return y1
                return _dense_eval(tab, ops, t, dt, y, y1, f0, f1, ks, t1)
            t, y, f0 = t + dt, y1, f1
        else:
            stats.rejected += 1
        dt = dt_next


def _adaptive_try(func, tab, ops, t, dt, y, f0, sign, stats, rtol=None, atol=None):
    ks = [f0]
    yi = y
    for alpha, row in zip(tab["alpha"], tab["beta"]):
        ti = t + dt if alpha == 1.0 else t + alpha * dt
        yi = ops.combine(y, ks, [dt * b for b in row])
        f, _ = _call(func, ti, yi, sign, stats)
        ks.append(f if sign == 1.0 else _neg(ops, f))
    fsal = tab["c_sol"][-1] == 0.0 and list(tab["c_sol"][:-1]) == list(tab["beta"][-1])
    y1 = yi if fsal else ops.combine(y, ks, [dt * c for c in tab["c_sol"]])
    if rtol is None:
        return y1, ks[-1], ks
    ratio = ops.scaled_norm(y, y1, ks, [dt * c for c in tab["c_err"]], rtol, atol)
    return y1, ks[-1], ks, ratio


def _next_step(dt, ratio, order):
    if ratio == 0.0:
        return dt * IFACTOR
    dfac = 1.0 if ratio < 1.0 else DFACTOR
    factor = min(IFACTOR, max(SAFETY / ratio ** (1.0 / order), dfac))
    return dt * factor


def _dense_eval(tab, ops, t, dt, y0, y1, f0, f1, ks, t_eval):
    """Quartic Hermite-type dense output through (y0, y_mid, y1, f0, f1) evaluated at t_eval."""
    y_mid = ops.combine(y0, ks, [dt * c for c in tab["c_mid"]])
    x = (t_eval - t) / dt
    # coefficients a..e of p(x) = e + x(d + x(c + x(b + x a))) as linear forms in (f0, f1, y0, y1, y_mid)
    a = dict(f0=-2 * dt, f1=2 * dt, y0=-8.0, y1=-8.0, ym=16.0)
    b = dict(f0=5 * dt, f1=-3 * dt, y0=18.0, y1=14.0, ym=-32.0)
    c = dict(f0=-4 * dt, f1=dt, y0=-11.0, y1=-5.0, ym=16.0)
    d = dict(f0=dt, f1=0.0, y0=0.0, y1=0.0, ym=0.0)
    w = {k: x * (d[k] + x * (c[k] + x * (b[k] + x * a[k]))) for k in a}
    # result = y0 + w.y0*y0 + w.y1*y1 + ...
    return ops.combine(y0, [f0, f1, y0, y1, y_mid], [w["f0"], w["f1"], w["y0"], w["y1"], w["ym"]])
