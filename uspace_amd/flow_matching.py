"""ODE driver for unconditional / class-conditional sampling (reference: flow_matching.py:15-180).

``CNF(net)`` exposes the reference's sampling entry points -- ``decode`` (noise -> data, t: 0 -> 1),
``encode`` (data -> noise, t: 1 -> 0), ``decode_fixadp`` (fixed steps up to t_edit, adaptive after)
-- over this package's own integrators (uspace_amd/odeint.py), because the reference's solver is
the external torchdiffeq.  ``sample_ode`` (the name BASELINE.json uses) aliases ``decode``.
"""
import torch
import torch.nn as nn

from .odeint import FIXED, Stats, odeint

_RTOL = 1e-5
_ATOL = 1e-5


class CNFBase(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.last_stats = None        # odeint.Stats of the most recent solve (NFE etc.)
        self.state_ops_factory = None  # None -> odeint.HipStateOps (the only implementation shipped)
        # adaptive step control over a batch sharded across ranks: None = every rank controls its own steps
        # (the reference's behaviour under `accelerate launch`); a process group, or True for the default group = one
        # all-reduced error norm per step attempt, i.e. the step sequence of the unsharded solve on every rank
        self.norm_group = None

    # ------------------------------------------------------------------ reference surface
    def is_dissection_mode(self, kwargs):
        return "dissect_name" in kwargs and kwargs["dissect_name"] is not None

    def get_ode_kwargs(self, **kwargs):
        """Same selection rule and dict shape as the reference (flow_matching.py:38-85)."""
        if not self.is_dissection_mode(kwargs):
            return dict(method="dopri5", rtol=_RTOL, atol=_ATOL, adjoint_params=())
        sk = kwargs["solver_kwargs"]
        fixed = dict(method=sk.get("solver_fix"), rtol=_RTOL, atol=_ATOL, adjoint_params=(),
                     options=dict(step_size=sk.get("solver_fix_step")))
        adaptive = dict(method=sk.get("solver_adaptive"), rtol=_RTOL, atol=_ATOL, adjoint_params=())
        if sk["solver"] == "fixed":
            return fixed
        if sk["solver"] == "adaptive":
            return adaptive
        if sk["solver"] == "fixadp":
            return fixed, adaptive
        raise NotImplementedError(f"solver={sk['solver']}")

    def training_losses(self, *args, **kwargs):
        raise NotImplementedError(
            "uspace_amd implements the sampling hot path only (forward kernels, no backward); "
            "train with the reference's PyTorch modules and load the state_dict here")

    # ------------------------------------------------------------------ integration helpers
    def _velocity(self, t, x, cond, kwargs):
        raise NotImplementedError

    def forward(self, t, x, *cond_args, **kwargs):
        raise NotImplementedError

    def _timesteps(self, t, x):
        """(B,) stride-0 fp32 view of the scalar time, like ``t.expand(B)`` (flow_matching.py:33).

        READ-ONLY: on a fixed time grid (``_grid_is_fixed``, set by ``_integrate`` for the duration of its solve from the solver method / ``n_steps``) the
        device scalar behind the view is cached per (device, t) and handed out again in every later evaluation and solve -- a
        fixed-step solve visits the same times every time, and the cache saves one fill launch per evaluation.  A consumer
        that wrote into it (``t *= 1000`` in a ``_velocity`` override or a hook) would change that time for every later use;
        the reference's own ``t.expand(B)`` view has the same aliasing.  Error-controlled solves choose their own times, no
        two alike: they get a fresh scalar per call and never touch the cache, which is bounded (oldest entry out first)."""
        if torch.is_tensor(t):
            if t.numel() != 1:
                return t, None
            th = float(t.item())
        else:
            th = float(t)
        if not getattr(self, "_grid_is_fixed", False):
            return torch.full((), th, dtype=torch.float32, device=x.device).expand(x.shape[0]), th
        cache = self.__dict__.setdefault("_t_scalars", {})
        key = (x.device, th)
        ts = cache.get(key)
        if ts is None:
            while len(cache) >= self._T_SCALARS_MAX:
                cache.pop(next(iter(cache)))
            ts = cache[key] = torch.full((), th, dtype=torch.float32, device=x.device)
        return ts.expand(x.shape[0]), th

    _T_SCALARS_MAX = 512

    def _integrate(self, func, y0, t0, t1, ode_kwargs, n_steps=None):
        stats = self.last_stats if self.last_stats is not None else Stats()
        opts = ode_kwargs.get("options") or {}
        if self.state_ops_factory is not None:
            ops = self.state_ops_factory(y0)
        elif self.norm_group is not None:
            from .odeint import HipStateOps
            ops = HipStateOps(y0, group=self.norm_group)
        else:
            ops = None
        # fixed grid (euler / midpoint / rk4, or an error-controlled method run on n_steps equal steps): the times repeat.  The flag
        # holds for THIS solve only: a forward() / _velocity() call outside a solve (or a solve nested in a hook) gets fresh scalars
        prev = getattr(self, "_grid_is_fixed", False)
        self._grid_is_fixed = ode_kwargs["method"] in FIXED or n_steps is not None
        try:
            out = odeint(func, y0, float(t0), float(t1), method=ode_kwargs["method"], rtol=ode_kwargs["rtol"],
                         atol=ode_kwargs["atol"], step_size=opts.get("step_size"), n_steps=n_steps, stats=stats, ops=ops)
        finally:
            self._grid_is_fixed = prev
        self.last_stats = stats
        return out

    def _solve(self, cond, x, t0, t1, kwargs, force_fixed=False):
        self.last_stats = Stats()
        func = lambda t, xx: self._velocity(t, xx, cond, kwargs)
        sk = kwargs["solver_kwargs"]            # KeyError if absent, as in the reference
        n_steps = sk.get("n_steps") if hasattr(sk, "get") else None
        if force_fixed:
            okw = dict(method=sk["solver_fix"], rtol=_RTOL, atol=_ATOL, adjoint_params=(),
                       options=dict(step_size=sk["solver_fix_step"]))
            return self._integrate(func, x, t0, t1, okw, n_steps)
        if sk["solver"] in ("fixed", "adaptive"):
            return self._integrate(func, x, t0, t1, self.get_ode_kwargs(**kwargs), n_steps)
        if sk["solver"] == "fixadp":
            return self._fixadp(func, x, t0, kwargs["t_edit"], t1, kwargs)
        raise NotImplementedError(f"unknown solver {sk}")

    def _fixadp(self, func, z, t0, t_mid, t1, kwargs):
        assert 0 <= t_mid <= 1, f"t_mid={t_mid}"
        fixed_kw, adaptive_kw = self.get_ode_kwargs(**kwargs)
        mid = self._integrate(func, z, t0, t_mid, fixed_kw)
        return self._integrate(func, mid, t_mid, t1, adaptive_kw)


class CNF(CNFBase):
    """``CNF(net).decode(z, y, **kwargs)``; the net is called as ``net(x, t, y, **kwargs)``."""

    def forward(self, t, x, y=None, **kwargs):
        ts, th = self._timesteps(t, x)
        if th is not None and "_t_host" not in kwargs:
            kwargs = dict(kwargs, _t_host=th)
        pred, _aux = self.net(x, ts, y, **kwargs)            # nnet returns (pred, None), flow_matching.py:34
        return pred

    def _velocity(self, t, x, cond, kwargs):
        return self.forward(t, x, cond, **kwargs)

    def encode(self, x, y=None, **kwargs):
        """Data -> noise with the FIXED solver, t: 1 -> 0 (flow_matching.py:102-128)."""
        return self._solve(y, x, 1.0, 0.0, kwargs, force_fixed=True)

    def decode(self, z, y=None, **kwargs):
        """Noise -> data, t: 0 -> 1 (flow_matching.py:130-151)."""
        return self._solve(y, z, 0.0, 1.0, kwargs)

    def decode_fixadp(self, z, y, t_mid, **kwargs):
        self.last_stats = Stats()
        func = lambda t, xx: self._velocity(t, xx, y, kwargs)
        return self._fixadp(func, z, 0.0, t_mid, 1.0, kwargs)

    sample_ode = decode

    def decode_write_scales(self, z, y, write_scales, **kwargs):
        """The reference's sweep ``for write_scale in write_scales: decode(z, write_scale=...)``
        (tools/utils_vis.py:189-198, nine full solves of the same z) as ONE solve over len(scales)*B rows
        with a per-row hook scale.  Returns [n_scales, B, C, H, W].  Exactly equal to the sequential sweep
        for fixed-step solvers; adaptive solvers would share one step-size sequence across the scales."""
        n, B = len(write_scales), z.shape[0]
        rows = torch.as_tensor([float(s) for s in write_scales], dtype=torch.float32).repeat_interleave(B)
        zz = z.repeat(n, 1, 1, 1)
        yy = y.repeat(n) if torch.is_tensor(y) else y
        out = self.decode(zz, yy, **dict(kwargs, write_scale=rows))
        return out.view(n, B, *z.shape[1:])
