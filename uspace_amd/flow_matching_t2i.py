"""ODE driver for text-conditioned sampling (reference: flow_matching_t2i.py:15-175).

Differences from the unconditional driver, as in the reference: the condition is passed by
keyword (``context=``), ``fm_direction`` is recorded in the kwargs ("encode" / "decode") for the
attention-map edit, and ``encode`` follows ``get_ode_kwargs`` instead of forcing the fixed solver.
"""
from .flow_matching import CNFBase, Stats


class CNF(CNFBase):
    def forward(self, t, x, context=None, **kwargs):
        ts, th = self._timesteps(t, x)
        if th is not None and "_t_host" not in kwargs:
            kwargs = dict(kwargs, _t_host=th)
        pred, _aux = self.net(x, ts, context=context, **kwargs)   # flow_matching_t2i.py:31
        return pred

    def _velocity(self, t, x, cond, kwargs):
        return self.forward(t, x, context=cond, **kwargs)

    def encode(self, x, context=None, **kwargs):
        kwargs.update({"fm_direction": "encode"})                 # flow_matching_t2i.py:107
        self.last_stats = Stats()
        func = lambda t, xx: self._velocity(t, xx, context, kwargs)
        okw = self.get_ode_kwargs(**kwargs)
        if isinstance(okw, tuple):                                # "fixadp" has no single kwargs set
            raise TypeError("encode() needs solver 'fixed' or 'adaptive' (get_ode_kwargs returned a pair)")
        sk = kwargs.get("solver_kwargs") or {}
        return self._integrate(func, x, 1.0, 0.0, okw, sk.get("n_steps") if hasattr(sk, "get") else None)

    def decode(self, z, context=None, **kwargs):
        kwargs.update({"fm_direction": "decode"})                 # flow_matching_t2i.py:130
        return self._solve(context, z, 0.0, 1.0, kwargs)

    def decode_fixadp(self, z, context, t_mid, **kwargs):
        self.last_stats = Stats()
        func = lambda t, xx: self._velocity(t, xx, context, kwargs)
        return self._fixadp(func, z, 0.0, t_mid, 1.0, kwargs)

    sample_ode = decode
