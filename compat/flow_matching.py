"""`from flow_matching import CNF` of the reference (flow_matching.py:15) -> uspace_amd.flow_matching.CNF, with `training_losses`
(flow_matching.py:88-100) delegated to the reference's own PyTorch U-ViT over this module's parameters (compat/_training.py)."""
import os as _os
import sys as _sys

from uspace_amd.flow_matching import CNF as _CNF
from uspace_amd.flow_matching import CNFBase  # noqa: F401

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
try:
    import _overlay
    import _training
finally:
    _sys.path.pop(0)


class CNF(_CNF):
    __module__ = _CNF.__module__          # the scripts (and their logs) see the class they asked for

    def training_losses(self, x, y, sigma_min, **kwargs):
        net = _training.unwrap(self.net)
        _training.note_training_step()
        twin = _training.reference_twin(net, _overlay)
        if twin is None:
            return super().training_losses(x, y, sigma_min, **kwargs)        # raises: nothing to delegate to
        twin.train(net.training)                 # the script's nnet.train() / .eval() reaches the twin too
        kwargs.setdefault("edit_loc", None)      # libs/uvit.py:313 reads it unconditionally
        return _training.flow_matching_loss(lambda t, xt: twin(xt, t, y, **kwargs)[0], x, sigma_min)

    # weights edited through `.data` (the reference's EMA update) are invisible to the packed blob: repack after training steps
    def decode(self, *a, **kw):
        _training.refresh(self.net)
        return super().decode(*a, **kw)

    def encode(self, *a, **kw):
        _training.refresh(self.net)
        return super().encode(*a, **kw)

    def decode_fixadp(self, *a, **kw):
        _training.refresh(self.net)
        return super().decode_fixadp(*a, **kw)

    sample_ode = decode


__all__ = ["CNF"]
