"""`from flow_matching_t2i import CNF` of the reference (flow_matching_t2i.py:15) -> uspace_amd.flow_matching_t2i.CNF, with
`training_losses` (flow_matching_t2i.py:86-101) delegated to the reference's own PyTorch U-ViT over this module's parameters."""
import os as _os
import sys as _sys

from uspace_amd.flow_matching_t2i import CNF as _CNF

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
try:
    import _overlay
    import _training
finally:
    _sys.path.pop(0)


class CNF(_CNF):
    __module__ = _CNF.__module__

    def training_losses(self, x, context, sigma_min, **kwargs):
        net = _training.unwrap(self.net)
        _training.note_training_step()
        twin = _training.reference_twin(net, _overlay)
        if twin is None:
            return super().training_losses(x, context, sigma_min, **kwargs)
        twin.train(net.training)                 # the script's nnet.train() / .eval() reaches the twin too
        return _training.flow_matching_loss(lambda t, xt: twin(xt, t, context=context, **kwargs)[0], x, sigma_min)

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
