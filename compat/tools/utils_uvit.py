"""`tools.utils_uvit` of the reference with `get_nnet` (tools/utils_uvit.py:27-41) answering "uvit" / "uvit_t2i" from
uspace_amd.  Everything else of the reference's module (set_logger, TrainState, sample2dir, ...) is re-exported from the
shadowed file when the reference tree is on the path."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_sys.path.insert(0, _os.path.dirname(_here))
try:
    import _overlay
finally:
    _sys.path.pop(0)

_ref = _overlay.load_shadowed(__name__, __file__)
if _ref is not None:
    globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})

from uspace_amd.tools.utils_uvit import amortize as _amd_amortize  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet as _amd_get_nnet  # noqa: E402


def get_nnet(name, **kwargs):
    if name in ("uvit", "uvit_t2i"):
        return _amd_get_nnet(name, **kwargs)
    if _ref is not None:
        return _ref.get_nnet(name, **kwargs)         # "unet_t2i": the reference's own SD UNet
    return _amd_get_nnet(name, **kwargs)             # raises NotImplementedError like the reference


if _ref is None:
    amortize = _amd_amortize
