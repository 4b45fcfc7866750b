"""`from libs.uvit import UViT` (tools/utils_uvit.py:29 of the reference) -> the MI355X module with the same constructor
arguments, state_dict keys and forward(x, timesteps, y=None, **kwargs) -> (pred, None)."""
from uspace_amd.libs.uvit import UViT  # noqa: F401

__all__ = ["UViT"]
