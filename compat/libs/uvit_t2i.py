"""`from libs.uvit_t2i import UViT` (tools/utils_uvit.py:33 of the reference) -> the MI355X text-conditioned module."""
from uspace_amd.libs.uvit_t2i import UViT  # noqa: F401

__all__ = ["UViT"]
