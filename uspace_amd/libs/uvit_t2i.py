"""Drop-in for the reference's text-conditioned U-ViT (libs/uvit_t2i.py:192-342).

    nnet(x, timesteps, context=ctx, **kwargs) -> (pred, None)

77 CLIP context tokens are embedded by the HIP GEMM and prepended after the time token.
The prompt-to-prompt attention-map edit (dissect_name in {p2p, local_prompt,
sampled_image_editing}) is applied inside the fused attention kernel as a per-key factor.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ..tools import utils_t2i
from ._uvit_core import ParamGroup, UViTBase, host_timestep, timestep_digit


class UViT(UViTBase):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=False, qk_scale=None, norm_layer=nn.LayerNorm, mlp_time_embed=False,
                 use_checkpoint=False, clip_dim=768, num_clip_token=77, conv=True, skip=True, use_latent1d=False):
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not used by any reference config")
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm")
        if clip_dim % 64:
            raise NotImplementedError("clip_dim must be a multiple of 64 (768 in every reference config)")
        self.clip_dim, self.num_clip_token = clip_dim, num_clip_token
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                         depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                         mlp_time_embed=mlp_time_embed, conv=conv, skip=skip,
                         n_extra=num_clip_token, clip_dim=clip_dim, time_first=1)
        self.use_checkpoint = bool(use_checkpoint)    # no effect on sampling (no backward here); kept for compat/_training.py's twin

        def extras():
            self.context_embed = ParamGroup()
            self.context_embed.add("weight", embed_dim, clip_dim)
            self.context_embed.add("bias", embed_dim)

        self._build_tree(extras)
        self._reference_init_([(self.context_embed.weight, self.context_embed.bias,
                                lambda: nn.Linear(clip_dim, embed_dim))])
        self._ks_cache = None

    def _extra_canonical(self):
        return [self.context_embed.weight, self.context_embed.bias]

    def forward(self, x, timesteps, context, **kwargs):
        _hip.require_device(x, "x")
        B = x.shape[0]
        dev = x.device
        if context.dim() != 3 or context.shape[0] != B or context.shape[1] != self.num_clip_token \
                or context.shape[2] != self.clip_dim:
            raise ValueError(f"context must be [{B},{self.num_clip_token},{self.clip_dim}], got {tuple(context.shape)}")
        ctx = context.detach().to(device=dev, dtype=torch.float32).contiguous()   # libs/uvit_t2i.py:318
        key_scale = None
        if utils_t2i.uses_attention_edit_path(kwargs):
            digit = timestep_digit(host_timestep(timesteps, kwargs))
            table = utils_t2i.key_scale_table(self.depth + 1, B, self.seq_len, digit, kwargs)
            if table is not None:
                key_scale = self._device_table(table, dev)
        out = self._run(x, timesteps, context=ctx, key_scale=key_scale)
        return out, None

    def _device_table(self, table, dev):
        key = (table.shape, hash(table.tobytes()), str(dev))
        if self._ks_cache is None or self._ks_cache[0] != key:
            self._ks_cache = (key, torch.from_numpy(table).to(dev))
        return self._ks_cache[1]
