"""Drop-in for the reference's unconditional / class-conditional U-ViT (libs/uvit.py:182-351).

    nnet(x, timesteps, y=None, **kwargs) -> (pred, None)

Same constructor keywords (dummies accepted), same state_dict keys, same kwargs contract
(``edit_loc`` + the u-space hook keys).  The velocity field is computed by one call into
libuspace_hip.so; the u-space hook's add runs on the GPU as well.
"""
import torch
import torch.nn as nn

from .. import _hip
from . import dissection
from ._uvit_core import ParamGroup, UViTBase, host_timestep, timestep_digit


class UViT(UViTBase):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=False, qk_scale=None, norm_layer=nn.LayerNorm, mlp_time_embed=False,
                 num_classes=-1, use_checkpoint=False, conv=True, skip=True, use_latent1d=0,
                 latent_1d_pooling=False):
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not used by any reference config")
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm")
        self.num_classes = num_classes
        has_label = num_classes > 0
        super().__init__(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                         depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                         mlp_time_embed=mlp_time_embed, conv=conv, skip=skip,
                         n_extra=1 if has_label else 0, clip_dim=0, time_first=0)
        self.latent_1d = use_latent1d
        self.use_checkpoint = bool(use_checkpoint)    # no effect on sampling (no backward here); kept for compat/_training.py's twin

        def extras():
            if has_label:
                self.label_emb = ParamGroup()
                self.label_emb.add("weight", num_classes, embed_dim)

        self._build_tree(extras)
        pre = []
        if has_label:
            def make_label():
                emb = nn.Embedding(num_classes, embed_dim)   # N(0,1) default init is what the reference keeps
                with torch.no_grad():
                    self.label_emb.weight.copy_(emb.weight)
            pre.append((None, None, make_label))
        self._reference_init_(pre)

    def forward(self, x, timesteps, y=None, **kwargs):
        edit_loc = kwargs.get("edit_loc")          # the reference indexes kwargs["edit_loc"]; be lenient
        plan = None
        if edit_loc in ("head", "mid", "tail"):
            digit = timestep_digit(host_timestep(timesteps, kwargs))
            plan = dissection.plan_uspace_hook(digit, kwargs)
        _hip.require_device(x, "x")
        B = x.shape[0]
        dev = x.device
        label_tok = None
        if y is not None:
            if self.num_classes <= 0:
                raise ValueError("y given but the model has no label embedding (num_classes <= 0)")
            # embedding gather = memory plumbing; the token enters the HIP forward as an extra token
            label_tok = self.label_emb.weight.detach()[y.to(dev)].to(torch.float32).contiguous()
        elif self.num_classes > 0:
            raise ValueError("class-conditional model called without y")

        mid_delta, mid_scale, mid_tap = None, 0.0, None
        rows = None
        if plan is not None and plan.kind == "write" and plan.row_scales is not None:
            if plan.row_scales.size != B:
                raise ValueError(f"write_scale has {plan.row_scales.size} entries for a batch of {B}")
            rows = torch.from_numpy(plan.row_scales).to(dev)
        if plan is not None and edit_loc == "head":
            if plan.kind == "read":
                dissection.save_activation(plan.path, x, kwargs)
            else:
                delta = self._deltas().get(plan.path, plan.ith, dev, x[0].numel())
                x = _hip.add_broadcast(x.detach().to(torch.float32).clone(), delta, plan.scale, row_scale=rows)
        if plan is not None and edit_loc == "mid":
            if plan.kind == "read":
                mid_tap = torch.empty(B, self.seq_len, self.embed_dim, dtype=torch.float32, device=dev)
            else:
                mid_delta = self._deltas().get(plan.path, plan.ith, dev, self.seq_len * self.embed_dim)
                mid_scale = plan.scale
        tail_write = plan is not None and edit_loc == "tail" and plan.kind == "write"
        out = self._run(x, timesteps, context=label_tok, mid_delta=mid_delta, mid_scale=mid_scale, mid_tap=mid_tap,
                        mid_row_scale=rows if mid_delta is not None else None, keep_f32=tail_write)
        if mid_tap is not None:
            dissection.save_activation(plan.path, mid_tap, kwargs)
        if plan is not None and edit_loc == "tail":
            if plan.kind == "read":
                dissection.save_activation(plan.path, out, kwargs)
            else:
                # the add runs on the fp32 result, the cast back to a half-precision x.dtype comes last
                delta = self._deltas().get(plan.path, plan.ith, dev, out[0].numel())
                out = _hip.add_broadcast(out, delta, plan.scale, row_scale=rows)
                if out.dtype != x.dtype:
                    out = out.to(x.dtype)
        return out, None

    def _deltas(self):
        if not isinstance(self._delta_cache, dissection.DeltaCache):
            self._delta_cache = dissection.DeltaCache()
        return self._delta_cache
