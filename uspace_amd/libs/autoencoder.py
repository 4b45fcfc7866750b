"""Decode-side drop-in for the reference's frozen SD KL-VAE (libs/autoencoder.py:303-409 ``Decoder``,
:412-450 ``FrozenAutoencoderKL.decode``): latents [B,4,32,32] -> images [B,3,256,256].

Same ``state_dict`` keys as the reference for ``decoder.*`` and ``post_quant_conv.*`` (the encoder half of a
checkpoint is accepted and ignored).  The arithmetic runs in libuspace_hip.so: 3x3 convolutions as 9-slab
bf16 MFMA GEMMs over a zero-bordered NHWC layout, GroupNorm+SiLU, nearest upsampling and the single-head
mid-block attention as HIP kernels; no torch compute.  SURVEY.md 8(f) rank 1.
"""
import ctypes
import json

import torch
import torch.nn as nn

from .. import _hip
from ._uvit_core import ParamGroup


def _conv(group, name, cout, cin, k):
    c = group.child(name)
    c.add("weight", cout, cin, k, k)
    c.add("bias", cout)
    return c


def _norm(group, name, ch):
    n = group.child(name)
    n.add("weight", ch)
    n.add("bias", ch)
    return n


def _resblock(cin, cout):
    b = ParamGroup()
    _norm(b, "norm1", cin)
    _conv(b, "conv1", cout, cin, 3)
    _norm(b, "norm2", cout)
    _conv(b, "conv2", cout, cout, 3)
    if cin != cout:
        _conv(b, "nin_shortcut", cout, cin, 1)
    b.cin, b.cout = cin, cout
    return b


class FrozenAutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim=4, pretrained_path=None, scale_factor=0.18215):
        super().__init__()
        dd = dict(ddconfig)
        if dd.get("attn_resolutions"):
            raise NotImplementedError("attention inside the up path is not used by the reference (attn_resolutions=[])")
        if dd.get("give_pre_end") or dd.get("tanh_out") or dd.get("use_linear_attn") or dd.get("resamp_with_conv") is False:
            raise NotImplementedError("non-default Decoder options")
        self.ch, self.ch_mult = dd["ch"], tuple(dd["ch_mult"])
        self.num_res_blocks, self.resolution = dd["num_res_blocks"], dd["resolution"]
        self.z_channels, self.out_ch = dd["z_channels"], dd["out_ch"]
        if embed_dim != self.z_channels or self.z_channels != 4 or self.out_ch != 3:
            raise NotImplementedError("embed_dim == z_channels == 4 and out_ch == 3 (the SD VAE the reference uses)")
        if self.ch % 64:
            raise NotImplementedError("ch must be a multiple of 64 (128 in the reference)")
        self.embed_dim, self.scale_factor = embed_dim, scale_factor
        nres = len(self.ch_mult)
        self.z_res = self.resolution // 2 ** (nres - 1)
        block_in = self.ch * self.ch_mult[-1]

        dec = ParamGroup()
        _conv(dec, "conv_in", block_in, self.z_channels, 3)
        mid = dec.child("mid")
        mid.add_module("block_1", _resblock(block_in, block_in))
        attn = mid.child("attn_1")
        _norm(attn, "norm", block_in)
        for n in ("q", "k", "v", "proj_out"):
            _conv(attn, n, block_in, block_in, 1)
        mid.add_module("block_2", _resblock(block_in, block_in))
        ups = [None] * nres
        for lvl in reversed(range(nres)):                       # construction order of the reference
            up = ParamGroup()
            block_out = self.ch * self.ch_mult[lvl]
            blocks = []
            for _ in range(self.num_res_blocks + 1):
                blocks.append(_resblock(block_in, block_out))
                block_in = block_out
            up.add_module("block", nn.ModuleList(blocks))
            up.add_module("attn", nn.ModuleList())
            if lvl != 0:
                _conv(up.child("upsample"), "conv", block_in, block_in, 3)
            ups[lvl] = up
        dec.add_module("up", nn.ModuleList(ups))
        _norm(dec, "norm_out", block_in)
        _conv(dec, "conv_out", self.out_ch, block_in, 3)
        self.decoder = dec
        self.post_quant_conv = ParamGroup()
        self.post_quant_conv.add("weight", self.z_channels, embed_dim, 1, 1)
        self.post_quant_conv.add("bias", self.z_channels)
        self._reference_init_()
        self._packed = None
        self._ws = {}
        if pretrained_path is not None:
            self.load_state_dict(torch.load(pretrained_path, map_location="cpu"))
        self.eval()
        self.requires_grad_(False)

    # ------------------------------------------------------------------ init / checkpoints
    def _conv_modules_in_construction_order(self):
        d = self.decoder
        yield d.conv_in
        for blk in (d.mid.block_1,):
            yield from self._block_convs(blk)
        a = d.mid.attn_1
        yield from (a.q, a.k, a.v, a.proj_out)
        yield from self._block_convs(d.mid.block_2)
        for lvl in reversed(range(len(self.ch_mult))):
            for blk in d.up[lvl].block:
                yield from self._block_convs(blk)
            if lvl != 0:
                yield d.up[lvl].upsample.conv
        yield d.conv_out
        yield self.post_quant_conv

    @staticmethod
    def _block_convs(blk):
        yield blk.conv1
        yield blk.conv2
        if hasattr(blk, "nin_shortcut"):
            yield blk.nin_shortcut

    @torch.no_grad()
    def _reference_init_(self):
        """torch's default Conv2d init in the reference's construction order (Decoder, then post_quant_conv),
        so the same ``torch.manual_seed`` yields the same weights; GroupNorm affine = (1, 0)."""
        for c in self._conv_modules_in_construction_order():
            cout, cin, k, _ = c.weight.shape
            ref = nn.Conv2d(cin, cout, k, padding=k // 2)
            c.weight.copy_(ref.weight)
            c.bias.copy_(ref.bias)
        for name, p in self.named_parameters():       # GroupNorm affine defaults
            if name.split(".")[-2].startswith("norm"):
                p.fill_(1.0 if name.endswith("weight") else 0.0)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts a full autoencoder checkpoint: ``encoder.*`` / ``quant_conv.*`` entries are ignored."""
        sd = {k: v for k, v in state_dict.items() if not (k.startswith("encoder.") or k.startswith("quant_conv."))}
        return super().load_state_dict(sd, strict=strict)

    # ------------------------------------------------------------------ HIP decode
    def invalidate_packed(self):
        """Forget the packed weight blob; needed only after in-place edits through ``p.data`` (same contract as
        UViT.invalidate_packed)."""
        self._packed = None

    def _packed_blob(self, device):
        ps = list(self.parameters())
        versions = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is not None and self._packed[0] == device and self._packed[1] == versions:
            return self._packed[2]
        L = _hip.lib()
        mult = (ctypes.c_int * 4)(*(list(self.ch_mult) + [0] * (4 - len(self.ch_mult))))
        self._cfg = _hip.VaeConfig(self.ch, mult, len(self.ch_mult), self.num_res_blocks, self.resolution)
        n = L.uspace_vae_num_params(ctypes.byref(self._cfg))
        if n != len(ps):
            raise _hip.UspaceHipError(f"VAE parameter count mismatch: module {len(ps)} vs library {n}")
        srcs = []
        for i, p in enumerate(ps):
            _hip.require_device(p, "parameter")
            if p.numel() != L.uspace_vae_param_numel(ctypes.byref(self._cfg), i):
                raise _hip.UspaceHipError(f"VAE parameter {i}: unexpected size {tuple(p.shape)}")
            srcs.append(p.detach().to(torch.float32).contiguous())
        nbytes = L.uspace_vae_weight_bytes(ctypes.byref(self._cfg))
        blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
        _hip.check(L.uspace_vae_pack_weights(ctypes.byref(self._cfg), arr, n, _hip.ptr(blob), nbytes, _hip.stream_ptr()),
                   "uspace_vae_pack_weights")
        torch.cuda.current_stream().synchronize()
        self._packed = (device, versions, blob)
        return blob

    def decode(self, z, chunk=8):
        """z [B,4,h,h] (scaled latents, as produced by the sampler) -> images [B,3,R,R] fp32.  Decodes ``chunk``
        images at a time (the reference chunks by 50, dissect_lfm.py:86-98)."""
        _hip.require_device(z, "z")
        if z.dim() != 4 or z.shape[1] != self.z_channels or z.shape[2] != self.z_res or z.shape[3] != self.z_res:
            raise ValueError(f"z must be [B,{self.z_channels},{self.z_res},{self.z_res}], got {tuple(z.shape)}")
        dev = z.device
        if z.shape[0] == 0:
            return torch.empty(0, self.out_ch, self.resolution, self.resolution, dtype=z.dtype, device=dev)
        blob = self._packed_blob(dev)
        L = _hip.lib()
        zin = z.detach().to(torch.float32).contiguous()
        B = zin.shape[0]
        out = torch.empty(B, self.out_ch, self.resolution, self.resolution, dtype=torch.float32, device=dev)
        max_chunk = max(1, ((1 << 30) - 1) // ((self.resolution + 2) ** 2 * 512))
        chunk = max(1, min(chunk, max_chunk, B))
        key = (chunk, str(dev))
        if key not in self._ws:
            nbytes = L.uspace_vae_workspace_bytes(ctypes.byref(self._cfg), chunk)
            self._ws = {key: torch.empty(nbytes, dtype=torch.uint8, device=dev)}
        ws = self._ws[key]
        for lo in range(0, B, chunk):
            n = min(chunk, B - lo)
            _hip.check(L.uspace_vae_decode(ctypes.byref(self._cfg), _hip.ptr(blob), _hip.ptr(ws), ws.numel(),
                                           _hip.ptr(zin[lo:lo + n]), float(self.scale_factor), _hip.ptr(out[lo:lo + n]),
                                           n, _hip.stream_ptr()), "uspace_vae_decode")
        return out if z.dtype == torch.float32 else out.to(z.dtype)

    def decode_tap(self, z, stage):
        """Test aid: the fp32 feature map after ``stage`` (see uspace_vae_decode_tap) as [B, C, H, W]."""
        dev = z.device
        blob = self._packed_blob(dev)
        L = _hip.lib()
        zin = z.detach().to(torch.float32).contiguous()
        B = zin.shape[0]
        nbytes = L.uspace_vae_workspace_bytes(ctypes.byref(self._cfg), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dump = torch.zeros(B * (self.resolution + 2) ** 2 * 512, dtype=torch.float32, device=dev)
        hc = (ctypes.c_int * 2)()
        _hip.check(L.uspace_vae_decode_tap(ctypes.byref(self._cfg), _hip.ptr(blob), _hip.ptr(ws), ws.numel(), _hip.ptr(zin),
                                           float(self.scale_factor), B, int(stage), _hip.ptr(dump), hc, _hip.stream_ptr()),
                   "uspace_vae_decode_tap")
        torch.cuda.synchronize()
        H, C = hc[0], hc[1]
        m = dump[: B * (H + 2) * (H + 2) * C].view(B, H + 2, H + 2, C)
        return m[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).contiguous()

    def forward(self, inputs, fn):
        if fn == "decode":
            return self.decode(inputs)
        raise NotImplementedError(f"{fn}: only the decode side is implemented on the MI355X path")


def get_model(pretrained_path, scale_factor=0.18215):
    """The SD KL-f8 autoencoder the reference samples through (libs/autoencoder.py:463-476): 256^2 images,
    4x32x32 latents, ch=128, multipliers 1-2-4-4, two res blocks per level, no attention in the up path."""
    sd_vae = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                  ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    return FrozenAutoencoderKL(sd_vae, 4, pretrained_path, scale_factor)
