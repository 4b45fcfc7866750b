"""Shared machinery of the two drop-in U-ViT modules (uncond / class-cond and text-to-image).

The modules keep the reference's state_dict keys and shapes (libs/uvit.py:183-291,
libs/uvit_t2i.py:193-293 -- part of the ABI) but hold no torch compute:
``forward`` repacks the parameters once into the kernel layout and issues ONE call into
libuspace_hip.so per network evaluation.
"""
import ctypes
import os
import math

import torch
import torch.nn as nn

from .. import _hip


class ParamGroup(nn.Module):
    """A node of the parameter tree: holds Parameters and child groups, no forward."""

    def add(self, name, *shape):
        p = nn.Parameter(torch.zeros(*shape))
        self.register_parameter(name, p)
        return p

    def child(self, name):
        g = ParamGroup()
        self.add_module(name, g)
        return g


def _make_block(D, hidden, skip):
    blk = ParamGroup()
    n1 = blk.child("norm1"); n1.add("weight", D); n1.add("bias", D)
    attn = blk.child("attn")
    attn.child("qkv").add("weight", 3 * D, D)                      # qkv_bias=False in every config
    pr = attn.child("proj"); pr.add("weight", D, D); pr.add("bias", D)
    n2 = blk.child("norm2"); n2.add("weight", D); n2.add("bias", D)
    mlp = blk.child("mlp")
    f1 = mlp.child("fc1"); f1.add("weight", hidden, D); f1.add("bias", hidden)
    f2 = mlp.child("fc2"); f2.add("weight", D, hidden); f2.add("bias", D)
    if skip:
        sk = blk.child("skip_linear"); sk.add("weight", D, 2 * D); sk.add("bias", D)
    return blk


class UViTBase(nn.Module):
    """Parameter container + HIP forward.  Subclasses define the token layout and kwargs handling."""

    def __init__(self, *, img_size, patch_size, in_chans, embed_dim, depth, num_heads, mlp_ratio,
                 qkv_bias, mlp_time_embed, conv, skip, n_extra, clip_dim, time_first):
        super().__init__()
        if qkv_bias or mlp_time_embed or not conv or not skip:
            raise NotImplementedError(
                "uspace_amd implements the configurations the reference ships: qkv_bias=False, "
                "mlp_time_embed=False, conv=True, skip=True (configs/*.py)")
        if embed_dim % num_heads or embed_dim // num_heads != 64:
            raise NotImplementedError("head_dim must be 64 (every reference config: 512/8, 1024/16)")
        if depth % 2:
            raise NotImplementedError("depth must be even")
        self.num_features = self.embed_dim = embed_dim
        self.in_chans = in_chans
        self.img_size, self.patch_size, self.depth, self.num_heads = img_size, patch_size, depth, num_heads
        self.hidden = int(embed_dim * mlp_ratio)
        self.patch_dim = patch_size ** 2 * in_chans
        self.extras = 1 + n_extra
        self.num_patches = (img_size // patch_size) ** 2
        self.seq_len = self.extras + self.num_patches
        self._cfg = _hip.UvitConfig(img_size, patch_size, in_chans, embed_dim, depth, num_heads, self.hidden,
                                    n_extra, clip_dim, time_first)
        self._packed = None          # (device, versions, blob)
        self._workspace = {}         # (B, device) -> uint8 tensor, at most _MAX_WORKSPACES, least recently used first
        self._delta_cache = {}
        # Replay a captured hipGraph for plain (un-hooked) evaluations.  Off by default since round 3: one C call enqueues a whole
        # evaluation and the kernels take longer to run than to launch at every batch size, so the replay only adds its three small
        # copies into / out of the graph's static buffers (U-ViT-S at batch 4: 15.14 ms per 20-step solve against 14.84 eager).
        # ``USPACE_UVIT_GRAPH=1`` or ``net.use_graph = True`` turn it on (a host that cannot keep ahead of the GPU).
        self.use_graph = os.environ.get("USPACE_UVIT_GRAPH", "0") == "1"
        self._graphs = {}            # (B, device, blob ptr, has ctx, LN-fold mode, K-split-tail switch) -> _GraphEntry (at most _MAX_GRAPHS)

    # ------------------------------------------------------------------ parameter tree
    def _build_tree(self, extra_builder):
        D = self.embed_dim
        self.pos_embed = nn.Parameter(torch.zeros(1, self.seq_len, D))
        pe = ParamGroup()
        proj = pe.child("proj")
        proj.add("weight", D, self.in_chans, self.patch_size, self.patch_size)
        proj.add("bias", D)
        self.patch_embed = pe
        extra_builder()
        half = self.depth // 2
        self.in_blocks = nn.ModuleList([_make_block(D, self.hidden, False) for _ in range(half)])
        self.mid_block = _make_block(D, self.hidden, False)
        self.out_blocks = nn.ModuleList([_make_block(D, self.hidden, True) for _ in range(half)])
        self.norm = ParamGroup(); self.norm.add("weight", D); self.norm.add("bias", D)
        self.decoder_pred = ParamGroup(); self.decoder_pred.add("weight", self.patch_dim, D); self.decoder_pred.add("bias", self.patch_dim)
        self.final_layer = ParamGroup(); self.final_layer.add("weight", self.in_chans, self.in_chans, 3, 3); self.final_layer.add("bias", self.in_chans)

    def _blocks(self):
        return list(self.in_blocks) + [self.mid_block] + list(self.out_blocks)

    # ------------------------------------------------------------------ init (RNG-order faithful)
    @torch.no_grad()
    def _reference_init_(self, pre_block_linears):
        """Reproduce the reference's random init bit-for-bit under the same torch.manual_seed.

        The reference builds stock nn.Conv2d / nn.Linear / nn.Embedding modules (their default
        inits consume the global RNG in construction order), then draws pos_embed and re-draws
        every Linear weight with trunc_normal(std=.02) in module order, zeroing Linear biases
        (libs/uvit.py:183-300).  Throw-away modules replay the consumption; only the conv
        parameters (never re-initialised there) and the trunc-normal draws are kept.
        ``pre_block_linears``: list of (weight_param, bias_param_or_None, maker) for the modules
        registered between patch_embed and the blocks (label_emb / context_embed).
        """
        D = self.embed_dim

        def burn_linear(i, o, bias=True):
            nn.Linear(i, o, bias=bias)

        conv = nn.Conv2d(self.in_chans, D, kernel_size=self.patch_size, stride=self.patch_size)
        self.patch_embed.proj.weight.copy_(conv.weight)
        self.patch_embed.proj.bias.copy_(conv.bias)
        for _w, _b, maker in pre_block_linears:
            maker()                                       # consumes RNG exactly like the reference module
        for blk in self._blocks():
            burn_linear(D, 3 * D, bias=False)
            burn_linear(D, D)
            burn_linear(D, self.hidden)
            burn_linear(self.hidden, D)
            if hasattr(blk, "skip_linear"):
                burn_linear(2 * D, D)
        burn_linear(D, self.patch_dim)
        fin = nn.Conv2d(self.in_chans, self.in_chans, 3, padding=1)
        self.final_layer.weight.copy_(fin.weight)
        self.final_layer.bias.copy_(fin.bias)

        tn = lambda p: nn.init.trunc_normal_(p, std=0.02, a=-2.0, b=2.0)
        tn(self.pos_embed)
        for w, b, _maker in pre_block_linears:
            if w is not None:
                tn(w)
                if b is not None:
                    b.zero_()
        for blk in self._blocks():
            tn(blk.attn.qkv.weight)
            tn(blk.attn.proj.weight); blk.attn.proj.bias.zero_()
            tn(blk.mlp.fc1.weight); blk.mlp.fc1.bias.zero_()
            tn(blk.mlp.fc2.weight); blk.mlp.fc2.bias.zero_()
            if hasattr(blk, "skip_linear"):
                tn(blk.skip_linear.weight); blk.skip_linear.bias.zero_()
            for n in (blk.norm1, blk.norm2):
                n.weight.fill_(1.0); n.bias.zero_()
        tn(self.decoder_pred.weight); self.decoder_pred.bias.zero_()
        self.norm.weight.fill_(1.0); self.norm.bias.zero_()

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed"}

    # ------------------------------------------------------------------ packing
    def _canonical_params(self):
        """Parameters in the canonical order of include/uspace_hip.h."""
        ps = [self.pos_embed, self.patch_embed.proj.weight, self.patch_embed.proj.bias]
        ps += self._extra_canonical()
        for blk in self._blocks():
            if hasattr(blk, "skip_linear"):
                ps += [blk.skip_linear.weight, blk.skip_linear.bias]
            ps += [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.proj.weight, blk.attn.proj.bias,
                   blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight,
                   blk.mlp.fc2.bias]
        ps += [self.norm.weight, self.norm.bias, self.decoder_pred.weight, self.decoder_pred.bias,
               self.final_layer.weight, self.final_layer.bias]
        return ps

    def _extra_canonical(self):
        return []

    def _packed_blob(self, device):
        ps = self._canonical_params()
        versions = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is not None and self._packed[0] == device and self._packed[1] == versions:
            return self._packed[2]
        L = _hip.lib()
        cfg = self._cfg
        n = L.uspace_uvit_num_params(ctypes.byref(cfg))
        if n != len(ps):
            raise _hip.UspaceHipError(f"parameter count mismatch: module {len(ps)} vs library {n}")
        srcs = []
        for i, p in enumerate(ps):
            _hip.require_device(p, "parameter")
            want = L.uspace_uvit_param_numel(ctypes.byref(cfg), i)
            if p.numel() != want:
                raise _hip.UspaceHipError(f"parameter {i}: numel {p.numel()} != {want}")
            srcs.append(p.detach().to(torch.float32).contiguous())
        nbytes = L.uspace_uvit_weight_bytes(ctypes.byref(cfg))
        blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        arr = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
        _hip.check(L.uspace_uvit_pack_weights(ctypes.byref(cfg), arr, n, _hip.ptr(blob), nbytes, _hip.stream_ptr()),
                   "uspace_uvit_pack_weights")
        _hip.sync_current_stream()   # srcs may be temporaries
        self._packed = (device, versions, blob)
        return blob

    def invalidate_packed(self):
        """Forget the packed bf16 weight blob (and the hipGraphs captured over it): the next forward repacks from the
        parameters.  Needed only after IN-PLACE edits through ``p.data`` (``p.data.copy_(w)``, ``p.data.mul_()``), which
        change neither the parameter's version counter nor its storage -- the two things ``_packed_blob`` watches;
        ``load_state_dict``, ``.to()``, optimiser steps and plain in-place ops on the parameter are picked up by itself."""
        self._packed = None
        for ent in self._graphs.values():
            ent.destroy()
        self._graphs = {}

    repack = invalidate_packed

    _MAX_WORKSPACES = 2

    def _workspace_for(self, B, device):
        """Workspace of ``uspace_uvit_workspace_bytes(B)`` bytes.  The two most recently used batch sizes stay resident
        (least recently used goes first): the ``write_scales`` sweep of BASELINE config 5 alternates one B x n_scales solve
        with plain B solves (tools/utils_vis.py:189-198), and one slot would reallocate hundreds of MB at every switch."""
        key = (B, str(device))
        ws = self._workspace.pop(key, None)
        # (asked every time: a host-side query; the size depends on the library's process-wide switches -- uspace_gemm_set_sk --, and a
        # workspace sized under another setting must not be handed on)
        nbytes = _hip.lib().uspace_uvit_workspace_bytes(ctypes.byref(self._cfg), B)
        if ws is None or ws.numel() < nbytes:
            while len(self._workspace) >= self._MAX_WORKSPACES:
                self._workspace.pop(next(iter(self._workspace)))          # dicts keep insertion order: the oldest use
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._workspace[key] = ws            # most recently used last
        return ws

    # ------------------------------------------------------------------ hipGraph replay (plain evaluations)
    _MAX_GRAPHS = 4

    def _graph_entry(self, B, dev, blob, context):
        # a captured graph replays the launch sequence of the LayerNorm mode it was captured in
        key = (B, str(dev), blob.data_ptr(), context is not None, _hip.lib().uspace_uvit_get_ln_fold(), _hip.lib().uspace_gemm_get_sk())
        ent = self._graphs.get(key)
        if ent is not None:
            return ent
        stale = [k for k in self._graphs if k[2] != blob.data_ptr()]       # weights were repacked
        for k in stale:
            self._graphs.pop(k).destroy()
        while len(self._graphs) >= self._MAX_GRAPHS:
            self._graphs.pop(next(iter(self._graphs))).destroy()
        L = _hip.lib()
        ent = _GraphEntry()
        ent.x = torch.empty(B, self.in_chans, self.img_size, self.img_size, dtype=torch.float32, device=dev)
        ent.t = torch.zeros(1, dtype=torch.float32, device=dev)
        ent.ctx = torch.empty_like(context) if context is not None else None
        ent.out = torch.empty_like(ent.x)
        nbytes = L.uspace_uvit_workspace_bytes(ctypes.byref(self._cfg), B)
        ent.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ent.blob = blob
        ent.io = _hip.UvitIO(_hip.ptr(ent.x), _hip.ptr(ent.t), 0, _hip.ptr(ent.ctx), None, 0.0, None, None,
                             _hip.ptr(ent.out), None)
        cur = torch.cuda.current_stream(dev)
        cap = torch.cuda.Stream(device=dev)            # capture needs a real (non-NULL) stream
        ent.x.zero_()
        if ent.ctx is not None:
            ent.ctx.zero_()
        cap.wait_stream(cur)
        handle = ctypes.c_void_p()
        rc = L.uspace_uvit_graph_create(ctypes.byref(self._cfg), _hip.ptr(blob), _hip.ptr(ent.ws), ent.ws.numel(),
                                        ctypes.byref(ent.io), B, ctypes.c_void_p(cap.cuda_stream), ctypes.byref(handle))
        _hip.check(rc, "uspace_uvit_graph_create")
        cap.synchronize()
        ent.handle = handle
        self._graphs[key] = ent
        return ent

    def _run_graph(self, xin, t, context, B, dev, out_dtype):
        blob = self._packed_blob(dev)
        ent = self._graph_entry(B, dev, blob, context)
        ent.x.copy_(xin, non_blocking=True)
        ent.t.copy_(t.reshape(-1)[:1], non_blocking=True)
        if context is not None:
            ent.ctx.copy_(context, non_blocking=True)
        _hip.check(_hip.lib().uspace_uvit_graph_launch(ent.handle, _hip.stream_ptr()), "uspace_uvit_graph_launch")
        out = ent.out.clone()                     # the caller owns its result (solvers keep several alive)
        return out if out_dtype == torch.float32 else out.to(out_dtype)

    # ------------------------------------------------------------------ the single HIP call
    def _run(self, x, timesteps, context=None, mid_delta=None, mid_scale=0.0, mid_tap=None, key_scale=None,
             mid_row_scale=None, keep_f32=False):
        """``keep_f32``: return the fp32 result even for a half-precision ``x`` (the caller still has fp32 work to do)."""
        _hip.require_device(x, "x")
        if x.dim() != 4 or x.shape[1] != self.in_chans or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"x must be [B,{self.in_chans},{self.img_size},{self.img_size}], got {tuple(x.shape)}")
        B = x.shape[0]
        dev = x.device
        if B == 0:                                  # empty batch: the reference returns an empty prediction
            return torch.empty(0, self.in_chans, self.img_size, self.img_size,
                               dtype=torch.float32 if keep_f32 else x.dtype, device=dev)
        xin = x.detach().to(torch.float32).contiguous()
        t = timesteps
        if not torch.is_tensor(t):
            t = torch.tensor(float(t), dtype=torch.float32, device=dev)
        t = t.detach().to(device=dev, dtype=torch.float32)
        if t.dim() == 0:
            t_stride = 0
        else:
            if t.shape[0] != B:
                raise ValueError(f"timesteps must have {B} entries, got {tuple(t.shape)}")
            t_stride = t.stride(0)
            if t_stride not in (0, 1):
                t = t.contiguous(); t_stride = 1
        plain = mid_delta is None and mid_tap is None and key_scale is None
        if self.use_graph and plain and t_stride == 0:
            return self._run_graph(xin, t, context, B, dev, torch.float32 if keep_f32 else x.dtype)
        out = torch.empty(B, self.in_chans, self.img_size, self.img_size, dtype=torch.float32, device=dev)
        blob = self._packed_blob(dev)
        ws = self._workspace_for(B, dev)
        io = _hip.UvitIO(_hip.ptr(xin), _hip.ptr(t), t_stride, _hip.ptr(context), _hip.ptr(mid_delta),
                         float(mid_scale), _hip.ptr(mid_tap), _hip.ptr(key_scale), _hip.ptr(out),
                         _hip.ptr(mid_row_scale))
        _hip.check(_hip.lib().uspace_uvit_forward(ctypes.byref(self._cfg), _hip.ptr(blob), _hip.ptr(ws), ws.numel(),
                                                  ctypes.byref(io), B, _hip.stream_ptr()), "uspace_uvit_forward")
        return out if (keep_f32 or x.dtype == torch.float32) else out.to(x.dtype)


class _GraphEntry:
    """Static buffers + instantiated hipGraph of one (batch size, weights) combination."""

    handle = None

    def destroy(self):
        if self.handle is not None:
            _hip.lib().uspace_uvit_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def host_timestep(timesteps, kwargs):
    """Value of timesteps[0] on the host.  Our own solver passes it as ``_t_host`` (no device
    sync); a foreign caller pays the same ``.item()`` the reference pays (libs/dissection.py:120)."""
    th = kwargs.get("_t_host")
    if th is not None:
        return float(th)
    if torch.is_tensor(timesteps):
        return float(timesteps.reshape(-1)[0].item())
    return float(timesteps)


def timestep_digit(t_host):
    import numpy as np
    return f"{float(np.float32(t_host)):.2f}"
