"""fp32 CPU restatement of the reference U-ViT forward and its two editing hooks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations are relative to
/root/reference.  Pinned against tests/golden/{tiny_u,tiny_u_cond,tiny_t2i,hooks_u,
p2p_t2i,big_*}.npz, which were produced by importing the reference itself
(tests/golden/make_golden.py); tests/test_oracle_golden.py is the pin.

The heavy operators are plain C (oracle/ops.c); the data flow is stated here in numpy.
"""
import os

import numpy as np

from . import _cops as C


# ----------------------------------------------------------------------------- config
class UViTSpec:
    """Shape parameters of libs/uvit.py:183-202 / libs/uvit_t2i.py:193-211."""

    def __init__(self, img_size=32, patch_size=2, in_chans=4, embed_dim=1024, depth=20, num_heads=16,
                 mlp_ratio=4, num_classes=-1, t2i=False, clip_dim=768, num_clip_token=77):
        self.img_size, self.patch_size, self.in_chans = img_size, patch_size, in_chans
        self.D, self.depth, self.H = embed_dim, depth, num_heads
        self.hidden = int(embed_dim * mlp_ratio)
        self.num_classes = num_classes
        self.t2i, self.clip_dim, self.num_clip_token = t2i, clip_dim, num_clip_token
        self.n_patch = (img_size // patch_size) ** 2
        if t2i:
            self.extras = 1 + num_clip_token            # libs/uvit_t2i.py:236
        else:
            self.extras = 2 if num_classes > 0 else 1   # libs/uvit.py:225-232
        self.L = self.extras + self.n_patch

    def block_names(self):
        n = self.depth // 2
        return [f"in_blocks.{i}" for i in range(n)] + ["mid_block"] + [f"out_blocks.{i}" for i in range(n)]


def flops_per_sample(spec):
    """SURVEY.md §8(d) algorithmic FLOPs per sample per NFE."""
    L, D = spec.L, spec.D
    nb, ns = spec.depth + 1, spec.depth // 2
    f = 2 * L * D * D * (12 * nb + 2 * ns) + 4 * L * L * D * nb
    f += 2 * spec.n_patch * 16 * D + 2 * L * D * 16 + 2 * 4 * 4 * 9 * 1024
    if spec.t2i:
        f += 2 * spec.num_clip_token * spec.clip_dim * D
    return f


# ----------------------------------------------------------------------------- hooks
def _timestep_digit(t):
    # libs/dissection.py:120 / tools/utils_t2i.py:270:  f"{timesteps[0].item():.2f}"
    return f"{float(np.float32(t)):.2f}"


def should_edit(timestep_digit, t_edit):
    """libs/dissection.py:21-34."""
    if timestep_digit == "0.00":
        return False
    if isinstance(t_edit, (float, int)):
        return float(timestep_digit) <= t_edit
    if isinstance(t_edit, str) and t_edit.startswith("every_"):
        return float(timestep_digit) % float(t_edit.replace("every_", "")) == 0.0
    raise ValueError(t_edit)


def select_delta(table, ith):
    """libs/dissection.py:55-70: int -> row; "a_b_c" -> mean of rows; leading 1 added."""
    table = np.asarray(table, np.float32)
    if isinstance(ith, (int, np.integer)):
        return table[int(ith)][None]
    if isinstance(ith, str):
        ids = [int(s) for s in ith.split("_")]
        acc = np.zeros_like(table[0])
        for i in ids:
            acc = acc + table[i]
        return (acc / np.float32(len(ids)))[None]
    raise TypeError(ith)


def uspace_hook(x, t, kw):
    """dissect_helper_uvit, libs/dissection.py:115-186.  x: [B,C,H,W] (head/tail) or [B,L,D] (mid)."""
    if kw.get("dissect_task") != "uspace_uvit":
        return x
    name = kw.get("dissect_name")
    digit = _timestep_digit(t)
    if name == "read":
        root = kw.get("read_path_root")
        os.makedirs(root, exist_ok=True)
        np.save(os.path.join(root, f"{kw['batch_id']}_{digit}"), np.asarray(x))
        return x
    if name in ("write_attr", "write_pca"):
        if not should_edit(digit, kw.get("t_edit")):
            return x
        if name == "write_attr":
            fn, ith = f"delta_{digit}.npy", kw.get("ith_attr")
        else:
            fn, ith = f"pca{kw.get('pca_n')}_{digit}.npy", kw.get("ith_component")
        table = np.load(os.path.join(kw.get("write_path_root"), fn))
        return (x + select_delta(table, ith) * np.float32(kw.get("write_scale"))).astype(np.float32)
    raise ValueError(f"dissect_name should be read or write, here is {name}")


def _block_selected(target, block_id):
    """should_edit_attention_by_blockids, tools/utils_t2i.py:227-238."""
    if isinstance(target, (int, np.integer)):
        return block_id == int(target)
    if isinstance(target, (list, tuple)):
        return block_id in target
    if isinstance(target, str) and target == "all":
        return True
    if target is None:
        return True
    raise ValueError(f"unknown target_block_id {target}")


def p2p_column_scale(B, L, t, kw, block_id):
    """editing_attention_map_vit -> real_editing_attention_map_vit -> _p2p_rescale
    (tools/utils_t2i.py:265-296, 241-262, 196-224) expressed as the per-(batch,key) factor
    the post-softmax map is multiplied by (None = untouched).  Valid because the hook
    runs after softmax with no renormalisation (libs/uvit_t2i.py:101-105)."""
    name = kw.get("dissect_name")
    if name not in ("p2p", "local_prompt", "sampled_image_editing"):
        raise ValueError(f"dissect_name should be read or write, here is {name}")
    direction = kw.get("fm_direction")
    if direction == "encode":
        return None
    if direction != "decode":
        raise NotImplementedError(direction)
    if not float(_timestep_digit(t)) <= kw.get("t_edit"):
        return None
    tk = kw["token_kwargs"]
    mode = tk["token_dissect"]
    if mode == "p2p_rescale":
        if not _block_selected(kw.get("block_id"), block_id):
            return None
        ids = kw["target_context_ids"]
        mult = tk["p2p_multiplier"]
        if isinstance(mult, (int, float)):
            mult = [mult] * len(ids)
        elif not isinstance(mult, list):
            raise ValueError(f"unknown p2p_multiplier {mult}")
        cs = np.ones((B, L), np.float32)
        for b, tid in enumerate(ids):
            tid = np.asarray(tid)
            if len(tid) > 0:
                cs[b, tid.astype(np.int64) + 1] = np.float32(mult[b])  # +TIME_TOKEN_NUM; duplicates once
        return cs
    if mode.startswith("lp_"):
        return None
    raise NotImplementedError(mode)


# ----------------------------------------------------------------------------- forward
def _block(x, sd, pre, H, skip=None, colscale=None, taps=None, tag=None):
    """Block._forward, libs/uvit.py:157-162 (pre-LN; skip_linear(cat([x, skip])) first)."""
    if skip is not None:
        D = x.shape[-1]
        W = sd[pre + ".skip_linear.weight"]
        y = C.linear(x, np.ascontiguousarray(W[:, :D]), sd[pre + ".skip_linear.bias"])
        C.linear(skip, np.ascontiguousarray(W[:, D:]), None, out=y, accumulate=True)
        x = y
        if taps is not None and tag == "out0":
            taps["o0_skip"] = x.copy()
    h = C.layernorm(x, sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"])
    qkv = C.linear(h, sd[pre + ".attn.qkv.weight"], None)
    a = C.attention(qkv, H, colscale)
    a = C.linear(a, sd[pre + ".attn.proj.weight"], sd[pre + ".attn.proj.bias"])
    if taps is not None and tag == "in0":
        taps["b0_norm1"], taps["b0_qkv"], taps["b0_attn"] = h, qkv, a
    x = x + a
    h = C.layernorm(x, sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"])
    f = C.linear(h, sd[pre + ".mlp.fc1.weight"], sd[pre + ".mlp.fc1.bias"])
    if taps is not None and tag == "in0":
        taps["b0_fc1"] = f.copy()
    f = C.gelu(f)
    f = C.linear(f, sd[pre + ".mlp.fc2.weight"], sd[pre + ".mlp.fc2.bias"])
    if taps is not None and tag == "in0":
        taps["b0_mlp"] = f
    return x + f


def uvit_forward(spec, sd, x, t, y=None, context=None, taps=None, **kw):
    """UViT.forward -- libs/uvit.py:306-351 (uncond / class-cond) and libs/uvit_t2i.py:308-342.

    x [B,C,H,W] fp32; t [B] (only t[0] is consulted by the hooks, as in the reference);
    returns v [B,C,H,W].  ``taps`` (dict) receives the intermediates the golden files hold.
    """
    x = np.asarray(x, np.float32)
    t = np.broadcast_to(np.asarray(t, np.float32), (x.shape[0],)).copy()
    B = x.shape[0]
    edit_loc = kw.get("edit_loc")
    if not spec.t2i and edit_loc == "head":                                   # libs/uvit.py:313
        x = uspace_hook(x, t[0], kw)
    tok = C.patch_embed(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
    time_tok = C.timestep_embedding(t, spec.D)[:, None, :]                    # time_embed = Identity
    if spec.t2i:
        ctx = C.linear(np.asarray(context, np.float32), sd["context_embed.weight"], sd["context_embed.bias"])
        h = np.concatenate([time_tok, ctx, tok], axis=1)                      # libs/uvit_t2i.py:323
    else:
        h = np.concatenate([time_tok, tok], axis=1)                           # libs/uvit.py:322
        if y is not None:
            lab = np.asarray(sd["label_emb.weight"], np.float32)[np.asarray(y)][:, None, :]
            h = np.concatenate([lab, h], axis=1)                              # libs/uvit.py:323-326 (label first)
    h = (h + sd["pos_embed"]).astype(np.float32)
    if taps is not None:
        taps["tok"] = h.copy()

    explicit = spec.t2i and kw.get("dissect_name") in ("p2p", "local_prompt", "sampled_image_editing")
    counter = 0

    def colscale():
        nonlocal counter
        if not explicit:
            return None
        cs = p2p_column_scale(B, spec.L, t[0], kw, counter)                   # libs/uvit_t2i.py:103
        counter += 1                                                          # libs/uvit_t2i.py:107
        return cs

    n = spec.depth // 2
    skips = []
    for i in range(n):
        h = _block(h, sd, f"in_blocks.{i}", spec.H, colscale=colscale(), taps=taps, tag=f"in{i}")
        skips.append(h)
        if taps is not None:
            taps[f"in{i}"] = h.copy()
    h = _block(h, sd, "mid_block", spec.H, colscale=colscale())
    if taps is not None:
        taps["mid"] = h.copy()
    if not spec.t2i and edit_loc == "mid":                                    # libs/uvit.py:336
        h = uspace_hook(h, t[0], kw)
    for i in range(n):
        h = _block(h, sd, f"out_blocks.{i}", spec.H, skip=skips.pop(), colscale=colscale(),
                   taps=taps, tag=f"out{i}")                                  # LIFO, libs/uvit.py:340
        if taps is not None:
            taps[f"out{i}"] = h.copy()
    h = C.layernorm(h, sd["norm.weight"], sd["norm.bias"])
    if taps is not None:
        taps["norm"] = h.copy()
    h = C.linear(h, sd["decoder_pred.weight"], sd["decoder_pred.bias"])       # all L tokens, then slice
    if taps is not None:
        taps["dec"] = h.copy()
    h = np.ascontiguousarray(h[:, spec.extras:, :])                           # libs/uvit.py:345
    img = C.unpatchify(h, spec.in_chans)
    img = C.conv3x3(img, sd["final_layer.weight"], sd["final_layer.bias"])
    if not spec.t2i and edit_loc == "tail":                                   # libs/uvit.py:349
        img = uspace_hook(img, t[0], kw)
    return img
