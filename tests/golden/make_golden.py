#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the GPU box and the test
suite only ever read the committed ``*.npz`` / ``*.json`` outputs.  The reference's
source never travels: this script imports it in place (see _refshim.py) and stores
inputs, seeds, weights of the tiny configurations and the reference's outputs.

    python tests/golden/make_golden.py            # regenerates every fixture
    python tests/golden/make_golden.py --skip-large   # tiny fixtures only

What is pinned:
  tiny_u / tiny_u_cond / tiny_t2i   full state_dict + (x, t[, ctx|y]) -> out + per-stage
                                    intermediates of libs/uvit.py:306 / libs/uvit_t2i.py:308
  hooks_u                           libs/dissection.py:115 u-space hook (head/tail through the
                                    reference itself; mid through the reference's blocks with
                                    the add applied by a forward hook, because the reference's
                                    own reader raises EinopsError on [n,L,D] deltas)
  p2p_t2i                           tools/utils_t2i.py:265 attention-map hook
  attr_directions                   tools/utils_attr.py:124 mean(pos) - mean(neg) attribute directions
  vae_decoder_tiny                  libs/autoencoder.py:303-409,446-450 SD-VAE decoder, tiny config + taps
  pca_components                    tools/utils_vis.py:80-118 sklearn PCA directions for the write_pca hook
  clip_text_tiny                    libs/clip.py:40-91 CLIP text transformer (HF CLIPTextModel), tiny config + hidden states
  big_{S,L}_{u,t}                   seed-regenerated weights (sha256 pinned) -> out, B=2
  euler20_S_u                       BASELINE config 1: 20 fixed Euler steps, B=4, driven by
                                    a plain loop written here around the reference nnet
  hooked_traj                       whole HOOKED solves through the reference networks with the reference's own hooks active: tiny
                                    uncond write_attr (tail / head) over 100 Euler steps, tiny T2I p2p_rescale encode -> decode, and the
                                    50-step Euler end state of BASELINE configs[2] (U-ViT-L T2I, B = 2); which steps edited is recorded
                                    from the reference's own file reads / hook calls
  traj_L_u                          BASELINE config 2 (the headline shape) end to end at B=2: 50 fixed Euler steps and
                                    50 fixed Dormand-Prince steps (FSAL, 301 evaluations) of the reference U-ViT-L,
                                    both driven by loops written here (flow_matching.py:130-151 selects the solver;
                                    the integrator itself is torchdiffeq, absent -- the Dormand-Prince loop below uses
                                    the published tableau, checked against scipy's RK45 in tests/test_host_logic.py)
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402

WEIGHT_SEED = 1234
INPUT_SEED = 7

SHAPES = {
    "S": dict(embed_dim=512, depth=16, num_heads=8),
    "L": dict(embed_dim=1024, depth=20, num_heads=16),
}
COMMON = dict(img_size=32, patch_size=2, in_chans=4, mlp_ratio=4, qkv_bias=False, mlp_time_embed=False)


def sd_numpy(model):
    return {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}


def sd_sha256(model):
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def expand_t(tval, B):
    # flow_matching.py:30-34: 0-d t expanded to (B,) with stride 0
    return torch.tensor(tval, dtype=torch.float32).expand(B)


class Tap:
    """Forward hooks that record module outputs (and the first block's input)."""

    def __init__(self, model):
        self.store = {}
        self.handles = []
        m = model
        self._pre(m.in_blocks[0], "tok")
        for i, b in enumerate(m.in_blocks):
            self._post(b, f"in{i}")
        self._post(m.mid_block, "mid")
        for i, b in enumerate(m.out_blocks):
            self._post(b, f"out{i}")
        self._post(m.norm, "norm")
        self._post(m.decoder_pred, "dec")
        b0 = m.in_blocks[0]
        self._post(b0.norm1, "b0_norm1")
        self._post(b0.attn.qkv, "b0_qkv")
        self._post(b0.attn, "b0_attn")
        self._post(b0.mlp.fc1, "b0_fc1")
        self._post(b0.mlp, "b0_mlp")
        o0 = m.out_blocks[0]
        self._post(o0.skip_linear, "o0_skip")

    def _pre(self, mod, name):
        def fn(_m, args, kwargs=None):
            self.store[name] = args[0].detach().clone().numpy()

        self.handles.append(mod.register_forward_pre_hook(fn))

    def _post(self, mod, name):
        def fn(_m, _a, out):
            self.store[name] = out.detach().clone().numpy()

        self.handles.append(mod.register_forward_hook(fn))

    def close(self):
        for h in self.handles:
            h.remove()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# --------------------------------------------------------------------------- tiny configs
TINY = dict(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1,
            mlp_ratio=4, qkv_bias=False, mlp_time_embed=False)


def make_tiny_u(uvit):
    torch.manual_seed(WEIGHT_SEED)
    m = uvit.UViT(num_classes=-1, **TINY).eval()
    g = torch.Generator().manual_seed(INPUT_SEED)
    x = torch.randn(3, 4, 16, 16, generator=g)
    out = {f"sd/{k}": v for k, v in sd_numpy(m).items()}
    out["x"] = x.numpy()
    tvals = [0.0, 0.3, 1.0]
    out["tvals"] = np.array(tvals, np.float32)
    for i, tv in enumerate(tvals):
        tap = Tap(m)
        with torch.no_grad():
            o, aux = m(x, expand_t(tv, 3), None, edit_loc=None)
        assert aux is None
        tap.close()
        out[f"out{i}"] = o.numpy()
        if i == 1:
            for k, v in tap.store.items():
                out[f"tap/{k}"] = v
    save("tiny_u.npz", **out)
    return m, x


def make_tiny_u_cond(uvit):
    torch.manual_seed(WEIGHT_SEED + 1)
    m = uvit.UViT(num_classes=10, **TINY).eval()
    g = torch.Generator().manual_seed(INPUT_SEED)
    x = torch.randn(3, 4, 16, 16, generator=g)
    y = torch.tensor([0, 7, 3], dtype=torch.int64)
    tap = Tap(m)
    with torch.no_grad():
        o, _ = m(x, expand_t(0.55, 3), y, edit_loc=None)
    tap.close()
    out = {f"sd/{k}": v for k, v in sd_numpy(m).items()}
    out.update(x=x.numpy(), y=y.numpy(), tval=np.float32(0.55), out=o.numpy(), tok=tap.store["tok"])
    save("tiny_u_cond.npz", **out)


def make_tiny_t2i(uvit_t2i):
    torch.manual_seed(WEIGHT_SEED + 2)
    cfg = dict(TINY)
    m = uvit_t2i.UViT(clip_dim=64, num_clip_token=77, **cfg).eval()
    g = torch.Generator().manual_seed(INPUT_SEED)
    x = torch.randn(3, 4, 16, 16, generator=g)
    ctx = torch.randn(3, 77, 64, generator=g)
    out = {f"sd/{k}": v for k, v in sd_numpy(m).items()}
    out.update(x=x.numpy(), ctx=ctx.numpy())
    tvals = [0.0, 0.62]
    out["tvals"] = np.array(tvals, np.float32)
    for i, tv in enumerate(tvals):
        tap = Tap(m)
        with torch.no_grad():
            o, _ = m(x, expand_t(tv, 3), context=ctx)
        tap.close()
        out[f"out{i}"] = o.numpy()
        if i == 1:
            for k, v in tap.store.items():
                out[f"tap/{k}"] = v
    save("tiny_t2i.npz", **out)
    return m, x, ctx


# --------------------------------------------------------------------------- u-space hook
def make_hooks_u(uvit, m, x):
    """libs/dissection.py:115-186 through libs/uvit.py:313,336,349."""
    rng = np.random.default_rng(11)
    L, D = 65, 64
    out = {}
    cases = []
    with tempfile.TemporaryDirectory() as d:
        # image-shaped deltas (head / tail): [n_attr, C, H, W]; token-shaped (mid): [n_attr, L, D]
        img_attr = (rng.standard_normal((5, 4, 16, 16)) * 0.3).astype(np.float32)
        img_pca = (rng.standard_normal((4, 4, 16, 16)) * 0.3).astype(np.float32)
        tok_attr = (rng.standard_normal((5, L, D)) * 0.3).astype(np.float32)
        for ts in ("0.00", "0.20", "0.40", "0.41"):
            np.save(os.path.join(d, f"delta_{ts}.npy"), img_attr)
            np.save(os.path.join(d, f"pca4_{ts}.npy"), img_pca)
        out["img_attr"], out["img_pca"], out["tok_attr"] = img_attr, img_pca, tok_attr

        def run(**kw):
            base = dict(dissect_task="uspace_uvit", t_edit=0.4, write_path_root=d)
            base.update(kw)
            tv = base.pop("tval")
            with torch.no_grad():
                o, _ = m(x, expand_t(tv, x.shape[0]), None, **base)
            return o.numpy()

        spec = [
            dict(edit_loc="head", dissect_name="write_attr", ith_attr=2, write_scale=1.0, tval=0.2),
            dict(edit_loc="head", dissect_name="write_attr", ith_attr="1_3", write_scale=-2.0, tval=0.2),
            dict(edit_loc="tail", dissect_name="write_attr", ith_attr=4, write_scale=0.5, tval=0.2),
            dict(edit_loc="tail", dissect_name="write_attr", ith_attr="0_2_4", write_scale=1.5, tval=0.4),
            dict(edit_loc="head", dissect_name="write_pca", ith_component=3, pca_n=4, write_scale=2.0, tval=0.2),
            dict(edit_loc="tail", dissect_name="write_pca", ith_component=0, pca_n=4, write_scale=-1.0, tval=0.2),
            # skipped edits: t formats to "0.00" (libs/dissection.py:22) / t > t_edit
            dict(edit_loc="head", dissect_name="write_attr", ith_attr=2, write_scale=1.0, tval=0.004),
            dict(edit_loc="tail", dissect_name="write_attr", ith_attr=2, write_scale=1.0, tval=0.41),
            # 0.404 formats to "0.40" <= t_edit: still edited (dopri5-stage rounding case)
            dict(edit_loc="head", dissect_name="write_attr", ith_attr=2, write_scale=1.0, tval=0.404),
            # write_scale 0 == no hook
            dict(edit_loc="head", dissect_name="write_attr", ith_attr=2, write_scale=0.0, tval=0.2),
            # unknown edit_loc string is a no-op
            dict(edit_loc="nowhere", dissect_name="write_attr", ith_attr=2, write_scale=1.0, tval=0.2),
            # "every_0.2" string t_edit
            dict(edit_loc="head", dissect_name="write_attr", ith_attr=1, write_scale=1.0, tval=0.4, t_edit="every_0.2"),
        ]
        for i, s in enumerate(spec):
            out[f"case{i}"] = run(**s)
            cases.append({k: v for k, v in s.items()})

        # mid: reference reader raises on [n,L,D]; apply the documented add with a forward hook
        def mid_case(ith, scale, tv):
            sel = tok_attr[ith] if isinstance(ith, int) else np.mean(
                [tok_attr[int(a)] for a in ith.split("_")], axis=0)
            add = torch.from_numpy(sel.astype(np.float32))[None] * scale
            h = m.mid_block.register_forward_hook(lambda _m, _a, o: o + add)
            with torch.no_grad():
                o, _ = m(x, expand_t(tv, x.shape[0]), None, edit_loc=None)
            h.remove()
            return o.numpy()

        out["mid0"] = mid_case(2, 1.0, 0.2)
        out["mid1"] = mid_case("1_3", -0.5, 0.2)

        # read mode: file naming + content (libs/dissection.py:126-136)
        rd = os.path.join(d, "read")
        with torch.no_grad():
            m(x, expand_t(0.37, 3), None, edit_loc="tail", dissect_task="uspace_uvit",
              dissect_name="read", read_path_root=rd, batch_id=5)
        names = sorted(os.listdir(rd))
        assert names == ["5_0.37.npy"], names
        out["read_tail"] = np.load(os.path.join(rd, names[0]))

        # error convention: unknown dissect_name -> ValueError (libs/dissection.py:182)
        try:
            run(edit_loc="head", dissect_name="bogus", tval=0.2)
            raise SystemExit("expected ValueError")
        except ValueError:
            pass
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("hooks_u.npz", **out)


# --------------------------------------------------------------------------- attention-map hook
def make_p2p_t2i(m, x, ctx):
    """tools/utils_t2i.py:265-296 through libs/uvit_t2i.py:91-107."""
    B = x.shape[0]
    ids_a = [np.array([3, 5], dtype=np.int64), np.array([], dtype=np.int64), np.array([0, 76, 76], dtype=np.int64)]
    spec = [
        dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, tval=0.3, block_id="all",
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=3.0), ids="a"),
        dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, tval=0.3, block_id=[1, 2],
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=[0.0, 5.0, 2.5]), ids="a"),
        dict(dissect_name="local_prompt", fm_direction="decode", t_edit=0.5, tval=0.3, block_id=0,
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=4), ids="a"),
        dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, tval=0.3, block_id=None,
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=0.25), ids="a"),
        # not edited: encode direction / t above t_edit / lp_* token_dissect / multiplier 1
        dict(dissect_name="p2p", fm_direction="encode", t_edit=0.5, tval=0.3, block_id="all",
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=3.0), ids="a"),
        dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, tval=0.51, block_id="all",
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=3.0), ids="a"),
        dict(dissect_name="sampled_image_editing", fm_direction="decode", t_edit=0.5, tval=0.3, block_id="all",
             token_kwargs=dict(token_dissect="lp_anything", p2p_multiplier=3.0), ids="a"),
        dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, tval=0.3, block_id="all",
             token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=1.0), ids="a"),
    ]
    out = {"ids_a0": ids_a[0], "ids_a1": ids_a[1], "ids_a2": ids_a[2]}
    cases = []
    for i, s in enumerate(spec):
        kw = dict(s)
        tv = kw.pop("tval")
        kw.pop("ids")
        kw["target_context_ids"] = [a.copy() for a in ids_a]
        with torch.no_grad():
            o, _ = m(x, expand_t(tv, B), context=ctx, **kw)
        out[f"case{i}"] = o.numpy()
        cases.append(s)
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("p2p_t2i.npz", **out)


# --------------------------------------------------------------------------- S / L shapes
def build_big(uvit, uvit_t2i, shape, kind):
    torch.manual_seed(WEIGHT_SEED)
    if kind == "u":
        return uvit.UViT(num_classes=-1, **COMMON, **SHAPES[shape]).eval()
    return uvit_t2i.UViT(clip_dim=768, num_clip_token=77, **COMMON, **SHAPES[shape]).eval()


def make_big(uvit, uvit_t2i, timing):
    for shape in ("S", "L"):
        for kind in ("u", "t"):
            m = build_big(uvit, uvit_t2i, shape, kind)
            g = torch.Generator().manual_seed(INPUT_SEED)
            B = 2
            x = torch.randn(B, 4, 32, 32, generator=g)
            ctx = torch.randn(B, 77, 768, generator=g)
            tv = 0.35
            with torch.no_grad():
                t0 = time.perf_counter()
                if kind == "u":
                    o, _ = m(x, expand_t(tv, B), None, edit_loc=None)
                else:
                    o, _ = m(x, expand_t(tv, B), context=ctx)
                dt = time.perf_counter() - t0
            sd = m.state_dict()
            probe = {k: float(v.double().sum()) for k, v in list(sd.items())[:6]}
            meta = dict(shape=shape, kind=kind, weight_seed=WEIGHT_SEED, input_seed=INPUT_SEED, tval=tv,
                        sha256=sd_sha256(m), n_params=int(sum(p.numel() for p in m.parameters())),
                        probe_sums=probe, torch=torch.__version__)
            save(f"big_{shape}_{kind}.npz", x=x.numpy(), ctx=ctx.numpy().astype(np.float32) if kind == "t" else np.zeros(0, np.float32),
                 out=o.numpy(), meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
            timing[f"fwd_{shape}_{kind}_B2_first_call_s"] = dt
            if shape == "L" and kind == "u":
                # BASELINE config 2 CPU figure: timed B=8 forward, extrapolated linearly in B and NFE
                x8 = torch.randn(8, 4, 32, 32, generator=g)
                reps = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    m(x8, expand_t(tv, 8), None, edit_loc=None)
                    reps.append(time.perf_counter() - t0)
                timing["fwd_L_u_B8_s_min_of_3"] = float(min(reps))
                timing["cfg2_L_u_B64_dopri5_50_images_per_s_extrapolated"] = 8.0 / (min(reps) * 301)
                timing["cfg2_L_u_B64_euler50_images_per_s_extrapolated"] = 8.0 / (min(reps) * 50)
            del m


def make_euler20(uvit, timing):
    """BASELINE.json configs[0]: S-deep16, 20 Euler steps, B=4, random-init, CPU reference."""
    m = build_big(uvit, None, "S", "u")
    g = torch.Generator().manual_seed(INPUT_SEED)
    z = torch.randn(4, 4, 32, 32, generator=g)
    n = 20
    h = np.float32(1.0) / np.float32(n)
    y = z.clone()
    per = []
    with torch.no_grad():
        for k in range(n):
            tk = np.float32(k) * h
            t0 = time.perf_counter()
            v, _ = m(y, expand_t(float(tk), 4), None, edit_loc=None)
            per.append(time.perf_counter() - t0)
            y = y + float(h) * v
    save("euler20_S_u.npz", z=z.numpy(), x1=y.numpy(), n_steps=np.int32(n))
    timing["cfg1_S_u_B4_euler20_total_s"] = float(np.sum(per))
    timing["cfg1_S_u_B4_fwd_median_s"] = float(np.median(per))
    timing["cfg1_images_per_s"] = 4.0 / float(np.sum(per))


# Dormand-Prince 5(4) tableau (Dormand & Prince 1980; the same numbers as scipy.integrate.RK45.A / .B / .C)
_DP_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_A = [
    [],
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]


def make_traj_L_u(uvit, timing):
    """BASELINE.json configs[1] at B=2: the reference U-ViT-L driven through a whole solve, two ways."""
    m = build_big(uvit, None, "L", "u")
    g = torch.Generator().manual_seed(INPUT_SEED)
    B = 2
    z = torch.randn(B, 4, 32, 32, generator=g)
    n = 50
    h = np.float32(1.0) / np.float32(n)

    def f(tk, y):
        return m(y, expand_t(float(tk), B), None, edit_loc=None)[0]

    t0 = time.perf_counter()
    y = z.clone()
    for k in range(n):
        y = y + float(h) * f(np.float32(k) * h, y)
    x1_euler = y.clone()
    timing["traj_L_u_B2_euler50_total_s"] = time.perf_counter() - t0

    t0 = time.perf_counter()
    y = z.clone()
    k1 = f(np.float32(0.0), y)
    nfe = 1
    for s in range(n):
        ts = np.float32(s) * h
        ks = [k1]
        for i in range(1, 7):
            yi = y.clone()
            for j, a in enumerate(_DP_A[i]):
                if a != 0.0:
                    yi = yi + float(np.float32(h) * np.float32(a)) * ks[j]
            ti = np.float32(1.0) if (s == n - 1 and i >= 5) else ts + np.float32(_DP_C[i]) * h
            ks.append(f(ti, yi))
            nfe += 1
            if i == 6:
                y = yi          # stage 7's argument IS the 5th-order solution (FSAL)
        k1 = ks[6]
    timing["traj_L_u_B2_dopri5_50_total_s"] = time.perf_counter() - t0
    save("traj_L_u.npz", z=z.numpy(), x1_euler50=x1_euler.numpy(), x1_dopri5_50=y.numpy(), n_steps=np.int32(n), nfe_dopri5=np.int32(nfe))


# --------------------------------------------------------------------------- hooked trajectories (round 4)
def _fixed_grid(t0, t1, h):
    """torchdiffeq FixedGridODESolver._grid_constructor_from_step_size on fp32 tensors (published algorithm; the package is absent):
    niters = ceil((t1 - t0) / h + 1); grid = arange(niters) * h + t0; grid[-1] = t1.  Decreasing spans are integrated by the caller in
    reversed time (torchdiffeq's _flip: the solver sees t' = -t and f'(t', y) = -f(-t', y))."""
    t0, t1, h = torch.tensor(t0, dtype=torch.float32), torch.tensor(t1, dtype=torch.float32), torch.tensor(h, dtype=torch.float32)
    niters = torch.ceil((t1 - t0) / h + 1).item()
    grid = torch.arange(0, niters, dtype=torch.float32) * h + t0
    grid[-1] = t1
    return grid


def _euler(f, y, t0, t1, h):
    """y(t1) by fixed-grid Euler steps as torchdiffeq takes them: dt = t_{k+1} - t_k in fp32, y += dt * f(t_k, y)."""
    if t1 < t0:                                    # reversed time
        grid = _fixed_grid(-t0, -t1, h)
        for a, b in zip(grid[:-1], grid[1:]):
            y = y + (b - a) * (-f(-a, y))
        return y, len(grid) - 1
    grid = _fixed_grid(t0, t1, h)
    for a, b in zip(grid[:-1], grid[1:]):
        y = y + (b - a) * f(a, y)
    return y, len(grid) - 1


def hooked_delta_table(k, shape=(5, 4, 16, 16)):
    """The direction table of step file delta_{k/100:.2f}.npy: a different table for every time, so that a wrong file shows."""
    return (np.random.default_rng(4000 + k).standard_normal(shape) * 0.3).astype(np.float32)


def make_hooked_traj(uvit, uvit_t2i, m_u, x_u, m_t, x_t, ctx_t, timing, skip_large=False):
    """Whole hooked solves driven through the REFERENCE networks with the reference's own hooks active (VERDICT r3 task 4):
    (a) tiny uncond, libs/dissection.py:115-186 write_attr at edit_loc tail / head, Euler step 0.01 over [0, 1], t_edit 0.4 --
        which steps edit on the fp32 grid (0.29999998 -> "0.30", "0.00" never) is recorded from the reference's own file reads;
    (b) tiny T2I, tools/utils_t2i.py:265-296 p2p_rescale on block 1: encode 1 -> 0 (fm_direction "encode": never edits), then decode
        0 -> 1 of the encoded latent (fm_direction "decode": edits while "{t:.2f}" <= t_edit, t = 0.00 included), Euler step 0.05,
        flow_matching_t2i.py:105-146;
    (c) BASELINE configs[2]: U-ViT-L T2I, 50 Euler steps, B = 2, end state."""
    import importlib
    dis = importlib.import_module("libs.dissection")
    ut2i = importlib.import_module("tools.utils_t2i")
    out = {}
    # ---- (a)
    with tempfile.TemporaryDirectory() as d:
        for k in range(0, 101):
            np.save(os.path.join(d, f"delta_{k / 100:.2f}.npy"), hooked_delta_table(k))
        reads = []
        orig = dis._read_npz_bcwh

        def counting(npz_path, ith_ele, device):
            reads.append(os.path.basename(npz_path))
            return orig(npz_path=npz_path, ith_ele=ith_ele, device=device)

        dis._read_npz_bcwh = counting
        try:
            for tag, kw in (("tail", dict(edit_loc="tail", ith_attr=2, write_scale=1.0)),
                            ("head", dict(edit_loc="head", ith_attr="1_3", write_scale=-4.0))):
                reads.clear()
                kwargs = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=d, **kw)
                B = x_u.shape[0]
                with torch.no_grad():
                    f = lambda t, y: m_u(y, t.expand(B), None, **kwargs)[0]          # flow_matching.py:30-34
                    y1, n = _euler(f, x_u.clone(), 0.0, 1.0, 0.01)
                    g = lambda t, y: m_u(y, t.expand(B), None, edit_loc=None)[0]
                    y_plain, _ = _euler(g, x_u.clone(), 0.0, 1.0, 0.01)
                assert n == 100
                out[f"u_{tag}_x1"] = y1.numpy()
                out[f"u_{tag}_files"] = np.frombuffer(json.dumps(list(reads)).encode(), dtype=np.uint8)
                assert reads == [f"delta_{k / 100:.2f}.npy" for k in range(1, 41)], reads[:5]
            out["u_plain_x1"] = y_plain.numpy()
            # mid block, DOCUMENTED SEMANTICS (the reference's reader raises on [n, L, D] tables, see make_hooks_u): the reference's own
            # should_edit() and file naming decide per evaluation, the add x + table[ith] * scale is applied by a forward hook on the
            # reference's mid_block; token-shaped tables mid_{t:.2f} drawn per time like the image-shaped ones
            L_tok, D_emb = 65, 64
            mid_files = []
            cur = {}

            def mid_hook(_m, _a, o):
                digit = f"{cur['t']:.2f}"                                   # libs/dissection.py:120
                if dis.should_edit(digit, 0.4):
                    mid_files.append(f"delta_{digit}.npy")
                    k = int(round(float(digit) * 100))
                    tab = hooked_delta_table(k, (5, L_tok, D_emb))
                    sel = torch.from_numpy(np.mean([tab[1], tab[3]], axis=0).astype(np.float32))
                    return o + sel[None] * 6.0
                return o

            h = m_u.mid_block.register_forward_hook(mid_hook)
            try:
                with torch.no_grad():
                    def fm(t, y):
                        cur["t"] = t.expand(x_u.shape[0])[0].item()
                        return m_u(y, t.expand(x_u.shape[0]), None, edit_loc=None)[0]
                    y_mid, n = _euler(fm, x_u.clone(), 0.0, 1.0, 0.01)
            finally:
                h.remove()
            assert n == 100 and mid_files == [f"delta_{k / 100:.2f}.npy" for k in range(1, 41)]
            out["u_mid_x1"] = y_mid.numpy()
        finally:
            dis._read_npz_bcwh = orig
    # ---- (b)
    ids = [np.array([3, 5], dtype=np.int64), np.array([], dtype=np.int64), np.array([0, 76, 76], dtype=np.int64)]
    calls = []
    orig_p2p = ut2i._p2p_rescale

    def counting_p2p(attention_map, target_context_ids, p2p_multiplier=0):
        calls.append(1)
        return orig_p2p(attention_map, target_context_ids=target_context_ids, p2p_multiplier=p2p_multiplier)

    ut2i._p2p_rescale = counting_p2p
    try:
        B = x_t.shape[0]
        base = dict(dissect_name="p2p", t_edit=0.5, block_id=[1], token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=40.0))
        with torch.no_grad():
            def field(direction):
                return lambda t, y: m_t(y, t.expand(B), context=ctx_t, fm_direction=direction,
                                        target_context_ids=[a.copy() for a in ids], **base)[0]
            z_enc, n_enc = _euler(field("encode"), x_t.clone(), 1.0, 0.0, 0.05)
            n_calls_enc = len(calls)
            x_dec, n_dec = _euler(field("decode"), z_enc.clone(), 0.0, 1.0, 0.05)
            n_calls_dec = len(calls) - n_calls_enc
            plain = lambda t, y: m_t(y, t.expand(B), context=ctx_t)[0]
            x_dec_plain, _ = _euler(plain, z_enc.clone(), 0.0, 1.0, 0.05)
        assert (n_enc, n_dec, n_calls_enc, n_calls_dec) == (20, 20, 0, 11), (n_enc, n_dec, n_calls_enc, n_calls_dec)
        out.update(t_z_enc=z_enc.numpy(), t_x_dec=x_dec.numpy(), t_x_dec_plain=x_dec_plain.numpy(),
                   t_edit_calls=np.array([n_calls_enc, n_calls_dec], np.int32))
    finally:
        ut2i._p2p_rescale = orig_p2p
    # ---- (c)
    if not skip_large:
        m = build_big(uvit, uvit_t2i, "L", "t")
        g = torch.Generator().manual_seed(INPUT_SEED)
        B = 2
        z = torch.randn(B, 4, 32, 32, generator=g)
        ctx = torch.randn(B, 77, 768, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            x1, n = _euler(lambda t, y: m(y, t.expand(B), context=ctx)[0], z.clone(), 0.0, 1.0, 0.02)
        assert n == 50
        timing["traj_L_t_B2_euler50_total_s"] = time.perf_counter() - t0
        out.update(Lt_z=z.numpy(), Lt_ctx=ctx.numpy(), Lt_x1_euler50=x1.numpy(), Lt_sha256=np.frombuffer(sd_sha256(m).encode(), dtype=np.uint8))
        del m
        # BASELINE configs[3]: U-ViT-S-deep16 T2I, 50 Euler steps (B = 2 of the 512)
        m = build_big(uvit, uvit_t2i, "S", "t")
        with torch.no_grad():
            x1s, n = _euler(lambda t, y: m(y, t.expand(B), context=ctx)[0], z.clone(), 0.0, 1.0, 0.02)
        assert n == 50
        out.update(St_x1_euler50=x1s.numpy())
        del m
    save("hooked_traj.npz", **out)



def make_attr_directions():
    """tools/utils_attr.py:124-145 cal_delta_direction on synthetic features: mean(pos) - mean(neg) per
    attribute (the direction files the write hook consumes)."""
    import importlib
    ua = importlib.import_module("tools.utils_attr")
    rng = np.random.default_rng(5)
    N, T = 37, 3
    out = {}
    for tag, adim, fshape in (("celeba", 40, (4, 8, 8)), ("ffhq", 11, (9, 16))):
        attrs = (rng.random((N, adim)) < 0.4).astype(np.int64)
        attrs[:, 1] = 1          # an attribute with no negative example (mean of empty -> nan in the reference)
        attrs[rng.integers(0, N, 5), 2] = -1   # values other than 0/1 belong to neither side
        feats = rng.standard_normal((N, T) + fshape).astype(np.float32)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            deltas = np.concatenate([ua.cal_delta_direction(a, attrs, feats) for a in range(adim)], axis=0)
        out[f"{tag}_attrs"], out[f"{tag}_feats"], out[f"{tag}_delta"] = attrs, feats, deltas.astype(np.float32)
    try:
        ua.cal_delta_direction(0, np.zeros((4, 7), np.int64), np.zeros((4, 1, 2), np.float32))
        raise SystemExit("expected ValueError for an attribute table that is neither CelebA-40 nor FFHQ-11")
    except ValueError:
        pass
    save("attr_directions.npz", **out)


def make_vae_decoder():
    """libs/autoencoder.py:303-409 Decoder + :446-450 decode (post_quant_conv, 1/scale_factor) at a tiny
    configuration (ch=64, ch_mult (1,2), 1 res block, z 8x8 -> image 16x16), seeded default init."""
    import importlib
    ae = importlib.import_module("libs.autoencoder")
    dd = dict(double_z=True, z_channels=4, resolution=16, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2],
              num_res_blocks=1, attn_resolutions=[], dropout=0.0)
    torch.manual_seed(WEIGHT_SEED + 7)
    dec = ae.Decoder(**dd).eval()
    pq = torch.nn.Conv2d(4, 4, 1)
    g = torch.Generator().manual_seed(INPUT_SEED)
    z = torch.randn(3, 4, 8, 8, generator=g) * 0.18215
    store = {}
    hooks = []
    for name, mod in (("conv_in", dec.conv_in), ("mid1", dec.mid.block_1), ("attn", dec.mid.attn_1),
                      ("mid2", dec.mid.block_2), ("up1_b0", dec.up[1].block[0]), ("up1_b1", dec.up[1].block[1]),
                      ("up1_us", dec.up[1].upsample), ("up0_b0", dec.up[0].block[0]), ("up0_b1", dec.up[0].block[1]),
                      ("norm_out", dec.norm_out)):
        hooks.append(mod.register_forward_hook(lambda _m, _a, o, n=name: store.__setitem__(n, o.detach().numpy().copy())))
    with torch.no_grad():
        img = dec(pq(z * (1.0 / 0.18215)))
    for h in hooks:
        h.remove()
    # weights are NOT stored (5 MB): the product module's seeded init replays the same RNG consumption
    # (Decoder first, then post_quant_conv); the sha256 of the state_dict pins bit-equality
    h = hashlib.sha256()
    full = {f"decoder.{k}": v for k, v in dec.state_dict().items()}
    full.update({f"post_quant_conv.{k}": v for k, v in pq.state_dict().items()})
    for k, v in full.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().numpy()).tobytes())
    meta = dict(ddconfig=dd, weight_seed=WEIGHT_SEED + 7, sha256=h.hexdigest(), keys=list(full.keys()),
                scale_factor=0.18215, n_params=int(sum(v.numel() for v in full.values())))
    out = {f"tap/{k}": store[k] for k in ("conv_in", "attn", "up1_us", "up0_b0", "norm_out")}
    out.update(z=z.numpy(), img=img.numpy(), meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    save("vae_decoder_tiny.npz", **out)


def make_pca_components():
    """tools/utils_vis.py:80-118 get_pca_components_sklearn (sklearn PCA, svd_solver="full") on synthetic activations:
    the principal directions the ``write_pca`` hook consumes (``pca{n}_{t}.npy``, tools/utils_pca.py:13-50;
    SURVEY.md 8(f) rank 3)."""
    import importlib
    uv = importlib.import_module("tools.utils_vis")
    rng = np.random.default_rng(9)
    N, shape, n = 48, (4, 6, 6), 6
    basis = rng.standard_normal((10,) + shape).astype(np.float32)
    coef = rng.standard_normal((N, 10)).astype(np.float32) * np.linspace(5.0, 0.5, 10, dtype=np.float32)   # distinct spectrum
    feats = np.einsum("nk,kchw->nchw", coef, basis) + 0.05 * rng.standard_normal((N,) + shape).astype(np.float32) + 0.7
    comps = uv.get_pca_components_sklearn(torch.from_numpy(feats), n_components=n).numpy()
    save("pca_components.npz", feats=feats.astype(np.float32), components=comps.astype(np.float32),
         n_components=np.array(n))


def make_clip_text():
    """libs/clip.py:40-91 FrozenCLIPEmbedder.forward = HF ``CLIPTextModel(input_ids=tokens).last_hidden_state``.
    The pretrained weights and the tokenizer files are not in the image, so the fixture is the same HF module
    (transformers, version recorded below) at a tiny configuration with its own random init: state_dict + token
    ids -> last_hidden_state and the hidden state after every layer."""
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                         num_attention_heads=2, max_position_embeddings=77, hidden_act="quick_gelu")
    torch.manual_seed(WEIGHT_SEED + 9)
    m = CLIPTextModel(cfg).eval()
    for prm in m.parameters():                       # HF init leaves LayerNorm at (1, 0) and biases at 0: perturb them
        if prm.dim() == 1:
            prm.data.add_(0.05 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 1000, (3, 77), generator=g)
    ids[:, 0] = 998                                  # bos-like / eos-like ids: plain table rows here
    ids[1, 40:] = 999
    with torch.no_grad():
        out = m(input_ids=ids, output_hidden_states=True)
    arrays = {"sd/" + k: v.detach().numpy() for k, v in m.state_dict().items()}
    for i, h in enumerate(out.hidden_states):
        arrays[f"hidden/{i}"] = h.numpy()
    meta = dict(transformers=transformers.__version__, hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                num_attention_heads=2, max_position_embeddings=77, vocab_size=1000, hidden_act="quick_gelu",
                layer_norm_eps=cfg.layer_norm_eps, keys=list(m.state_dict().keys()))
    arrays.update(ids=ids.numpy().astype(np.int64), out=out.last_hidden_state.numpy(),
                  meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    save("clip_text_tiny.npz", **arrays)


class ToyWordPieceTokenizer:
    """Greedy longest-match word-piece tokenizer over a fixed vocabulary (continuations carry ``##``), with the two
    calls ``get_word_inds`` uses: ``encode(text)`` -> ``[bos] + ids + [eos]`` and ``decode([id])`` -> piece text.
    The real CLIP tokenizer files are not in the image; the helper only needs pieces whose characters add up to the
    words, which this provides deterministically.  The same class (rebuilt from the stored vocabulary) is in
    tests/test_host_logic.py."""

    def __init__(self, vocab):
        self.vocab = list(vocab)
        self.index = {p: i + 2 for i, p in enumerate(self.vocab)}          # 0 = bos, 1 = eos

    def _word(self, w):
        out, pos = [], 0
        while pos < len(w):
            for end in range(len(w), pos, -1):
                piece = w[pos:end] if pos == 0 else "##" + w[pos:end]
                if piece in self.index:
                    out.append(self.index[piece])
                    pos = end
                    break
            else:
                raise KeyError(w)
        return out

    def encode(self, text):
        return [0] + [i for w in text.split(" ") for i in self._word(w)] + [1]

    def decode(self, ids):
        return self.vocab[ids[0] - 2] if ids[0] >= 2 else ""


def make_word_inds():
    """libs/clip.py:6-27 get_word_inds (the reference's own function, imported) on a toy word-piece tokenizer:
    text + word_place -> token positions."""
    import contextlib
    import io
    import importlib
    _refshim.install()
    if _refshim.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, _refshim.REFERENCE_ROOT)
    ref_clip = importlib.import_module("libs.clip")
    letters = list("abcdefghijklmnopqrstuvwxyz")
    vocab = (["a", "photo", "of", "dog", "cat", "run", "the", "smil", "wom", "an", "old", "young", "face", "with", "glass"]
             + ["##ning", "##ing", "##an", "##es", "##s", "##er"] + letters + ["##" + c for c in letters])
    vocab = list(dict.fromkeys(vocab))
    tok = ToyWordPieceTokenizer(vocab)
    cases = []
    for text, places in [
        ("a photo of a running dog", ["running", "a", "dog", 0, 3, 5, "cat"]),
        ("the smiling woman with glasses", ["smiling", "woman", "glasses", 1, 4, "the"]),
        ("an old man and a younger woman smiling", ["younger", "and", "smiling", 2, 7, "a"]),
        ("cats", ["cats", 0]),
        ("the the the dog", ["the", 3]),
    ]:
        for wp in places:
            with contextlib.redirect_stdout(io.StringIO()):        # the reference prints its word pieces
                out = ref_clip.get_word_inds(text, wp, tok)
            cases.append(dict(text=text, word_place=wp, expected=[int(v) for v in np.asarray(out).tolist()]))
    with open(os.path.join(HERE, "word_inds.json"), "w") as f:
        json.dump(dict(vocab=vocab, cases=cases), f, indent=1)
    print("word_inds.json:", len(cases), "cases")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    ap.add_argument("--only-pca", action="store_true", help="regenerate only pca_components.npz")
    ap.add_argument("--only-traj", action="store_true", help="regenerate only traj_L_u.npz")
    ap.add_argument("--only-clip", action="store_true", help="regenerate only clip_text_tiny.npz (no reference import needed)")
    ap.add_argument("--only-word-inds", action="store_true", help="regenerate only word_inds.json")
    ap.add_argument("--only-hooked", action="store_true", help="regenerate only hooked_traj.npz")
    args = ap.parse_args()
    if args.only_hooked:
        uvit, uvit_t2i = _refshim.load_reference()
        torch.set_grad_enabled(False)
        # the tiny networks and inputs of tiny_u.npz / tiny_t2i.npz (same seeds, same construction order)
        torch.manual_seed(WEIGHT_SEED)
        m_u = uvit.UViT(num_classes=-1, **TINY).eval()
        x_u = torch.randn(3, 4, 16, 16, generator=torch.Generator().manual_seed(INPUT_SEED))
        torch.manual_seed(WEIGHT_SEED + 2)
        m_t = uvit_t2i.UViT(clip_dim=64, num_clip_token=77, **TINY).eval()
        g = torch.Generator().manual_seed(INPUT_SEED)
        x_t = torch.randn(3, 4, 16, 16, generator=g)
        ctx_t = torch.randn(3, 77, 64, generator=g)
        timing = {}
        make_hooked_traj(uvit, uvit_t2i, m_u, x_u, m_t, x_t, ctx_t, timing, skip_large=args.skip_large)
        tpath = os.path.join(HERE, "ref_cpu_timing.json")
        old = json.load(open(tpath))
        old.update(timing)
        json.dump(old, open(tpath, "w"), indent=1)
        return
    if args.only_word_inds:
        make_word_inds()
        return
    if args.only_clip:
        torch.set_grad_enabled(False)
        make_clip_text()
        return
    if args.only_traj:
        uvit, _ = _refshim.load_reference()
        torch.set_grad_enabled(False)
        timing = {}
        make_traj_L_u(uvit, timing)
        tpath = os.path.join(HERE, "ref_cpu_timing.json")
        old = json.load(open(tpath))
        old.update(timing)
        json.dump(old, open(tpath, "w"), indent=1)
        print(json.dumps(timing, indent=1))
        return
    if args.only_pca:
        _refshim.load_reference()
        make_pca_components()
        return
    uvit, uvit_t2i = _refshim.load_reference()
    torch.set_grad_enabled(False)
    m, x = make_tiny_u(uvit)
    make_hooks_u(uvit, m, x)
    make_tiny_u_cond(uvit)
    mt, xt, ctx = make_tiny_t2i(uvit_t2i)
    make_p2p_t2i(mt, xt, ctx)
    make_attr_directions()
    make_vae_decoder()
    make_pca_components()
    make_clip_text()
    make_word_inds()
    hooked_timing = {}
    make_hooked_traj(uvit, uvit_t2i, m, x, mt, xt, ctx, hooked_timing, skip_large=args.skip_large)
    if not args.skip_large:
        timing = dict(threads=torch.get_num_threads(), nproc=os.cpu_count(),
                      cpu=[l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
                      torch=torch.__version__, dtype="float32", note="reference PyTorch-CPU path, this container")
        make_big(uvit, uvit_t2i, timing)
        make_euler20(uvit, timing)
        make_traj_L_u(uvit, timing)
        timing.update(hooked_timing)
        with open(os.path.join(HERE, "ref_cpu_timing.json"), "w") as f:
            json.dump(timing, f, indent=1)
        print(json.dumps(timing, indent=1))


if __name__ == "__main__":
    main()
