"""Whole-path parity on the GPU: the drop-in modules (one C-ABI call per network evaluation) against
(a) the committed golden outputs of the REFERENCE and (b) the CPU oracle, plus the hook paths.

Tolerance contract (bf16 operands, fp32 accumulation / residual stream / softmax / LayerNorm):
    single forward:  rel-L2 <= 1e-2 and max-abs <= 3e-2 * max|ref|      (the reference's
                     own fp32 -> bf16-autocast drift is 6.4e-3 / 7.3e-3 rel-L2)
"""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import uvit_oracle as O
from tests.util import load_sd, rel_l2

pytestmark = pytest.mark.gpu

REL_TOL = 1e-2
MAX_TOL = 3e-2
TINY = dict(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4,
            qkv_bias=False, mlp_time_embed=False)
COMMON = dict(img_size=32, patch_size=2, in_chans=4, mlp_ratio=4, qkv_bias=False, mlp_time_embed=False)
SHAPES = {"S": dict(embed_dim=512, depth=16, num_heads=8), "L": dict(embed_dim=1024, depth=20, num_heads=16)}


def close(got, ref, rel=REL_TOL, mx=MAX_TOL):
    got = np.asarray(got, np.float32)
    assert np.isfinite(got).all()
    r = rel_l2(got, ref)
    m = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert r <= rel and m <= mx, f"rel-L2 {r:.3e} (tol {rel}), max-abs/max {m:.3e} (tol {mx})"
    return r


def build(name, sd=None, **cfg):
    from uspace_amd.tools.utils_uvit import get_nnet
    net = get_nnet(name, **cfg)
    if sd is not None:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return net.to("cuda").eval()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda")


def expand_t(tv, B):
    return torch.tensor(float(tv), dtype=torch.float32, device="cuda").expand(B)   # stride-0, like the solver


def test_tiny_u_matches_reference_golden(golden_dir):
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(z["x"])
    for i, tv in enumerate(z["tvals"]):
        out, aux = net(x, expand_t(tv, 3), None, edit_loc=None)
        assert aux is None and out.dtype == torch.float32 and out.shape == x.shape
        close(out.cpu().numpy(), z[f"out{i}"])
    # contiguous per-row timesteps (training-style call) give the same answer as the stride-0 view
    a, _ = net(x, torch.full((3,), float(z["tvals"][1]), device="cuda"), None, edit_loc=None)
    b, _ = net(x, expand_t(z["tvals"][1], 3), None, edit_loc=None)
    assert torch.equal(a, b)
    # deterministic and input not mutated
    x0 = x.clone()
    c, _ = net(x, expand_t(z["tvals"][1], 3), None, edit_loc=None)
    assert torch.equal(b, c) and torch.equal(x, x0)
    # missing edit_loc is tolerated (the reference raises KeyError)
    d, _ = net(x, expand_t(z["tvals"][1], 3))
    assert torch.equal(b, d)


def test_per_row_timesteps_differ(golden_dir):
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(z["x"])
    t = torch.tensor([0.0, 0.3, 1.0], device="cuda")
    out, _ = net(x, t, None, edit_loc=None)
    for i in range(3):
        close(out[i:i + 1].cpu().numpy(), z[f"out{i}"][i:i + 1], mx=5e-2)


def test_tiny_cond_and_t2i_match_reference_golden(golden_dir):
    z, sd = load_sd(golden_dir, "tiny_u_cond.npz")
    net = build("uvit", sd, num_classes=10, **TINY)
    out, _ = net(dev(z["x"]), expand_t(z["tval"], 3), dev(z["y"]), edit_loc=None)
    close(out.cpu().numpy(), z["out"])
    z, sd = load_sd(golden_dir, "tiny_t2i.npz")
    net = build("uvit_t2i", sd, clip_dim=64, num_clip_token=77, **TINY)
    for i, tv in enumerate(z["tvals"]):
        out, aux = net(dev(z["x"]), expand_t(tv, 3), context=torch.from_numpy(z["ctx"]))   # ctx on CPU: moved
        assert aux is None
        close(out.cpu().numpy(), z[f"out{i}"])


@pytest.mark.parametrize("shape,kind", [("S", "u"), ("S", "t"), ("L", "u"), ("L", "t")])
def test_big_shapes_match_reference_golden(golden_dir, shape, kind):
    z = np.load(os.path.join(golden_dir, f"big_{shape}_{kind}.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    torch.manual_seed(meta["weight_seed"])
    if kind == "u":
        net = build("uvit", num_classes=-1, **COMMON, **SHAPES[shape])
    else:
        net = build("uvit_t2i", clip_dim=768, num_clip_token=77, **COMMON, **SHAPES[shape])
    assert sum(p.numel() for p in net.parameters()) == meta["n_params"]
    sd = net.state_dict()
    for k, want in meta["probe_sums"].items():       # seeded init reproduces the reference's weights
        assert abs(float(sd[k].double().sum()) - want) <= 1e-6 * max(1.0, abs(want)), k
    x = dev(z["x"])
    if kind == "u":
        out, _ = net(x, expand_t(meta["tval"], 2), None, edit_loc=None)
    else:
        out, _ = net(x, expand_t(meta["tval"], 2), context=dev(z["ctx"]))
    close(out.cpu().numpy(), z["out"])


def test_uspace_hook_cases_match_reference_golden(golden_dir):
    zt, sd = load_sd(golden_dir, "tiny_u.npz")
    z = np.load(os.path.join(golden_dir, "hooks_u.npz"))
    cases = json.loads(bytes(z["cases_json"]).decode())
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(zt["x"])
    with tempfile.TemporaryDirectory() as d:
        for ts in ("0.00", "0.20", "0.40", "0.41"):
            np.save(os.path.join(d, f"delta_{ts}.npy"), z["img_attr"])
            np.save(os.path.join(d, f"pca4_{ts}.npy"), z["img_pca"])
        for i, c in enumerate(cases):
            kw = dict(dissect_task="uspace_uvit", t_edit=0.4, write_path_root=d)
            kw.update(c)
            tv = kw.pop("tval")
            out, _ = net(x, expand_t(tv, 3), None, **kw)
            close(out.cpu().numpy(), z[f"case{i}"])
        md = os.path.join(d, "mid")
        os.makedirs(md)
        np.save(os.path.join(md, "delta_0.20.npy"), z["tok_attr"])
        base = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=md,
                    edit_loc="mid")
        out, _ = net(x, expand_t(0.2, 3), None, ith_attr=2, write_scale=1.0, **base)
        close(out.cpu().numpy(), z["mid0"])
        out, _ = net(x, expand_t(0.2, 3), None, ith_attr="1_3", write_scale=-0.5, **base)
        close(out.cpu().numpy(), z["mid1"])
        plain, _ = net(x, expand_t(0.2, 3), None, edit_loc=None)
        zero, _ = net(x, expand_t(0.2, 3), None, ith_attr=2, write_scale=0.0, **base)
        assert torch.equal(plain, zero)                      # write_scale 0 == no hook, bit-exact
        # read mode at each location: file naming and payload
        for loc, want in (("tail", z["read_tail"]),):
            rd = os.path.join(d, "rd_" + loc)
            net(x, expand_t(0.37, 3), None, edit_loc=loc, dissect_task="uspace_uvit", dissect_name="read",
                read_path_root=rd, batch_id=5)
            assert sorted(os.listdir(rd)) == ["5_0.37.npy"]
            close(np.load(os.path.join(rd, "5_0.37.npy")), want)
        rd = os.path.join(d, "rd_mid")
        net(x, expand_t(0.3, 3), None, edit_loc="mid", dissect_task="uspace_uvit", dissect_name="read",
            read_path_root=rd, batch_id=0)
        close(np.load(os.path.join(rd, "0_0.30.npy")), zt["tap/mid"])
        rd = os.path.join(d, "rd_head")
        net(x, expand_t(0.3, 3), None, edit_loc="head", dissect_task="uspace_uvit", dissect_name="read",
            read_path_root=rd, batch_id=1)
        np.testing.assert_array_equal(np.load(os.path.join(rd, "1_0.30.npy")), zt["x"])
        with pytest.raises(ValueError):
            net(x, expand_t(0.2, 3), None, edit_loc="head", dissect_task="uspace_uvit", dissect_name="bogus")


def test_p2p_cases_match_reference_golden(golden_dir):
    zt, sd = load_sd(golden_dir, "tiny_t2i.npz")
    z = np.load(os.path.join(golden_dir, "p2p_t2i.npz"))
    cases = json.loads(bytes(z["cases_json"]).decode())
    net = build("uvit_t2i", sd, clip_dim=64, num_clip_token=77, **TINY)
    x, ctx = dev(zt["x"]), dev(zt["ctx"])
    ids = [z["ids_a0"], z["ids_a1"], z["ids_a2"]]
    outs = []
    for i, c in enumerate(cases):
        kw = dict(c)
        tv = kw.pop("tval")
        kw.pop("ids")
        kw["target_context_ids"] = ids
        out, _ = net(x, expand_t(tv, 3), context=ctx, **kw)
        close(out.cpu().numpy(), z[f"case{i}"])
        outs.append(out)
    plain, _ = net(x, expand_t(0.3, 3), context=ctx)
    for i in (4, 6, 7):                       # encode / lp_* / multiplier 1 == unedited, bit-exact here
        assert torch.equal(outs[i], plain)
    assert not torch.equal(outs[0], plain)
    with pytest.raises(NotImplementedError):
        net(x, expand_t(0.3, 3), context=ctx, dissect_name="p2p", fm_direction="sideways", t_edit=0.5)


def test_state_dict_roundtrip_and_repack(golden_dir):
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(z["x"])
    a, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
    with torch.no_grad():
        net.decoder_pred.bias.add_(1.0)            # in-place parameter edit must invalidate the packed blob
    b, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
    assert not torch.equal(a, b)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    c, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
    assert torch.equal(a, c)
    with pytest.raises(RuntimeError):
        net.load_state_dict({"pos_embed": torch.zeros(1, 65, 64)}, strict=True)


def test_cpu_tensor_is_rejected_loudly(golden_dir):
    from uspace_amd import _hip
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    with pytest.raises(_hip.UspaceHipError):
        net(torch.from_numpy(z["x"]), torch.zeros(3), None, edit_loc=None)


def test_hipgraph_replay_is_bit_identical_to_eager(golden_dir):
    """Plain evaluations replay a captured hipGraph; results equal the eager launch sequence bit for bit,
    across repeated calls, changing inputs / timesteps, batch-size changes and a weight update."""
    z, sd = load_sd(golden_dir, "tiny_t2i.npz")
    net = build("uvit_t2i", sd, clip_dim=64, num_clip_token=77, **TINY)
    x, ctx = dev(z["x"]), dev(z["ctx"])
    outs = {}
    for use in (False, True):
        net.use_graph = use
        res = []
        for tv in (0.1, 0.62, 0.62, 0.9):
            o, _ = net(x, expand_t(tv, 3), context=ctx)
            res.append(o)
        o2, _ = net(x[:2].contiguous(), expand_t(0.3, 2), context=ctx[:2].contiguous())     # other batch size
        o3, _ = net(x * 0.5, expand_t(0.62, 3), context=ctx)                                 # other input, same graph
        outs[use] = res + [o2, o3]
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    assert torch.equal(outs[True][1], outs[True][2]) and not torch.equal(outs[True][0], outs[True][1])
    held = outs[True][0].clone()
    net(x, expand_t(0.5, 3), context=ctx)                    # a later replay must not clobber earlier results
    assert torch.equal(held, outs[True][0])
    with torch.no_grad():
        net.decoder_pred.bias.add_(0.5)                      # repack -> stale graphs dropped, new one captured
    net.use_graph = True
    g, _ = net(x, expand_t(0.62, 3), context=ctx)
    net.use_graph = False
    e, _ = net(x, expand_t(0.62, 3), context=ctx)
    assert torch.equal(g, e) and not torch.equal(g, outs[True][1])
    # hooked evaluations stay on the eager path and still work with use_graph on
    net.use_graph = True
    ids = [z2 for z2 in (np.array([3]), np.array([], dtype=np.int64), np.array([5]))]
    h, _ = net(x, expand_t(0.3, 3), context=ctx, dissect_name="p2p", fm_direction="decode", t_edit=0.5, block_id="all",
               target_context_ids=ids, token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=3.0))
    assert bool(torch.isfinite(h).all())


def test_empty_single_and_ragged_batches(golden_dir):
    """Edge cases of the batch dimension: B = 0 (the reference returns an empty prediction), B = 1, and a batch whose
    row count is not a multiple of any tile (rows are independent: each row equals its single-row evaluation)."""
    zt, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(zt["x"])
    out0, aux = net(x[:0], expand_t(0.3, 0), None, edit_loc=None)
    assert aux is None and out0.shape == (0, 4, 16, 16)
    g = torch.Generator().manual_seed(3)
    big = torch.randn(7, 4, 16, 16, generator=g).cuda()
    net.use_graph = False
    full, _ = net(big, expand_t(0.3, 7), None, edit_loc=None)
    for i in (0, 3, 6):
        one, _ = net(big[i:i + 1].contiguous(), expand_t(0.3, 1), None, edit_loc=None)
        assert rel_l2(one.cpu().numpy(), full[i:i + 1].cpu().numpy()) < 2e-3
    from uspace_amd.flow_matching import CNF
    z = CNF(net).decode(x[:0], None, dissect_name="none", edit_loc=None,
                        solver_kwargs=dict(solver="fixed", solver_fix="euler", solver_fix_step=0.5))
    assert z.shape == (0, 4, 16, 16)


def test_layernorm_fold_switch_is_part_of_the_graph_key(golden_dir):
    """Folded and separate LayerNorm give the same forward within rounding, eagerly and through hipGraph replay, and toggling
    the mode after a graph was captured does not replay the stale graph (the mode is in the cache key)."""
    from uspace_amd import _hip
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(z["x"])
    L = _hip.lib()
    outs = {}
    try:
        for fold in (1, 0, 1, 0):                 # toggled back and forth on a warm module
            _hip.check(L.uspace_uvit_set_ln_fold(fold), "set_ln_fold")
            assert L.uspace_uvit_get_ln_fold() == fold
            for use in (True, False):
                net.use_graph = use
                o, _ = net(x, expand_t(float(z["tvals"][1]), 3), None, edit_loc=None)
                outs.setdefault((fold, use), []).append(o)
    finally:
        L.uspace_uvit_set_ln_fold(-1)
    assert L.uspace_uvit_get_ln_fold() == 1
    for fold in (0, 1):
        a, b = outs[(fold, True)], outs[(fold, False)]
        assert torch.equal(a[0], a[1]) and torch.equal(b[0], b[1]) and torch.equal(a[0], b[0])   # graph == eager, per mode
        close(a[0].cpu().numpy(), z["out1"])
    assert not torch.equal(outs[(0, True)][0], outs[(1, True)][0])      # different launch sequences, different rounding
    assert rel_l2(outs[(0, True)][0].cpu().numpy(), outs[(1, True)][0].cpu().numpy()) < 2e-3


def test_data_edits_need_invalidate_packed(golden_dir):
    """`p.data.copy_()` changes neither the version counter nor the storage: the documented contract is an explicit
    invalidate_packed(); plain in-place edits of the parameter are still picked up by themselves."""
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(z["x"])
    for use in (False, True):
        net.use_graph = use
        a, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
        net.decoder_pred.bias.data.add_(1.0)
        net.invalidate_packed()
        b, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
        assert not torch.equal(a, b)
        net.decoder_pred.bias.data.sub_(1.0)
        net.repack()
        c, _ = net(x, expand_t(0.3, 3), None, edit_loc=None)
        assert torch.equal(a, c)


def test_scalar_write_scale_types_and_half_precision_tail_hook(golden_dir):
    """write_scale as np.float32 / 0-dim tensor / 0-dim array is ONE factor (the reference multiplies by it whatever its type,
    libs/dissection.py:157); the tail hook adds on the fp32 result also for a half-precision input."""
    zt, sd = load_sd(golden_dir, "tiny_u.npz")
    z = np.load(os.path.join(golden_dir, "hooks_u.npz"))
    net = build("uvit", sd, num_classes=-1, **TINY)
    x = dev(zt["x"])
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "delta_0.20.npy"), z["img_attr"])
        base = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=d, ith_attr=1)
        for loc in ("head", "tail"):
            ref, _ = net(x, expand_t(0.2, 3), None, edit_loc=loc, write_scale=0.75, **base)
            for s in (np.float32(0.75), np.float64(0.75), torch.tensor(0.75), np.array(0.75, np.float32),
                      np.linspace(0.0, 1.5, 3)[1]):
                got, _ = net(x, expand_t(0.2, 3), None, edit_loc=loc, write_scale=s, **base)
                assert torch.equal(ref, got), (loc, type(s))
            rows, _ = net(x, expand_t(0.2, 3), None, edit_loc=loc, write_scale=[0.75, 0.75, 0.75], **base)
            assert torch.equal(ref, rows)
        ref, _ = net(x, expand_t(0.2, 3), None, edit_loc="tail", write_scale=0.75, **base)
        half, _ = net(x.half(), expand_t(0.2, 3), None, edit_loc="tail", write_scale=0.75, **base)
        assert half.dtype == torch.float16
        close(half.float().cpu().numpy(), ref.cpu().numpy(), rel=5e-3, mx=1e-2)
