"""BASELINE.json full-size configurations on the GPU.  The CPU oracle cannot finish these in seconds, so
they are checked through size-independent properties: sampled rows against the oracle, batch-slice
invariance (trajectories are independent), determinism, hook algebra, encode/decode round trip."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import _cops as C
from tests.util import bf16_round, rel_l2

pytestmark = pytest.mark.gpu

COMMON = dict(img_size=32, patch_size=2, in_chans=4, mlp_ratio=4, qkv_bias=False, mlp_time_embed=False)
L_CFG = dict(embed_dim=1024, depth=20, num_heads=16)
S_CFG = dict(embed_dim=512, depth=16, num_heads=8)


@pytest.fixture(scope="module")
def net_L_u():
    from uspace_amd.tools.utils_uvit import get_nnet
    torch.manual_seed(1234)
    return get_nnet("uvit", num_classes=-1, **COMMON, **L_CFG).cuda().eval()


def _z(B, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 4, 32, 32, generator=g).cuda()


def _t(v, B):
    return torch.tensor(float(v), device="cuda").expand(B)


def test_fullsize_gemm_sampled_rows_vs_oracle():
    """fc1-shaped GEMM at M = 64*257 (extra-strip path, 4 full rounds of 256x256 tiles): 96 sampled rows,
    including the strip rows and tile corners, against the CPU oracle."""
    from uspace_amd import _hip
    g = torch.Generator().manual_seed(3)
    M, N, K = 64 * 257, 4096, 1024
    A = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    _hip.gemm(A.cuda(), W.cuda(), bias=b.cuda(), out_f32=out)
    rows = np.unique(np.concatenate([np.arange(0, 8), np.arange(250, 262), np.arange(16376, 16448),
                                     np.random.default_rng(0).integers(0, M, 16)]))
    ref = C.linear(A.float().numpy()[rows], W.float().numpy(), b.numpy())
    got = out[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert rel_l2(got, ref) < 1e-5
    assert np.isfinite(out.sum().item())


def test_fullsize_attention_sampled_heads_vs_oracle():
    from uspace_amd import _hip
    g = torch.Generator().manual_seed(4)
    for L in (257, 334):
        B, H = 64, 16
        qkv = (torch.randn(B * L, 3 * H * 64, generator=g) * 1.2).to(torch.bfloat16)
        out = _hip.attention(qkv.cuda(), B, L, H).float().cpu().numpy().reshape(B, L, H, 64)
        q3 = qkv.float().numpy().reshape(B, L, 3, H, 64)
        for (bb, hh) in ((0, 0), (63, 15), (17, 9)):
            one = np.ascontiguousarray(q3[bb:bb + 1, :, :, hh:hh + 1, :]).reshape(1, L, 192)
            ref = C.attention(one, 1)
            assert rel_l2(out[bb, :, hh, :], ref[0]) < 6e-3


def test_config2_batch_slices_are_independent_and_deterministic(net_L_u):
    """U-ViT-L, batch 64: every row of the batch-64 result equals the same sample evaluated in a batch of 8
    (no cross-sample coupling; identical K-order accumulation), and two runs are bit-identical."""
    z = _z(64)
    a, _ = net_L_u(z, _t(0.35, 64), None, edit_loc=None)
    b, _ = net_L_u(z, _t(0.35, 64), None, edit_loc=None)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    sub, _ = net_L_u(z[24:32].contiguous(), _t(0.35, 8), None, edit_loc=None)
    r = rel_l2(sub.cpu().numpy(), a[24:32].cpu().numpy())
    # a SMOKE line only: different tile shapes change the fp32 summation grouping of the folded LayerNorm statistics (and fc2's K
    # parts); through 21 blocks of bf16 operands that moves the result by about the distance to the fp32 reference itself (5e-3,
    # tests/test_gpu_forward.py::test_big_shapes_match_reference_golden), so this bound cannot see a small tile-plan bug -- the
    # launch-by-launch comparison below (test_every_launch_of_a_block_is_the_same_at_every_batch_size) is the check that can
    assert r < 6e-3, r
    assert float(a.std()) > 1e-3


@pytest.mark.parametrize("sub", [(24, 32), (0, 32), (16, 64)])
def test_every_launch_of_a_block_is_the_same_at_every_batch_size(sub):
    """The end-to-end slice check above cannot see an error below one forward's bf16 noise.  This one compares
    launch by launch: each GEMM of a U-ViT-L block on 64 x 257 rows (256x256 tiles + strips) against the same rows [lo, hi) x 257
    launched on their own -- 8, 32 and 48 samples: 64x64 / 128x128, 256x128, 192x256 tiles, other row plans, the in-launch K-split
    tail for fc2.  Every tile form walks K in the same order with the same MFMA, so outputs are BIT-EQUAL whenever no form splits K
    (qkv, fc1 + GELU, skip_linear over two K slabs, proj with its residual: the residual is the accumulator's initial value in every
    form); fc2 at the smaller row counts runs in K parts (p0 + p1 + ...): equal to fp32 summation order."""
    import ctypes
    from uspace_amd import _hip
    lo, hi = sub[0] * 257, sub[1] * 257
    M, D = 64 * 257, 1024
    g = torch.Generator().manual_seed(21)
    x_bf = (torch.randn(M, D, generator=g) * 1.0).to(torch.bfloat16).cuda()
    f_bf = (torch.randn(M, 4 * D, generator=g) * 0.7).to(torch.bfloat16).cuda()
    s_bf = (torch.randn(M, D, generator=g) * 1.0).to(torch.bfloat16).cuda()
    resid = torch.randn(M, D, generator=g).cuda()
    W = {k: (torch.randn(n, kk, generator=g) * 0.03).to(torch.bfloat16).cuda()
         for k, (n, kk) in dict(qkv=(3 * D, D), proj=(D, D), fc1=(4 * D, D), fc2=(D, 4 * D), skip=(D, 2 * D)).items()}
    bias = {k: torch.randn(w.shape[0], generator=g).cuda() for k, w in W.items()}
    plan = (ctypes.c_int * 8)()

    def form(m, n, k):
        _hip.check(_hip.lib().uspace_gemm_plan_k(m, n, k, 0, plan), "plan")
        return plan[0]

    def run(name, rows):
        a_x, a_f, a_s, r = x_bf[rows], f_bf[rows], s_bf[rows], resid[rows].clone()
        m = a_x.shape[0]
        sk = None
        need = _hip.lib().uspace_gemm_sk_ws_bytes(m, D, 4 * D)
        if name == "fc2" and need:
            sk = torch.empty(need, dtype=torch.uint8, device="cuda")
        if name == "qkv":
            o = torch.empty(m, 3 * D, dtype=torch.bfloat16, device="cuda")
            _hip.gemm(a_x, W["qkv"], out_bf16=o)
            return o
        if name == "fc1":
            o = torch.empty(m, 4 * D, dtype=torch.bfloat16, device="cuda")
            _hip.gemm(a_x, W["fc1"], bias=bias["fc1"], gelu=True, out_bf16=o)
            return o
        if name == "skip":
            o = torch.empty(m, D, device="cuda")
            ob = torch.empty(m, D, dtype=torch.bfloat16, device="cuda")
            _hip.gemm(a_x, W["skip"], A2=a_s, bias=bias["skip"], out_f32=o, out_bf16=ob)
            return o
        if name == "proj":
            _hip.gemm(a_x, W["proj"], bias=bias["proj"], resid=r, out_f32=r)
            return r
        ob = torch.empty(m, D, dtype=torch.bfloat16, device="cuda")
        _hip.gemm(a_f, W["fc2"], bias=bias["fc2"], resid=r, out_f32=r, out_bf16=ob, sk_ws=sk)
        return r

    all_rows, part = slice(0, M), slice(lo, hi)
    forms = {}
    for name, (n, k) in dict(qkv=(3 * D, D), fc1=(4 * D, D), skip=(D, 2 * D), proj=(D, D), fc2=(D, 4 * D)).items():
        big = run(name, all_rows)[lo:hi]
        small = run(name, part)
        forms[name] = (form(M, n, k), form(hi - lo, n, k))
        split_k = name == "fc2" and 6 in forms[name]
        if split_k:
            d = (big.float() - small.float()).abs().max().item()
            assert d <= 2e-5 * big.float().abs().max().item(), (name, forms[name], d)
        else:
            assert torch.equal(big, small), (name, forms[name], (big.float() - small.float()).abs().max().item())
    # the point of the comparison: the sub-batch really ran on other tile forms / row plans than the batch of 64
    assert any(a != b for a, b in forms.values()) or sub == (16, 64), forms


@pytest.mark.parametrize("B", [32, 16])
def test_k_split_tail_inside_the_model_is_reproducible_and_matches_the_switch_off(net_L_u, B):
    """U-ViT-L at 32 (config 5's share) and 16 per GPU: every fc2 of the forward shares its tiles between 2 / 4 workgroups that
    exchange partial sums inside the launch (write-through stores + one arrival counter per tile, counters of all launches zeroed by one
    memset per forward).  (a) 24 forwards in a row, other inputs in between (the exchange workspace is reused by all 21 fc2 launches of
    every forward, under the load of the forward itself): bit-identical -- a stale slab line or a counter left over would show;
    (b) hipGraph replay (the memset and the per-launch counter slices are captured) == eager; (c) with the switch off
    (uspace_gemm_set_sk(0): the round-5 tile forms) the same numbers to the bf16 noise floor of a forward."""
    from uspace_amd import _hip
    L = _hip.lib()
    import ctypes
    plan = (ctypes.c_int * 8)()
    _hip.check(L.uspace_gemm_plan_k(B * 257, 1024, 4096, 1, plan), "plan")
    assert plan[0] == 6 and plan[1] == (2 if B == 32 else 4), list(plan)
    z, z2 = _z(B), _z(B, seed=8)
    first, _ = net_L_u(z, _t(0.35, B), None, edit_loc=None)
    first = first.clone()
    for rep in range(24):
        net_L_u(z2, _t(0.1 + 0.03 * rep, B), None, edit_loc=None)
        again, _ = net_L_u(z, _t(0.35, B), None, edit_loc=None)
        assert torch.equal(again, first), rep
    was = net_L_u.use_graph
    try:
        net_L_u.use_graph = True
        for _ in range(3):
            g, _aux = net_L_u(z, _t(0.35, B), None, edit_loc=None)
            assert torch.equal(g, first)
    finally:
        net_L_u.use_graph = was
    try:
        _hip.check(L.uspace_gemm_set_sk(0), "set_sk")
        net_L_u._workspace.clear()                        # sized for the other plan
        _hip.check(L.uspace_gemm_plan_k(B * 257, 1024, 4096, 1, plan), "plan")
        assert plan[0] != 6
        off, _ = net_L_u(z, _t(0.35, B), None, edit_loc=None)
    finally:
        L.uspace_gemm_set_sk(-1)
        net_L_u._workspace.clear()
    assert L.uspace_gemm_get_sk() == 1
    r = rel_l2(off.cpu().numpy(), first.cpu().numpy())
    assert 0 < r < 6e-3, r


def test_config2_solver_roundtrip_and_shard_equivalence(net_L_u):
    """Euler/RK4 fixed steps at batch 64: encode then decode returns the input; solving two half batches
    (what two ranks would do) reproduces the full-batch solve."""
    from uspace_amd.flow_matching import CNF
    cnf = CNF(net_L_u)
    sk = dict(solver="fixed", solver_fix="rk4", solver_fix_step=0.125, solver_adaptive="dopri5", solver_adaptive_prec=0.01)
    kw = dict(dissect_name="none", edit_loc=None, solver_kwargs=sk)
    z = _z(64)
    x1 = cnf.decode(z, None, **kw)
    assert cnf.last_stats.nfe == 32
    back = cnf.encode(x1, None, **kw)
    assert rel_l2(back.cpu().numpy(), z.cpu().numpy()) < 5e-3
    halves = torch.cat([cnf.decode(z[:32].contiguous(), None, **kw), cnf.decode(z[32:].contiguous(), None, **kw)])
    assert rel_l2(halves.cpu().numpy(), x1.cpu().numpy()) < 2e-3


def test_config5_mid_hook_algebra_at_full_width(net_L_u):
    """U-ViT-L u-space edit at the mid block, batch 32 (config 5's per-GPU share), synthetic direction table
    [40,257,1024] ~ N(0, 0.01^2): scale 0 == no hook bit-exactly, attribute string averages rows, skipped
    when t > t_edit, and the tail hook is exactly additive."""
    rng = np.random.default_rng(11)
    table = (rng.standard_normal((40, 257, 1024)) * 0.01).astype(np.float32)
    z = _z(32)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "delta_0.20.npy"), table)
        np.save(os.path.join(d, "delta_0.60.npy"), table)
        base = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=d, edit_loc="mid")
        plain, _ = net_L_u(z, _t(0.2, 32), None, edit_loc=None)
        zero, _ = net_L_u(z, _t(0.2, 32), None, ith_attr="31_39_20", write_scale=0.0, **base)
        assert torch.equal(plain, zero)
        e1, _ = net_L_u(z, _t(0.2, 32), None, ith_attr="31_39_20", write_scale=1.0, **base)
        assert rel_l2(e1.cpu().numpy(), plain.cpu().numpy()) > 1e-5
        # "31_39_20" == mean of the three rows: same result from a one-row table holding that mean
        ad = os.path.join(d, "avg")
        os.makedirs(ad)
        mean_row = (table[31] + table[39] + table[20]) / np.float32(3)
        np.save(os.path.join(ad, "delta_0.20.npy"), mean_row[None])
        e2, _ = net_L_u(z, _t(0.2, 32), None, ith_attr=0, write_scale=1.0, **dict(base, write_path_root=ad))
        assert torch.equal(e1, e2)
        late, _ = net_L_u(z, _t(0.6, 32), None, ith_attr=3, write_scale=5.0, **base)     # 0.60 > t_edit: untouched
        plain6, _ = net_L_u(z, _t(0.6, 32), None, edit_loc=None)
        assert torch.equal(late, plain6)
        # tail hook: out + s*delta exactly
        img = (rng.standard_normal((5, 4, 32, 32)) * 0.3).astype(np.float32)
        td = os.path.join(d, "tail")
        os.makedirs(td)
        np.save(os.path.join(td, "delta_0.20.npy"), img)
        tail, _ = net_L_u(z, _t(0.2, 32), None, dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4,
                          write_path_root=td, edit_loc="tail", ith_attr=2, write_scale=-1.5)
        want = plain + torch.from_numpy(img[2]).cuda()[None] * (-1.5)
        torch.testing.assert_close(tail, want, rtol=1e-6, atol=1e-6)


def test_config3_and_4_t2i_shapes_run_and_slice_consistently():
    """U-ViT-L T2I batch 64 (config 3) and U-ViT-S-deep16 T2I batch 64 = one GPU's share of config 4."""
    from uspace_amd.tools.utils_uvit import get_nnet
    g = torch.Generator().manual_seed(7)
    for cfg in (L_CFG, S_CFG):
        torch.manual_seed(1234)
        net = get_nnet("uvit_t2i", clip_dim=768, num_clip_token=77, **COMMON, **cfg).cuda().eval()
        z = torch.randn(64, 4, 32, 32, generator=g).cuda()
        ctx = torch.randn(64, 77, 768, generator=g).cuda()
        a, _ = net(z, _t(0.5, 64), context=ctx)
        assert bool(torch.isfinite(a).all()) and float(a.std()) > 1e-3
        sub, _ = net(z[8:16].contiguous(), _t(0.5, 8), context=ctx[8:16].contiguous())
        assert rel_l2(sub.cpu().numpy(), a[8:16].cpu().numpy()) < 6e-3      # see the config-2 test above
        # attention-map edit on every block, all rows: differs from the plain result, finite
        ids = [np.array([2, 5, 9])] * 64
        e, _ = net(z, _t(0.3, 64), context=ctx, dissect_name="p2p", fm_direction="decode", t_edit=0.5, block_id="all",
                   target_context_ids=ids, token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=6.0))
        p, _ = net(z, _t(0.3, 64), context=ctx)
        assert bool(torch.isfinite(e).all()) and not torch.equal(e, p)
        del net
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------
# Full-size forwards against the CPU oracle on sampled batch rows (rows are independent; the oracle does a pair of L-size
# samples in under a second on the GPU host).  These are the only end-to-end checks of the 256 x 256 + strip tile plans
# (the reference goldens are batch 2) -- VERDICT r1 "weak / parity" item 2.
# ------------------------------------------------------------------------------------------------------------------
ROWS = (0, 31, 63)
FWD_TOL = 1e-2            # the tolerance contract of DESIGN.md: one forward rel-L2 <= 1e-2 against the fp32 oracle


def _oracle_rows(net, spec_kw, x, tval, rows, context=None, **kw):
    from oracle import uvit_oracle as O
    spec = O.UViTSpec(img_size=32, patch_size=2, in_chans=4, **spec_kw)
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    idx = torch.tensor(rows)
    ctx = context[idx].cpu().numpy() if context is not None else None
    return O.uvit_forward(spec, sd, x[idx].cpu().numpy(), np.full(len(rows), tval, np.float32), context=ctx, **kw)


def test_config2_rows_match_oracle(net_L_u):
    z = _z(64)
    out, _ = net_L_u(z, _t(0.35, 64), None, edit_loc=None)
    ref = _oracle_rows(net_L_u, L_CFG, z, 0.35, ROWS, edit_loc=None)
    got = out[torch.tensor(ROWS)].cpu().numpy()
    for k, r in enumerate(ROWS):
        e = rel_l2(got[k], ref[k])
        assert e <= FWD_TOL, (r, e)
    assert rel_l2(got, ref) <= FWD_TOL


@pytest.mark.parametrize("cfg", [L_CFG, S_CFG], ids=["L_t", "S_t"])
def test_config3_and_4_rows_match_oracle(cfg):
    from uspace_amd.tools.utils_uvit import get_nnet
    torch.manual_seed(1234)
    net = get_nnet("uvit_t2i", clip_dim=768, num_clip_token=77, **COMMON, **cfg).cuda().eval()
    g = torch.Generator().manual_seed(7)
    z = torch.randn(64, 4, 32, 32, generator=g).cuda()
    ctx = torch.randn(64, 77, 768, generator=g).cuda()
    out, _ = net(z, _t(0.5, 64), context=ctx)
    ref = _oracle_rows(net, dict(cfg, t2i=True), z, 0.5, ROWS, context=ctx)
    assert rel_l2(out[torch.tensor(ROWS)].cpu().numpy(), ref) <= FWD_TOL
    # one prompt-to-prompt edit at L = 334: a different multiplier per row, every block, t below t_edit
    ids = [np.array([2 + (b % 5), 9, 40]) for b in range(64)]
    kw = dict(dissect_name="p2p", fm_direction="decode", t_edit=0.5, block_id="all",
              token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=8.0))
    e, _ = net(z, _t(0.3, 64), context=ctx, target_context_ids=ids, **kw)
    ref_e = _oracle_rows(net, dict(cfg, t2i=True), z, 0.3, ROWS, context=ctx, target_context_ids=[ids[r] for r in ROWS], **kw)
    ref_p = _oracle_rows(net, dict(cfg, t2i=True), z, 0.3, ROWS, context=ctx)
    got_e = e[torch.tensor(ROWS)].cpu().numpy()
    assert rel_l2(got_e, ref_e) <= FWD_TOL
    # the edit itself is reproduced: its effect is several times the bf16 noise floor of a forward (about 5e-3)
    assert rel_l2(ref_e, ref_p) > 1e-2 and rel_l2(got_e - ref_p, ref_e - ref_p) < 0.35
    del net
    torch.cuda.empty_cache()


def test_config5_mid_hook_rows_match_oracle(net_L_u):
    """BASELINE config 5's per-GPU share: U-ViT-L, batch 32, u-space write hook at the mid block."""
    rng = np.random.default_rng(11)
    table = (rng.standard_normal((40, 257, 1024)) * 1.5).astype(np.float32)
    z = _z(32)
    rows = (0, 17, 31)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "delta_0.20.npy"), table)
        kw = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=d, edit_loc="mid",
                  ith_attr="31_39_20", write_scale=2.0)
        out, _ = net_L_u(z, _t(0.2, 32), None, **kw)
        ref = _oracle_rows(net_L_u, L_CFG, z, 0.2, rows, **kw)
        plain = _oracle_rows(net_L_u, L_CFG, z, 0.2, rows, edit_loc=None)
    got = out[torch.tensor(rows)].cpu().numpy()
    assert rel_l2(got, ref) <= FWD_TOL
    assert rel_l2(ref, plain) > 1e-2 and rel_l2(got - plain, ref - plain) < 0.35


STRESS_CFGS = [pytest.param(S_CFG, id="S_u"), pytest.param(L_CFG, id="L_u-headline-width")]


@pytest.mark.parametrize("cfg", STRESS_CFGS)
def test_layernorm_fold_survives_large_row_means(cfg):
    """Whole-network stress of the folded LayerNorm: pos_embed x50 and constant proj / fc2 / skip biases
    make every token's mean dwarf its standard deviation and shift it at every sublayer -- the situation of outlier channels
    in trained U-ViTs, where a variance taken as E[(x-c)^2] - d^2 would lose bits if c tracked the mean badly.  Folded,
    separate-launch and oracle results must agree within the forward tolerance.  At both model widths: U-ViT-L at batch 64 is the
    headline launch geometry (256x256 tiles + remainder strips, rank-1 skips at D = 1024; libs/uvit.py:293-300 init is all anyone has)."""
    from uspace_amd import _hip
    from uspace_amd.tools.utils_uvit import get_nnet
    torch.manual_seed(1234)
    net = get_nnet("uvit", num_classes=-1, **COMMON, **cfg).cuda().eval()
    with torch.no_grad():
        net.pos_embed.mul_(50.0)
        net.pos_embed.add_(3.0)
        for i, blk in enumerate(net._blocks()):
            blk.attn.proj.bias.fill_(0.75 * (1 + i % 3))
            blk.mlp.fc2.bias.fill_(-0.5 * (1 + i % 2))
            if hasattr(blk, "skip_linear"):
                blk.skip_linear.bias.fill_(1.25)
            blk.norm1.bias.normal_(0.0, 0.3)
            blk.norm2.weight.normal_(1.0, 0.2)
    z = _z(64)
    L = _hip.lib()
    outs = {}
    try:
        for fold in (1, 0):
            _hip.check(L.uspace_uvit_set_ln_fold(fold), "set_ln_fold")
            outs[fold], _ = net(z, _t(0.4, 64), None, edit_loc=None)
    finally:
        L.uspace_uvit_set_ln_fold(-1)
    ref = _oracle_rows(net, cfg, z, 0.4, ROWS, edit_loc=None)
    idx = torch.tensor(ROWS)
    e_fold = rel_l2(outs[1][idx].cpu().numpy(), ref)
    e_sep = rel_l2(outs[0][idx].cpu().numpy(), ref)
    assert e_sep <= FWD_TOL and e_fold <= FWD_TOL, (e_fold, e_sep)
    assert e_fold <= 1.5 * e_sep + 1e-3, (e_fold, e_sep)          # folding must not be the less accurate path
    assert rel_l2(outs[1].cpu().numpy(), outs[0].cpu().numpy()) <= FWD_TOL


@pytest.mark.parametrize("cfg", STRESS_CFGS)
def test_outlier_channels_and_tokens_stay_within_the_forward_tolerance(cfg):
    """No trained checkpoint is available, so the statistics trained ViTs are known for are built in by hand:
    a few MASSIVE channels in the residual stream (pos_embed +-60 in four channels, kept alive by the matching proj / fc2 output
    rows at 8x and their biases), a few outlier TOKENS (pos_embed rows at 12x: their keys draw sharply peaked softmax rows, which
    is where the bf16 rounding of P sits), and LayerNorm gains that amplify the massive channels.  The bf16 centred copies, the
    centred long skips with their rank-1 term and the bf16 P of the attention kernel all see it; both LayerNorm modes against the
    fp32 oracle, and against each other.  Both widths (VERDICT r4 task 4: the L shapes saw random init only)."""
    from uspace_amd import _hip
    from uspace_amd.tools.utils_uvit import get_nnet
    torch.manual_seed(1234)
    net = get_nnet("uvit", num_classes=-1, **COMMON, **cfg).cuda().eval()
    D = cfg["embed_dim"]
    big = [7, 130, 301, D - 57]
    with torch.no_grad():
        for j, ch in enumerate(big):
            net.pos_embed[:, :, ch] += 60.0 if j % 2 == 0 else -60.0
        net.pos_embed[:, [3, 77, 200], :] *= 12.0
        for i, blk in enumerate(net._blocks()):
            blk.attn.proj.weight[big] *= 8.0
            blk.mlp.fc2.weight[big] *= 8.0
            blk.attn.proj.bias[big] = 2.0 * (1 + i % 2)
            blk.norm1.weight[big] = 3.0
            blk.norm2.weight[big] = 0.2
            if hasattr(blk, "skip_linear"):
                blk.skip_linear.weight[big] *= 4.0
    z = _z(64, seed=5)
    L = _hip.lib()
    outs = {}
    try:
        for fold in (1, 0):
            _hip.check(L.uspace_uvit_set_ln_fold(fold), "set_ln_fold")
            outs[fold], _ = net(z, _t(0.6, 64), None, edit_loc=None)
    finally:
        L.uspace_uvit_set_ln_fold(-1)
    ref = _oracle_rows(net, cfg, z, 0.6, ROWS, edit_loc=None)
    idx = torch.tensor(ROWS)
    e_fold, e_sep = rel_l2(outs[1][idx].cpu().numpy(), ref), rel_l2(outs[0][idx].cpu().numpy(), ref)
    assert np.isfinite(outs[1].sum().item()) and np.abs(ref).max() > 0.1
    assert e_sep <= FWD_TOL and e_fold <= FWD_TOL, (e_fold, e_sep)
    assert e_fold <= 1.5 * e_sep + 1e-3, (e_fold, e_sep)
    assert rel_l2(outs[1].cpu().numpy(), outs[0].cpu().numpy()) <= FWD_TOL


@pytest.mark.parametrize("solver,key,nfe", [("euler", "x1_euler50", 50), ("dopri5", "x1_dopri5_50", 301)])
def test_headline_solve_matches_reference_trajectory(net_L_u, golden_dir, solver, key, nfe):
    """BASELINE configs[1] end to end: the reference U-ViT-L driven through 50 Euler steps and through 50 fixed Dormand-Prince
    steps (301 evaluations) at B=2 in the build container (tests/golden/make_golden.py::make_traj_L_u, solver selection as
    /root/reference/flow_matching.py:130-151).  Here the two latents ride in a batch of 64 -- the exact headline launch shapes
    (M = 64*257 rows) -- and their end states must match the reference's (trajectories are independent across the batch)."""
    from uspace_amd.flow_matching import CNF
    zf = np.load(os.path.join(golden_dir, "traj_L_u.npz"))
    z = _z(64, seed=21)
    z[5] = torch.from_numpy(zf["z"][0]).cuda()
    z[63] = torch.from_numpy(zf["z"][1]).cuda()
    cnf = CNF(net_L_u)
    sk = dict(solver="adaptive" if solver == "dopri5" else "fixed", solver_fix="euler", solver_fix_step=0.02,
              solver_adaptive="dopri5", solver_adaptive_prec=0.01, n_steps=50)
    x1 = cnf.decode(z, None, dissect_name="none", edit_loc=None, solver_kwargs=sk)
    assert cnf.last_stats.nfe == nfe == (int(zf["nfe_dopri5"]) if solver == "dopri5" else 50)
    got = x1[[5, 63]].cpu().numpy()
    r = rel_l2(got, zf[key])
    assert r < 5e-3, r                       # SURVEY.md section 7 trajectory gate (bf16 operands, fp32 state)
    # the two integrators agree with each other to their truncation error, far above this tolerance's noise floor
    assert np.isfinite(x1.sum().item())
