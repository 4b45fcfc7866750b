"""Per-kernel parity: every HIP operator, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  bf16 operands are rounded once on the host so both sides see identical inputs;
what remains is fp32 accumulation order (and bf16 rounding of bf16 outputs)."""
import numpy as np
import pytest
import torch

from oracle import _cops as C
from tests.util import bf16_round, rel_l2, to_dev

pytestmark = pytest.mark.gpu

BF16_EPS = 2.0 ** -8   # half-ulp relative rounding of a bf16 store


@pytest.fixture(scope="module")
def hip():
    from uspace_amd import _hip
    _hip.lib()
    return _hip


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [
    (300, 64, 64),        # tiny-model shapes, ragged M, N smaller than the tile
    (771, 192, 128),
    (1030, 1024, 256),    # 128x128 tiles, ragged last M tile
    (4100, 4096, 128),    # 256x256 tiles (>= 256 of them), ragged last M tile
    (16, 256, 64),
])
def test_gemm_plain_and_transpose_detecting(hip, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)      # asymmetric random operands catch row/col swaps
    ref = C.linear(A, W)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    hip.gemm(to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), out_f32=out)
    got = out.cpu().numpy()
    assert rel_l2(got, ref) < 1e-5
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())


@pytest.mark.parametrize("seed", range(12))
def test_gemm_random_shapes_and_epilogues(hip, seed):
    """Seeded random [M, N, K] (ragged rows, N any multiple of 4 -- not of the 16-column MFMA tile --, 1 to 8 K tiles) with a
    random supported epilogue (libs/timm.py:106-112, libs/uvit.py:159-161), against the oracle's linear / GELU."""
    rng = np.random.default_rng(1000 + seed)
    M = int(rng.integers(1, 5000))
    N = 4 * int(rng.integers(1, 320))
    K = 64 * int(rng.integers(1, 9))
    kind = ["f32", "bias_bf16", "bias_gelu_bf16", "bias_resid_f32", "bias_resid_f32_bf16", "bias_f32", "bf16"][seed % 7]
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    R = _rand(rng, M, N)
    ref = C.linear(A, W, b if "bias" in kind else None)
    if "gelu" in kind:
        ref = C.gelu(ref)
    if "resid" in kind:
        ref = ref + R
    dA, dW = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16)
    x = to_dev(R).clone() if "resid" in kind else torch.full((M, N), float("nan"), device="cuda")
    xb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.gemm(dA, dW, bias=to_dev(b) if "bias" in kind else None, resid=x if "resid" in kind else None, gelu="gelu" in kind,
             out_f32=x if "f32" in kind else None, out_bf16=xb if "bf16" in kind else None)
    if "f32" in kind:
        got = x.cpu().numpy()
        assert rel_l2(got, ref) < 1e-5, (M, N, K, kind, rel_l2(got, ref))
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    if "bf16" in kind:
        gotb = xb.float().cpu().numpy()
        assert np.isfinite(gotb).all()
        assert (np.abs(gotb - ref) <= BF16_EPS * np.abs(ref) * 1.01 + 1e-3 * np.abs(ref).max()).all(), (M, N, K, kind)


def test_gemm_identity_weight_reproduces_rows(hip):
    # W = I (asymmetric A): output must be A itself, bit-exact in fp32
    rng = np.random.default_rng(0)
    A = bf16_round(_rand(rng, 130, 64))
    out = torch.empty(130, 64, dtype=torch.float32, device="cuda")
    hip.gemm(to_dev(A, torch.bfloat16), to_dev(np.eye(64, dtype=np.float32), torch.bfloat16), out_f32=out)
    np.testing.assert_array_equal(out.cpu().numpy(), A)


@pytest.mark.parametrize("big", [False, True])
def test_gemm_epilogues(hip, big):
    rng = np.random.default_rng(5 + big)
    M, N, K = (4099, 4096, 64) if big else (515, 256, 192)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    R = _rand(rng, M, N)
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    lin = C.linear(A, W, b)
    # bias + GELU -> bf16   (fc1)
    o16 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(dA, dW, bias=db, gelu=True, out_bf16=o16)
    ref = C.gelu(lin)
    err = np.abs(o16.float().cpu().numpy() - ref)
    assert (err <= BF16_EPS * np.abs(ref) * 1.01 + 1e-4).all()
    # bias + residual, in place, + bf16 copy   (proj / fc2)
    x = to_dev(R).clone()
    xb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(dA, dW, bias=db, resid=x, out_f32=x, out_bf16=xb)
    ref = lin + R
    got = x.cpu().numpy()
    assert rel_l2(got, ref) < 1e-5
    np.testing.assert_array_equal(xb.float().cpu().numpy(), bf16_round(got))
    # no bias, bf16 only   (qkv)
    hip.gemm(dA, dW, out_bf16=o16)
    ref = C.linear(A, W)
    err = np.abs(o16.float().cpu().numpy() - ref)
    assert (err <= BF16_EPS * np.abs(ref) * 1.01 + 1e-4).all()


def test_gemm_two_k_slabs_equal_concat(hip):
    # skip_linear(cat([x, skip], -1)) == x @ W[:, :D]^T + skip @ W[:, D:]^T   (libs/uvit.py:159)
    rng = np.random.default_rng(9)
    M, D = 517, 128
    X = bf16_round(_rand(rng, M, D))
    S = bf16_round(_rand(rng, M, D))
    W = bf16_round(_rand(rng, D, 2 * D) * 0.1)
    b = _rand(rng, D)
    ref = C.linear(np.concatenate([X, S], axis=1), W, b)
    out = torch.empty(M, D, dtype=torch.float32, device="cuda")
    hip.gemm(to_dev(X, torch.bfloat16), to_dev(W, torch.bfloat16), A2=to_dev(S, torch.bfloat16), bias=to_dev(b),
             out_f32=out)
    assert rel_l2(out.cpu().numpy(), ref) < 1e-5


def test_gemm_rejects_bad_shapes(hip):
    A = torch.zeros(8, 96, dtype=torch.bfloat16, device="cuda")     # K not a multiple of 64
    W = torch.zeros(64, 96, dtype=torch.bfloat16, device="cuda")
    out = torch.empty(8, 64, dtype=torch.float32, device="cuda")
    with pytest.raises(hip.UspaceHipError):
        hip.gemm(A, W, out_f32=out)


@pytest.mark.parametrize("M,D", [(5, 64), (1031, 512), (777, 1024), (9, 2048)])
def test_layernorm(hip, M, D):
    rng = np.random.default_rng(D + M)
    x = _rand(rng, M, D, scale=3.0) + 0.7
    g = _rand(rng, D) + 1.0
    b = _rand(rng, D)
    ref = C.layernorm(x, g, b)
    got = hip.layernorm(to_dev(x), to_dev(g), to_dev(b)).float().cpu().numpy()
    err = np.abs(got - ref)
    assert (err <= BF16_EPS * np.abs(ref) * 1.01 + 2e-5).all(), err.max()


# (B * H <= 64 with at most 17 key tiles runs four workgroups per head, each on a quarter of the query tiles, B * H <= 128 two;
# (9, 257, 16) is the one-workgroup-per-head form of the large batches)
@pytest.mark.parametrize("B,L,H", [(2, 65, 1), (2, 142, 2), (3, 257, 2), (2, 258, 1), (2, 334, 3), (1, 1, 1), (1, 17, 1), (9, 257, 16), (5, 257, 16), (4, 257, 8)])
@pytest.mark.parametrize("scaled", [False, True])
def test_attention(hip, B, L, H, scaled):
    rng = np.random.default_rng(B * 1000 + L + H)
    qkv = bf16_round(_rand(rng, B, L, 3 * H * 64, scale=1.5))
    ks = None
    if scaled:
        ks = np.ones((B, L), np.float32)
        ks[:, rng.integers(0, L, size=max(1, L // 9))] = 3.0
        ks[0, 0] = 0.0
    ref = C.attention(qkv, H, ks)
    got = hip.attention(to_dev(qkv.reshape(B * L, -1), torch.bfloat16), B, L, H,
                        key_scale=to_dev(ks) if scaled else None)
    got = got.float().cpu().numpy().reshape(B, L, H * 64)
    # P is rounded to bf16 before P.V (and the output is stored in bf16): ~1e-2 relative at worst
    assert rel_l2(got, ref) < 6e-3, rel_l2(got, ref)
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


@pytest.mark.parametrize("B,L,H", [(64, 257, 16), (33, 257, 16), (41, 257, 16), (64, 334, 16), (40, 334, 8), (65, 334, 4), (23, 300, 16)])
def test_attention_several_heads_per_workgroup_is_bit_equal_to_one(hip, B, L, H):
    """More heads than resident workgroups (two per CU at L <= 272, one above): a workgroup walks 2 - 4 heads and fetches the next
    head's K / V through registers while it computes (attention.hip, HPW).  Same arithmetic: every head's output is bit-equal to the
    one-head-per-workgroup launch that small batches take, checked by running the batch in pieces of at most 256 heads; against the
    oracle on a sample of heads."""
    rng = np.random.default_rng(B * 7 + L + H)
    qkv = torch.from_numpy(bf16_round(_rand(rng, B * L, 3 * H * 64, scale=1.5))).to("cuda", dtype=torch.bfloat16)
    whole = hip.attention(qkv, B, L, H)
    nb = max(1, 256 // H)
    parts = [hip.attention(qkv[b0 * L:min(B, b0 + nb) * L].contiguous(), min(B, b0 + nb) - b0, L, H) for b0 in range(0, B, nb)]
    assert torch.equal(whole, torch.cat(parts))
    for b in (0, B // 2, B - 1):
        ref = C.attention(qkv[b * L:(b + 1) * L].float().cpu().numpy().reshape(1, L, -1), H)
        assert rel_l2(whole[b * L:(b + 1) * L].float().cpu().numpy().reshape(1, L, H * 64), ref) < 6e-3


@pytest.mark.parametrize("B,L,H", [(64, 334, 16), (40, 257, 16)])
def test_attention_ignores_what_lies_behind_the_tensor(hip, B, L, H):
    """The padded key / value rows (L .. 16 * tiles) of the LAST batch element lie behind the caller's tensor.  Their P is 0, but 0 x NaN
    is NaN in the P.V product: whatever the kernel reads there must be finite by construction (LDS-DMA path: row L - 1 repeated;
    several-heads-per-workgroup path: out of the buffer descriptor's range, zeros), not by luck.  qkv is a view of a larger buffer whose
    tail holds NaN / Inf bf16 patterns; the result must equal the run on a tensor with an ordinary tail, bit for bit."""
    rng = np.random.default_rng(B + L + H)
    n = B * L * 3 * H * 64
    vals = torch.from_numpy(bf16_round(_rand(rng, n, scale=1.5))).to(torch.bfloat16)
    tail = 64 * 3 * H * 64                                       # far more rows than any tile padding reaches
    outs = []
    for poison in (False, True):
        big = torch.zeros(n + tail, dtype=torch.bfloat16, device="cuda")
        big[:n] = vals.cuda()
        if poison:
            pat = torch.from_numpy(np.array([0x7FC0, 0x7F80, 0xFF80, 0x7FFF], dtype=np.uint16).view(np.int16))   # NaN, +Inf, -Inf, NaN
            big[n:] = pat.repeat(tail // 4).view(torch.bfloat16).cuda()
        outs.append(hip.attention(big[:n].view(B * L, 3 * H * 64), B, L, H))
    assert bool(torch.isfinite(outs[1].float()).all())
    assert torch.equal(outs[0], outs[1])


def test_attention_spiked_row_softmax_is_stable(hip):
    # one key dominates one query by a large margin: exp underflow elsewhere must not produce NaN
    rng = np.random.default_rng(3)
    B, L, H = 1, 257, 1
    qkv = bf16_round(_rand(rng, B, L, 192))
    qkv[0, 5, 0:64] = 30.0
    qkv[0, 200, 64:128] = 30.0
    ref = C.attention(qkv, H)
    got = hip.attention(to_dev(qkv.reshape(L, -1), torch.bfloat16), B, L, H).float().cpu().numpy().reshape(B, L, 64)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 6e-3


def test_attention_rejects_long_sequences(hip):
    qkv = torch.zeros(400, 192, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(hip.UspaceHipError):
        hip.attention(qkv, 1, 400, 1)


def test_add_broadcast_and_cast(hip):
    rng = np.random.default_rng(4)
    x = _rand(rng, 3, 4, 16, 16)
    d = _rand(rng, 4, 16, 16)
    dx = to_dev(x).clone()
    xb = torch.empty(x.shape, dtype=torch.bfloat16, device="cuda")
    hip.add_broadcast(dx, to_dev(d).reshape(-1), -1.5, x_bf16=xb)
    ref = x + d[None] * np.float32(-1.5)
    np.testing.assert_allclose(dx.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(xb.float().cpu().numpy(), bf16_round(dx.cpu().numpy()))
    odd = _rand(rng, 3, 7)       # per-sample size not a multiple of 4 -> scalar tail kernel
    dodd = to_dev(odd).clone()
    hip.add_broadcast(dodd, to_dev(odd[0].copy()), 2.0)
    np.testing.assert_allclose(dodd.cpu().numpy(), odd + 2.0 * odd[0][None], rtol=1e-6, atol=1e-6)
    v = _rand(rng, 1003)
    np.testing.assert_array_equal(hip.cast_bf16(to_dev(v)).float().cpu().numpy(), bf16_round(v))


def test_ode_state_kernels(hip):
    rng = np.random.default_rng(8)
    n = 4 * 4 * 32 * 32 + 3
    y = _rand(rng, n)
    ks = [_rand(rng, n) for _ in range(7)]
    cs = [0.3, -1.2, 0.0, 2.5, 1e-3, -0.7, 0.11]
    out = torch.empty(n, device="cuda")
    hip.ode_combine(out, to_dev(y), [to_dev(k) for k in ks], cs)
    ref = y.copy()
    for k, c in zip(ks, cs):
        ref += np.float32(c) * k
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
    y1 = _rand(rng, n)
    scratch = torch.empty(1024, device="cuda")
    res = torch.empty(2, device="cuda")          # [rms, raw sum of squares]
    hip.ode_error_norm(to_dev(y), to_dev(y1), [to_dev(k) for k in ks], cs, 1e-3, 1e-4, scratch, res)
    err = np.zeros(n, np.float64)
    for k, c in zip(ks, cs):
        err += c * k.astype(np.float64)
    tol = 1e-4 + 1e-3 * np.maximum(np.abs(y), np.abs(y1))
    want = np.sqrt(np.mean((err / tol) ** 2))
    assert abs(float(res[0].item()) - want) / want < 1e-4
    assert abs(float(res[1].item()) - want * want * n) / (want * want * n) < 2e-4


@pytest.mark.parametrize("B,Ci,Co,H", [(2, 64, 128, 8), (1, 128, 64, 16), (3, 256, 256, 12)])
def test_conv3x3_as_nine_row_shifted_gemms(hip, B, Ci, Co, H):
    """Conv2d(Ci, Co, 3, padding=1) == sum over the 9 taps of GEMMs on a zero-bordered NHWC map whose rows are
    shifted by dy*(W+2)+dx (libs/autoencoder.py:85-112), checked on interior pixels against the oracle."""
    rng = np.random.default_rng(Ci + Co + H)
    W_ = H
    x = bf16_round(_rand(rng, B, Ci, H, W_))
    w = bf16_round(_rand(rng, Co, Ci, 3, 3) * 0.05)
    b = _rand(rng, Co)
    ref = C.conv2d(x, w, b)                                            # [B, Co, H, W]
    P = W_ + 2
    rows = B * (H + 2) * P
    guard = P + 1
    buf = torch.zeros(rows + 2 * guard, Ci, dtype=torch.bfloat16, device="cuda")
    pad = torch.zeros(B, H + 2, P, Ci)
    pad[:, 1:-1, 1:-1, :] = torch.from_numpy(x).permute(0, 2, 3, 1)
    buf[guard:guard + rows] = pad.reshape(rows, Ci).to(torch.bfloat16).cuda()
    A = buf[guard:guard + rows]
    wk = torch.from_numpy(w).permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous().to(torch.bfloat16).cuda()   # [Co][tap][Ci]
    shifts = [dy * P + dx for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    out = torch.empty(rows, Co, dtype=torch.float32, device="cuda")
    hip.gemm_slabs(A, wk, shifts, bias=to_dev(b), out_f32=out)
    got = out.view(B, H + 2, P, Co)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).cpu().numpy()
    assert rel_l2(got, ref) < 1e-5


@pytest.mark.parametrize("B,C_,H,silu", [(2, 64, 16, 1), (3, 128, 8, 0), (1, 256, 12, 1), (2, 512, 8, 1)])
def test_groupnorm_on_zero_bordered_map(hip, B, C_, H, silu):
    """GroupNorm(32, C, eps=1e-6) (+SiLU) of libs/autoencoder.py:26-32 on the NHWC map layout of the VAE path."""
    import ctypes
    rng = np.random.default_rng(C_ + H)
    x = (_rand(rng, B, C_, H, H, scale=2.0) + 0.5).astype(np.float32)
    g = _rand(rng, C_) + 1.0
    bt = _rand(rng, C_)
    ref = C.groupnorm(x, g, bt, groups=32, eps=1e-6)
    if silu:
        ref = ref / (1.0 + np.exp(-ref))
    pad = torch.full((B, H + 2, H + 2, C_), 7.0)                 # border holds garbage on purpose
    pad[:, 1:-1, 1:-1, :] = torch.from_numpy(x).permute(0, 2, 3, 1)
    xm = pad.contiguous().cuda()
    y = torch.empty(B, H + 2, H + 2, C_, dtype=torch.bfloat16, device="cuda")
    scratch = torch.empty(B * 257 * 64, device="cuda")
    gd, bd = to_dev(g), to_dev(bt)
    rc = hip.lib().uspace_groupnorm_map_bf16(hip.ptr(xm), hip.ptr(gd), hip.ptr(bd), hip.ptr(y), hip.ptr(scratch),
                                             B, H, C_, silu, ctypes.c_float(1e-6), hip.stream_ptr())
    assert rc == 0
    yy = y.float().cpu()
    got = yy[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).numpy()
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(ref).max() * 1.5 + 2e-3
    border = yy.clone()
    border[:, 1:-1, 1:-1, :] = 0
    assert float(border.abs().max()) == 0.0                       # zero border = the next conv's padding


@pytest.mark.parametrize("B,D,n_extra,time_first", [(3, 128, 0, 1), (2, 512, 1, 0), (2, 1024, 77, 1), (2, 64, 0, 1)])
def test_embed_tokens_matches_oracle(hip, B, D, n_extra, time_first):
    """PatchEmbed + timestep_embedding + [label | context] tokens + pos_embed (libs/uvit.py:26-46,171-179,315-327;
    libs/uvit_t2i.py:320-324), fp32 rows and their bf16 copy; D % 1024 != 0 and the register-weight patch kernel."""
    rng = np.random.default_rng(D + n_extra)
    S, p, Cc = 32, 2, 4
    g = S // p
    L = 1 + n_extra + g * g
    img = _rand(rng, B, Cc, S, S)
    t = rng.uniform(0.05, 0.95, B).astype(np.float32)
    pw, pb = _rand(rng, D, Cc, p, p, scale=0.2), _rand(rng, D, scale=0.1)
    pos = _rand(rng, L, D, scale=0.02)
    extra = _rand(rng, B, max(n_extra, 1), D)
    patches = C.patch_embed(img, pw, pb)
    ttok = C.timestep_embedding(t, D)[:, None, :]
    parts = [ttok, extra[:, :n_extra], patches] if time_first else [extra[:, :n_extra], ttok, patches]
    ref = np.concatenate(parts, axis=1) + pos[None]
    d = {k: to_dev(v) for k, v in dict(img=img, t=t, pw=pw, pb=pb, pos=pos, extra=extra).items()}
    tok = torch.empty(B, L, D, device="cuda")
    tokb = torch.empty(B, L, D, device="cuda", dtype=torch.bfloat16)
    rc = hip.lib().uspace_embed_tokens(hip.ptr(d["img"]), hip.ptr(d["t"]), 1, hip.ptr(d["extra"]) if n_extra else None, n_extra,
                                       time_first, hip.ptr(d["pw"]), hip.ptr(d["pb"]), hip.ptr(d["pos"]), hip.ptr(tok),
                                       hip.ptr(tokb), B, Cc, S, p, D, hip.stream_ptr())
    assert rc == 0
    got = tok.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert torch.equal(tokb, tok.to(torch.bfloat16))


@pytest.mark.parametrize("B,D,extras", [(3, 128, 1), (2, 512, 78), (5, 1024, 1), (2, 64, 2), (1, 1536, 1)])
def test_output_head_matches_oracle(hip, B, D, extras):
    """norm + decoder_pred + token slice + unpatchify + final 3x3 conv (libs/uvit.py:56-63,342-347): the LDS-weight
    kernel (D % 128 == 0, D <= 1024) and the per-token kernel."""
    import ctypes
    rng = np.random.default_rng(D + extras)
    S, p, Cc = 32, 2, 4
    g = S // p
    L = extras + g * g
    x = (_rand(rng, B, L, D, scale=1.5) + 0.3).astype(np.float32)
    ng, nb = _rand(rng, D) * 0.2 + 1.0, _rand(rng, D, scale=0.1)
    dw, db = _rand(rng, p * p * Cc, D, scale=0.05), _rand(rng, p * p * Cc, scale=0.1)
    cw, cb = _rand(rng, Cc, Cc, 3, 3, scale=0.2), _rand(rng, Cc, scale=0.1)
    h = C.layernorm(x, ng, nb, eps=1e-5)
    h = C.linear(h, dw, db)[:, extras:, :]
    ref = C.conv3x3(C.unpatchify(h, Cc), cw, cb)
    d = {k: to_dev(v) for k, v in dict(x=x, ng=ng, nb=nb, dw=dw, db=db, cw=cw, cb=cb).items()}
    scratch = torch.empty(B, Cc, S, S, device="cuda")
    out = torch.empty(B, Cc, S, S, device="cuda")
    rc = hip.lib().uspace_output_head(hip.ptr(d["x"]), L, extras, hip.ptr(d["ng"]), hip.ptr(d["nb"]), hip.ptr(d["dw"]),
                                      hip.ptr(d["db"]), hip.ptr(d["cw"]), hip.ptr(d["cb"]), hip.ptr(scratch), hip.ptr(out),
                                      B, Cc, S, p, D, ctypes.c_float(1e-5), hip.stream_ptr())
    assert rc == 0
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K,expect", [
    (64 * 334, 512, 128, 1),      # U-ViT-S T2I rows: 192x256 tiles (224 of them, one round), extra strips unused
    (16448, 1536, 64, 1),         # 192x256 with extra strips (80 tile rows + 14-row strips)
    (64 * 334, 1024, 64, None),   # U-ViT-L T2I proj/fc2 shape: 192x256 or the 256x256 + 128x128 split
    (16448, 1024, 64, 0),         # the headline shape: 256x256 with extra strips
    (32 * 257, 1024, 128, 4),     # config 5 rows: 256x128 tiles (32 x 8 = one round) with two extra strips
    (8 * 257, 4096, 64, 4),       # 256x128, 8 x 32 tiles, one strip
])
def test_gemm_tile_configurations_at_full_row_counts(hip, M, N, K, expect):
    """Every tile configuration the planner can pick at BASELINE row counts, with the fused proj/fc2 epilogue
    (bias + residual in place + bf16 copy); checked against the oracle on a sample of rows incl. the last ones."""
    import ctypes
    split = ctypes.c_int(0)
    choice = hip.lib().uspace_gemm_tile_choice(M, N, ctypes.byref(split))
    assert expect is None or choice == expect, (choice, split.value)
    rng = np.random.default_rng(M + N)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    R = _rand(rng, M, N)
    x = to_dev(R).clone()
    xb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), bias=to_dev(b), resid=x, out_f32=x, out_bf16=xb)
    rows = np.unique(np.concatenate([rng.integers(0, M, 1500), np.arange(M - 300, M), np.arange(0, 300),
                                     np.arange(max(split.value - 150, 0), min(split.value + 150, M))]))
    ref = C.linear(A[rows], W, b) + R[rows]
    got = x.cpu().numpy()[rows]
    np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    assert rel_l2(got, ref) < 1e-5
    assert torch.equal(xb, x.to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K,two_slabs", [
    (4 * 257, 512, 2048, False),   # U-ViT-S fc2 at batch 4: 36 tiles, K split in 4
    (4 * 257, 1024, 2048, True),   # U-ViT-L skip_linear ([x | skip], two K slabs) at batch 4: 72 tiles, K split in 2
    (4 * 257, 1024, 4096, False),  # U-ViT-L fc2 at batch 4: K split in 2
    (4 * 257, 1536, 1024, False),  # K too short to split with 108 tiles
    (4 * 257, 512, 1024, True),    # U-ViT-S skip_linear at batch 4: 36 tiles, K = 1024 split in 2
    (515, 256, 2048, False),       # ragged rows and few tiles: K split in 4
    (2 * 257, 512, 4096, False),   # 20 tiles: K split in 8
])
def test_gemm_small_launches_k_split(hip, M, N, K, two_slabs):
    """Small launches (128x128 tiles on a fraction of the CUs) given a workspace split a long K over gridDim.y; a second
    kernel adds the partial sums in split order and applies the epilogue (bias + residual in place + bf16 copy: proj / fc2 /
    skip_linear, libs/uvit.py:159-161).  Against the oracle, against the unsplit launch, and repeatedly over the same
    workspace (bit-identical results)."""
    import ctypes
    lib = hip.lib()
    split = ctypes.c_int(0)
    assert lib.uspace_gemm_tile_choice(M, N, ctypes.byref(split)) == 2
    rng = np.random.default_rng(M + N + K)
    K1 = K // 2 if two_slabs else K
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.05)
    b = _rand(rng, N)
    R = _rand(rng, M, N)
    ref = C.linear(A, W, b) + R
    dA = to_dev(np.ascontiguousarray(A[:, :K1]), torch.bfloat16)
    dA2 = to_dev(np.ascontiguousarray(A[:, K1:]), torch.bfloat16) if two_slabs else None
    dW, db, dR = to_dev(W, torch.bfloat16), to_dev(b), to_dev(R)
    need = lib.uspace_gemm_split_ws_bytes(M, N, K)
    assert (need > 0) == (K >= 2048 or (K >= 1024 and -(-M // 128) * -(-N // 128) <= 40)), need
    outs = []
    for ws_bytes in ([0, need] if need else [0]):
        x = torch.empty_like(dR)
        xb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ws = None
        if ws_bytes:
            ws = torch.full((ws_bytes // 4,), float("nan"), device="cuda")
        runs = []
        for _ in range(3 if ws_bytes else 1):
            x.copy_(dR)
            hip.gemm(dA, dW, A2=dA2, bias=db, resid=x, out_f32=x, out_bf16=xb, split_ws=ws)
            runs.append(x.clone())
        if ws_bytes:
            assert all(torch.equal(runs[0], r) for r in runs[1:])
        got = x.cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
        assert rel_l2(got, ref) < 1e-5
        assert torch.equal(xb, x.to(torch.bfloat16))
        outs.append(got)
    if len(outs) == 2:
        assert rel_l2(outs[1], outs[0]) < 1e-6
        # a workspace that is too small is ignored (unsplit launch), not an error
        x = dR.clone()
        hip.gemm(dA, dW, A2=dA2, bias=db, resid=x, out_f32=x, split_ws=torch.zeros(need // 4 - 64, device="cuda"))
        assert np.array_equal(x.cpu().numpy(), outs[0])


@pytest.mark.parametrize("M,D,Kp,N2", [
    (16448, 1024, 64, 1024),     # headline row count: 256x256 tiles with extra strips on both sides
    (16448, 1024, 64, 512),      # consumer on 128x128 tiles
    (32 * 257, 1024, 64, 1024),  # config 5 rows: producer and consumer on 256x128 tiles (4 x 2 waves) with extra strips
    (8 * 257, 1024, 64, 4096),   # consumer on 256x128 tiles with one strip
    (4 * 257, 512, 2048, 1536),  # small batch: producer in the K-split form (row sums in slot 0 of the partial-sum row)
    (64 * 334, 512, 64, 1536),   # U-ViT-S T2I rows: producer on 192x256 tiles
    (515, 256, 128, 256),        # small everything, ragged rows
    (4100, 64, 64, 256),         # consumer with a single K tile (no barrier inside its K loop)
])
def test_layernorm_folded_through_gemms(hip, M, D, Kp, N2):
    """Producer (x = A W^T + b + R, also centred bf16 copy + per-row partial sums) followed by a consumer computing
    LN(x; gamma, beta) W2^T + b2 from the centred copy (uspace_gemm_bf16_ext), against LayerNorm + Linear of the oracle
    (libs/uvit.py:135-161: norm1 -> qkv, norm2 -> fc1)."""
    import ctypes
    rng = np.random.default_rng(M + D + N2)
    A = bf16_round(_rand(rng, M, Kp))
    W = bf16_round(_rand(rng, D, Kp) * 0.2)
    b = _rand(rng, D)
    R = (_rand(rng, M, D) * 1.5 + _rand(rng, M, 1) * 2.0).astype(np.float32)      # rows with sizeable, different means
    gam, bet = (_rand(rng, D) * 0.2 + 1.0).astype(np.float32), _rand(rng, D, scale=0.1)
    W2 = (_rand(rng, N2, D) * 0.05).astype(np.float32)
    b2 = _rand(rng, N2)
    rows = np.unique(np.concatenate([rng.integers(0, M, 600), np.arange(min(M, 300)), np.arange(max(M - 300, 0), M)]))
    x_ref = C.linear(A[rows], W, b) + R[rows]
    y_ref = C.linear(C.layernorm(x_ref, gam, bet, eps=1e-5), W2, b2)
    # centring constants: the row mean of a *previous* state (here: of R alone) -- close to, not equal to, the new mean
    c = R.mean(axis=1).astype(np.float32)
    lib = hip.lib()
    slots = lib.uspace_gemm_part_slots_k(M, D, Kp)          # of THIS producer (64-wide tiles for few-tile, short-K launches)
    assert slots in (-(-D // 256), -(-D // 128), -(-D // 64)) and slots <= lib.uspace_gemm_part_slots(M, D)
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    x = to_dev(R).clone()
    xc = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    part = torch.full((M, slots, 2), float("nan"), device="cuda")
    dc = to_dev(c)
    ext = hip.GemmExt(hip.ptr(dc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
    ws_bytes = lib.uspace_gemm_split_ws_bytes(M, D, Kp)
    ws = torch.zeros(max(ws_bytes // 4, 4), device="cuda")
    ext.split_ws, ext.split_ws_bytes = hip.ptr(ws).value, ws_bytes
    flags = hip.EPI_BIAS | hip.EPI_RESIDUAL | hip.EPI_OUT_F32 | 32
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(dA), Kp, None, 0, Kp, hip.ptr(dW), Kp, M, D, Kp, flags, hip.ptr(db), hip.ptr(x), D,
                                  hip.ptr(x), D, None, 0, ctypes.byref(ext), hip.stream_ptr())
    assert rc == 0
    xg = x.cpu().numpy()
    np.testing.assert_allclose(xg[rows], x_ref, rtol=1e-3, atol=2e-3)
    assert torch.equal(xc, (x - dc[:, None]).to(torch.bfloat16))
    pg = part.cpu().numpy().astype(np.float64).sum(axis=1)
    cen = xg.astype(np.float64) - c[:, None]
    np.testing.assert_allclose(pg[:, 0], cen.sum(1), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(pg[:, 1], (cen ** 2).sum(1), rtol=1e-4)
    # consumer: gamma folded into the weights, beta into the bias, column sums of the bf16 weights
    W2g = bf16_round(W2 * gam[None, :])
    bias2 = (b2 + W2 @ bet).astype(np.float32)
    colsum = W2g.sum(axis=1).astype(np.float32)
    y = torch.empty(M, N2, dtype=torch.bfloat16, device="cuda")
    cout = torch.empty(M, device="cuda")
    dW2, dbias2, dcs = to_dev(W2g, torch.bfloat16), to_dev(bias2), to_dev(colsum)
    ext2 = hip.GemmExt(hip.ptr(dc).value, None, 0, None, hip.ptr(part).value, slots, hip.ptr(dcs).value, hip.ptr(cout).value, D,
                       1e-5)
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(xc), D, None, 0, D, hip.ptr(dW2), D, M, N2, D, hip.EPI_BIAS | hip.EPI_OUT_BF16 | 64,
                                  hip.ptr(dbias2), None, 0, None, 0, hip.ptr(y), N2, ctypes.byref(ext2), hip.stream_ptr())
    assert rc == 0
    yg = y.float().cpu().numpy()[rows]
    assert rel_l2(yg, y_ref) < 4e-3, rel_l2(yg, y_ref)
    np.testing.assert_allclose(cout.cpu().numpy(), xg.mean(axis=1), rtol=1e-4, atol=1e-4)
    # the plain entry point refuses the extended flags
    assert lib.uspace_gemm_bf16(hip.ptr(xc), D, None, 0, D, hip.ptr(dW2), D, M, N2, D, hip.EPI_OUT_BF16 | 64, None, None, 0,
                                None, 0, hip.ptr(y), N2, hip.stream_ptr()) != 0


@pytest.mark.parametrize("M,D", [(64 * 257, 1024), (32 * 257, 1024), (4 * 257, 512), (4 * 257, 1024), (515, 256), (64 * 334, 512)])
def test_skip_linear_with_a_centred_skip_slab(hip, M, D):
    """skip_linear(cat([x, skip])) (libs/uvit.py:158-159) with the skip stored as the centred bf16 copy its producer wrote for the
    next norm (skip_c = bf16(skip - c)): two K slabs + the rank-1 epilogue term c[m] * rowsum(bf16(W[:, D:]))[n] (USPACE_EPI_RANK1) on
    every tile form the planner picks, against the fp32 result of the concat, and the producer side outputs (centred copy, partial
    sums) of the same launch."""
    import ctypes
    rng = np.random.default_rng(M + D)
    xcur = bf16_round(_rand(rng, M, D))
    skip = (_rand(rng, M, D) + _rand(rng, M, 1) * 3.0).astype(np.float32)          # rows with large, different means
    c = (skip.mean(axis=1) + 0.05 * _rand(rng, M)).astype(np.float32)             # centring constants: close to the row means
    skip_c = bf16_round(skip - c[:, None])
    W = bf16_round(_rand(rng, D, 2 * D) * 0.05)
    b = _rand(rng, D)
    rows = np.unique(np.concatenate([rng.integers(0, M, 500), np.arange(min(M, 260)), np.arange(max(M - 300, 0), M)]))
    ref = C.linear(np.concatenate([xcur[rows], skip_c[rows] + c[rows, None]], axis=1), W, b)
    lib = hip.lib()
    slots = lib.uspace_gemm_part_slots_k(M, D, 2 * D)
    cs2 = W[:, D:].sum(axis=1).astype(np.float32)
    dx, ds, dW, db = to_dev(xcur, torch.bfloat16), to_dev(skip_c, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    drc = to_dev(_rand(rng, M) * 0.1)                       # this launch's own centring constants (producer role)
    dc, dcs2 = to_dev(c), to_dev(cs2)
    out = torch.empty(M, D, device="cuda")
    xc = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    part = torch.full((M, slots, 2), float("nan"), device="cuda")
    ext = hip.GemmExt(hip.ptr(drc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
    ext.row_add, ext.col_add = hip.ptr(dc).value, hip.ptr(dcs2).value
    ws_bytes = lib.uspace_gemm_split_ws_bytes(M, D, 2 * D)
    ws = torch.zeros(max(ws_bytes // 4, 4), device="cuda")
    ext.split_ws, ext.split_ws_bytes = hip.ptr(ws).value, ws_bytes
    flags = hip.EPI_RANK1 | hip.EPI_CEN_OUT | hip.EPI_BIAS | hip.EPI_OUT_F32
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(dx), D, hip.ptr(ds), D, D, hip.ptr(dW), 2 * D, M, D, 2 * D, flags, hip.ptr(db), None, 0,
                                  hip.ptr(out), D, None, 0, ctypes.byref(ext), hip.stream_ptr())
    assert rc == 0
    og = out.cpu().numpy()
    np.testing.assert_allclose(og[rows], ref, rtol=2e-4, atol=2e-3)
    assert torch.equal(xc, (out - drc[:, None]).to(torch.bfloat16))
    pg = part.cpu().numpy().astype(np.float64).sum(axis=1)
    cen = og.astype(np.float64) - drc.cpu().numpy()[:, None]
    np.testing.assert_allclose(pg[:, 0], cen.sum(1), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(pg[:, 1], (cen ** 2).sum(1), rtol=1e-4)
    # the flag needs its operands and a producer launch
    bad = hip.GemmExt(hip.ptr(drc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
    assert lib.uspace_gemm_bf16_ext(hip.ptr(dx), D, hip.ptr(ds), D, D, hip.ptr(dW), 2 * D, M, D, 2 * D, flags, hip.ptr(db), None, 0,
                                    hip.ptr(out), D, None, 0, ctypes.byref(bad), hip.stream_ptr()) != 0
    assert lib.uspace_gemm_bf16_ext(hip.ptr(dx), D, hip.ptr(ds), D, D, hip.ptr(dW), 2 * D, M, D, 2 * D, hip.EPI_RANK1 | hip.EPI_BIAS | hip.EPI_OUT_F32,
                                    hip.ptr(db), None, 0, hip.ptr(out), D, None, 0, ctypes.byref(ext), hip.stream_ptr()) != 0


@pytest.mark.parametrize("M,N,K,kind", [
    (4 * 257, 1536, 512, "ln_in"),          # U-ViT-S qkv at batch 4 (BASELINE config 1): consumer of the folded LayerNorm
    (4 * 257, 2048, 512, "bias_gelu_bf16"),  # fc1 shape
    (4 * 257, 512, 512, "producer"),        # proj: residual in place + centred copy + 8 partial-sum slots of 64 columns
    (4 * 257, 512, 1024, "producer_2slab"),  # skip_linear: two K slabs, producer
    (1030, 260, 192, "bias_resid_f32_bf16"),  # ragged rows and columns
    (70, 64, 64, "f32"),                    # one tile row and a 6-row strip
    (4 * 257, 512, 2048, "producer"),       # fc2 of the in-blocks: 32 K tiles through the four-stage ring, fused producer epilogue
    (4 * 257, 512, 2048, "bias_resid_f32_bf16"),   # fc2 of the mid / out blocks
    (1030, 256, 1024, "bias_gelu_bf16"),    # ring form, ragged rows (strip) and 16 K tiles
    (2 * 257, 4096, 1024, "bias_gelu_bf16"),  # fc1 of U-ViT-L at batch 2: ring form planned for 512 workgroups = 8 x 64 tiles + a 2-row strip
    (2 * 257 + 40, 4096, 1024, "bias_resid_f32_bf16"),   # ... three strips of 16 rows (42 remainder rows)
])
def test_gemm_64x64_tile_form(hip, M, N, K, kind):
    """Launches whose 128x128 tiling would fill 160 workgroups or fewer run as 64x64 tiles (round 3, `refine_small` in gemm.hip;
    K loops of 16 tiles or more -- and shorter ones of up to 448 tiles -- in the four-stage ring form): every epilogue family at
    the U-ViT-S batch-4 shapes against the oracle (libs/timm.py:106-112,
    libs/uvit.py:135-161), with the partial-sum slot count (64-wide) as the witness that this form is the one that ran."""
    import ctypes
    lib = hip.lib()
    assert -(-M // 128) * -(-N // 128) <= 160
    rng = np.random.default_rng(M * 7 + N + K)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    R = (_rand(rng, M, N) + _rand(rng, M, 1)).astype(np.float32)
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    if kind.startswith("producer"):
        slots = lib.uspace_gemm_part_slots_k(M, N, K)
        assert slots == -(-N // 64) and lib.uspace_gemm_part_slots_k(M, N, 4096) == -(-N // 64)     # every K: the long-K ring form
        c = R.mean(axis=1).astype(np.float32)
        ref = C.linear(A, W, b) + R
        x = to_dev(R).clone()
        xc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        part = torch.full((M, slots, 2), float("nan"), device="cuda")
        dc = to_dev(c)
        ext = hip.GemmExt(hip.ptr(dc).value, hip.ptr(xc).value, N, hip.ptr(part).value, None, 0, None, None, N, 1e-5)
        flags = hip.EPI_BIAS | hip.EPI_RESIDUAL | hip.EPI_OUT_F32 | 32
        if kind == "producer_2slab":
            K1 = K // 2
            a1, a2 = to_dev(np.ascontiguousarray(A[:, :K1]), torch.bfloat16), to_dev(np.ascontiguousarray(A[:, K1:]), torch.bfloat16)
            rc = lib.uspace_gemm_bf16_ext(hip.ptr(a1), K1, hip.ptr(a2), K1, K1, hip.ptr(dW), K, M, N, K, flags, hip.ptr(db), hip.ptr(x), N,
                                          hip.ptr(x), N, None, 0, ctypes.byref(ext), hip.stream_ptr())
        else:
            rc = lib.uspace_gemm_bf16_ext(hip.ptr(dA), K, None, 0, K, hip.ptr(dW), K, M, N, K, flags, hip.ptr(db), hip.ptr(x), N,
                                          hip.ptr(x), N, None, 0, ctypes.byref(ext), hip.stream_ptr())
        assert rc == 0
        xg = x.cpu().numpy()
        np.testing.assert_allclose(xg, ref, rtol=1e-3, atol=2e-3)
        assert torch.equal(xc, (x - dc[:, None]).to(torch.bfloat16))
        pg = part.cpu().numpy().astype(np.float64)
        assert np.isfinite(pg).all()
        cen = xg.astype(np.float64) - c[:, None]
        np.testing.assert_allclose(pg[:, :, 0].sum(1), cen.sum(1), rtol=1e-4, atol=2e-2)
        np.testing.assert_allclose(pg[:, :, 1].sum(1), (cen ** 2).sum(1), rtol=1e-4)
        # each slot is the sum over ITS 64 columns
        np.testing.assert_allclose(pg[:, 1, 0], cen[:, 64:128].sum(1), rtol=1e-4, atol=2e-2)
        return
    if kind == "ln_in":
        D = K
        x = (_rand(rng, M, D) * 1.5 + _rand(rng, M, 1) * 2.0).astype(np.float32)
        gam, bet = (_rand(rng, D) * 0.2 + 1.0).astype(np.float32), _rand(rng, D, scale=0.1)
        W2 = (_rand(rng, N, D) * 0.05).astype(np.float32)
        y_ref = C.linear(C.layernorm(x, gam, bet, eps=1e-5), W2, b)
        c = (x.mean(axis=1) + 0.1).astype(np.float32)
        xc_np = bf16_round(x - c[:, None])
        cen = (x - c[:, None]).astype(np.float64)
        part = np.zeros((M, 8, 2), np.float32)
        for q in range(8):
            part[:, q, 0] = cen[:, q * 64:(q + 1) * 64].sum(1)
            part[:, q, 1] = (cen[:, q * 64:(q + 1) * 64] ** 2).sum(1)
        W2g = bf16_round(W2 * gam[None, :])
        bias2 = (b + W2 @ bet).astype(np.float32)
        colsum = W2g.sum(axis=1).astype(np.float32)
        y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        cout = torch.empty(M, device="cuda")
        dxc, dW2, dbias2, dcs, dc, dpart = (to_dev(xc_np, torch.bfloat16), to_dev(W2g, torch.bfloat16), to_dev(bias2), to_dev(colsum),
                                            to_dev(c), to_dev(part))
        ext2 = hip.GemmExt(hip.ptr(dc).value, None, 0, None, hip.ptr(dpart).value, 8, hip.ptr(dcs).value, hip.ptr(cout).value, D, 1e-5)
        rc = lib.uspace_gemm_bf16_ext(hip.ptr(dxc), D, None, 0, D, hip.ptr(dW2), D, M, N, D, hip.EPI_BIAS | hip.EPI_OUT_BF16 | 64,
                                      hip.ptr(dbias2), None, 0, None, 0, hip.ptr(y), N, ctypes.byref(ext2), hip.stream_ptr())
        assert rc == 0
        assert rel_l2(y.float().cpu().numpy(), y_ref) < 4e-3
        np.testing.assert_allclose(cout.cpu().numpy(), x.mean(axis=1), rtol=1e-4, atol=1e-4)
        return
    ref = C.linear(A, W, b if "bias" in kind else None)
    if "gelu" in kind:
        ref = C.gelu(ref)
    if "resid" in kind:
        ref = ref + R
    x = to_dev(R).clone() if "resid" in kind else torch.full((M, N), float("nan"), device="cuda")
    xb = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.gemm(dA, dW, bias=db if "bias" in kind else None, resid=x if "resid" in kind else None, gelu="gelu" in kind,
             out_f32=x if "f32" in kind else None, out_bf16=xb if "bf16" in kind else None)
    if "f32" in kind:
        got = x.cpu().numpy()
        assert rel_l2(got, ref) < 1e-5
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    if "bf16" in kind:
        gotb = xb.float().cpu().numpy()
        assert (np.abs(gotb - ref) <= BF16_EPS * np.abs(ref) * 1.01 + 1e-3 * np.abs(ref).max()).all()


def test_launch_recorder_reports_every_gemm_and_attention_launch(hip):
    """uspace_prof_all_begin / _end (bench.py's roofline_all): launches are aggregated by (kind, flags, M, N, K) with plausible
    durations, nothing is recorded outside a begin / end pair, and the filtered recorder (uspace_prof_gemm_begin) still sees
    only its own key."""
    rng = np.random.default_rng(3)
    M, N, K = 515, 256, 128
    dA, dW = to_dev(bf16_round(_rand(rng, M, K)), torch.bfloat16), to_dev(bf16_round(_rand(rng, N, K)), torch.bfloat16)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty(M, N, device="cuda")
    B, L, H = 2, 257, 4
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
    hip.gemm(dA, dW, out_bf16=o1)                              # not recorded
    hip.prof_all_begin(64)
    for _ in range(3):
        hip.gemm(dA, dW, out_bf16=o1)
    for _ in range(2):
        hip.gemm(dA, dW, out_f32=o2)
    hip.attention(qkv, B, L, H)
    torch.cuda.synchronize()
    recs = hip.prof_all_end()
    hip.gemm(dA, dW, out_bf16=o1)                              # not recorded either
    assert hip.prof_all_end() == []
    by = {(r["kind"], r["flags"], r["M"], r["N"], r["K"]): r for r in recs}
    assert by[(0, hip.EPI_OUT_BF16, M, N, K)]["launches"] == 3 and by[(0, hip.EPI_OUT_F32, M, N, K)]["launches"] == 2
    assert by[(1, 0, B * H, L, 64)]["launches"] == 1
    assert all(0.0 < r["total_ms"] < 50.0 for r in recs) and len(recs) == 3
    hip.prof_gemm_begin(hip.EPI_OUT_F32, N, K, 16)
    hip.gemm(dA, dW, out_bf16=o1)
    hip.gemm(dA, dW, out_f32=o2)
    torch.cuda.synchronize()
    ms, n = hip.prof_gemm_end()
    assert n == 1 and 0.0 < ms < 50.0
    tf, gb, ghz = hip.prof_peaks(mfma_iters=2000, copy_bytes=1 << 26, copy_reps=2)
    assert 500.0 < tf < 2600.0 and 500.0 < gb < 8000.0 and 1.0 < ghz < 2.6     # the chip clocks to its power budget under the MFMA loop


def _sk_ext(hip, ext, M, N, K):
    """Give `ext` the workspace and a fresh set of zeroed arrival counters for the in-launch K-split tail; returns what must stay alive."""
    need = hip.lib().uspace_gemm_sk_ws_bytes(M, N, K)
    ws = torch.full((max(need // 4, 4),), float("nan"), device="cuda")
    cnt = torch.zeros(256, dtype=torch.int32, device="cuda")
    ext.sk_ws, ext.sk_ws_bytes, ext.sk_counters = hip.ptr(ws).value, need, hip.ptr(cnt).value
    return need, ws, cnt


@pytest.mark.parametrize("M,N,K,S,n_dp", [
    (64 * 334, 1024, 4096, 3, 256),   # config 3 fc2: 83 tile rows + 8 strips = 332 tiles: one whole round + 76 tiles in 3 K parts (21 / 21 / 22 K tiles)
    (32 * 257, 1024, 4096, 2, 0),     # config 5 fc2: 128 tiles + 2 strips, every tile in 2 K parts
    (16 * 257, 1024, 4096, 4, 0),     # 64 tiles in 4 K parts (16 K tiles each)
    (96 * 257, 1024, 4096, 2, 256),   # 384 tiles = one round + 128 tiles in 2 parts
    (30 * 256 + 500, 1024, 4096, 2, 0),   # 31 tile rows + 16 strips (244 rows left), 124 tiles in 2 parts
    (10 * 256 + 200, 1024, 4096, 4, 0),   # 200 rows left are more than 10 tile rows can take as strips: an 11th, partly filled tile row, no strips; 44 tiles in 4 parts
    (20 * 257 + 200, 1024, 4096, 3, 0),   # 20 tile rows + 14 strips: 80 tiles in 3 parts
    (24 * 257, 1024, 4096, 0, 0),         # 96 tiles: two parts would leave a quarter of the CUs idle -- no tail
    (64 * 334, 1024, 1024, 0, 0),     # K below 64 tiles: no tail (the exchange costs more than the K loop it saves, profiles/r06_sk_tail.md)
])
def test_gemm_k_split_tail_plain_epilogues(hip, M, N, K, S, n_dp):
    """The in-launch K-split tail (round 6): launches whose 256x256 tiles do not fill whole rounds share each remaining tile between
    S workgroups over K parts.  bias + residual in place + bf16 copy (fc2 of the out-blocks, libs/uvit.py:161) against the oracle
    on sampled rows, against the same launch without the workspace, bit-identical over repeats on a NaN-poisoned workspace that is
    reused with other data in between (a stale slab line would show)."""
    import ctypes
    lib = hip.lib()
    plan = (ctypes.c_int * 8)()
    assert lib.uspace_gemm_plan_k(M, N, K, 0, plan) == 0
    need = lib.uspace_gemm_sk_ws_bytes(M, N, K)
    if S == 0:
        assert plan[0] != 6 and need == 0
        return
    assert plan[0] == 6 and (plan[1], plan[7]) == (S, n_dp), list(plan)
    assert need > 0
    rng = np.random.default_rng(M + N + K)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.05)
    b = _rand(rng, N)
    R = _rand(rng, M, N)
    rows = np.unique(np.concatenate([rng.integers(0, M, 1200), np.arange(M - 300, M), np.arange(0, 300),
                                     np.arange(16384 - 150, min(16384 + 150, M)) if M > 16384 else np.arange(0)]))
    ref = C.linear(A[rows], W, b) + R[rows]
    dA, dW, db, dR = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b), to_dev(R)
    ws = torch.full((need // 4,), float("nan"), device="cuda")
    other_A = to_dev(bf16_round(_rand(rng, M, K)), torch.bfloat16)
    runs = []
    for rep in range(3):
        x = dR.clone()
        xb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        hip.gemm(dA, dW, bias=db, resid=x, out_f32=x, out_bf16=xb, sk_ws=ws)
        runs.append((x, xb))
        y = dR.clone()                                   # the same workspace with other operands in between
        hip.gemm(other_A, dW, bias=db, resid=y, out_f32=y, sk_ws=ws)
    torch.cuda.synchronize()
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    x, xb = runs[0]
    got = x.cpu().numpy()
    np.testing.assert_allclose(got[rows], ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    assert rel_l2(got[rows], ref) < 1e-5
    assert torch.equal(xb, x.to(torch.bfloat16))
    # without the workspace: another tile form, the same numbers up to fp32 summation order -- on EVERY element
    x0 = dR.clone()
    hip.gemm(dA, dW, bias=db, resid=x0, out_f32=x0)
    assert rel_l2(got, x0.cpu().numpy()) < 1e-6
    assert float((x - x0).abs().max()) < 1e-3 * float(x0.abs().max())


@pytest.mark.parametrize("M,D,Kp,N2", [
    (64 * 334, 1024, 4096, 1024),   # config 3: fc2 of an in-block as producer (one round + 76 tiles in 3 K parts, strips)
    (32 * 257, 1024, 4096, 3072),   # config 5: fc2 as producer (every tile in 2 K parts)
    (16 * 257, 1024, 4096, 512),    # 4 K parts
])
def test_layernorm_folded_through_gemms_with_k_split_tail(hip, M, D, Kp, N2):
    """test_layernorm_folded_through_gemms at the row counts where producers / consumers take the in-launch K-split tail: the
    producer's centred copy and per-row partial sums (one stride of N / 256 slots per row for the whole launch), the consumer's
    normalised output and the row means it publishes."""
    import ctypes
    lib = hip.lib()
    plan = (ctypes.c_int * 8)()
    assert lib.uspace_gemm_plan_k(M, D, Kp, 1, plan) == 0 and plan[0] == 6, list(plan)
    rng = np.random.default_rng(M + D + N2 + Kp)
    A = bf16_round(_rand(rng, M, Kp))
    W = bf16_round(_rand(rng, D, Kp) * 0.05)
    b = _rand(rng, D)
    R = (_rand(rng, M, D) * 1.5 + _rand(rng, M, 1) * 2.0).astype(np.float32)
    gam, bet = (_rand(rng, D) * 0.2 + 1.0).astype(np.float32), _rand(rng, D, scale=0.1)
    W2 = (_rand(rng, N2, D) * 0.05).astype(np.float32)
    b2 = _rand(rng, N2)
    rows = np.unique(np.concatenate([rng.integers(0, M, 600), np.arange(300), np.arange(M - 300, M)]))
    x_ref = C.linear(A[rows], W, b) + R[rows]
    y_ref = C.linear(C.layernorm(x_ref, gam, bet, eps=1e-5), W2, b2)
    c = R.mean(axis=1).astype(np.float32)
    slots = lib.uspace_gemm_part_slots_k(M, D, Kp)
    assert slots == D // 256
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    x = to_dev(R).clone()
    xc = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    part = torch.full((M, slots, 2), float("nan"), device="cuda")
    dc = to_dev(c)
    ext = hip.GemmExt(hip.ptr(dc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
    keep = _sk_ext(hip, ext, M, D, Kp)
    assert keep[0] > 0
    flags = hip.EPI_BIAS | hip.EPI_RESIDUAL | hip.EPI_OUT_F32 | hip.EPI_CEN_OUT
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(dA), Kp, None, 0, Kp, hip.ptr(dW), Kp, M, D, Kp, flags, hip.ptr(db), hip.ptr(x), D,
                                  hip.ptr(x), D, None, 0, ctypes.byref(ext), hip.stream_ptr())
    assert rc == 0
    xg = x.cpu().numpy()
    np.testing.assert_allclose(xg[rows], x_ref, rtol=1e-3, atol=2e-3)
    assert torch.equal(xc, (x - dc[:, None]).to(torch.bfloat16))
    pgr = part.cpu().numpy()
    assert np.isfinite(pgr).all()
    pg = pgr.astype(np.float64).sum(axis=1)
    cen = xg.astype(np.float64) - c[:, None]
    np.testing.assert_allclose(pg[:, 0], cen.sum(1), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(pg[:, 1], (cen ** 2).sum(1), rtol=1e-4)
    # the same producer without the workspace: plain 256x256 tiles, the same partial-sum stride
    x1 = to_dev(R).clone()
    xc1 = torch.empty_like(xc)
    part1 = torch.full((M, slots, 2), float("nan"), device="cuda")
    ext1 = hip.GemmExt(hip.ptr(dc).value, hip.ptr(xc1).value, D, hip.ptr(part1).value, None, 0, None, None, D, 1e-5)
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(dA), Kp, None, 0, Kp, hip.ptr(dW), Kp, M, D, Kp, flags, hip.ptr(db), hip.ptr(x1), D,
                                  hip.ptr(x1), D, None, 0, ctypes.byref(ext1), hip.stream_ptr())
    assert rc == 0
    assert rel_l2(xg, x1.cpu().numpy()) < 1e-6
    np.testing.assert_allclose(part1.cpu().numpy().astype(np.float64).sum(axis=1), pg, rtol=1e-4, atol=2e-2)
    # consumer
    W2g = bf16_round(W2 * gam[None, :])
    bias2 = (b2 + W2 @ bet).astype(np.float32)
    colsum = W2g.sum(axis=1).astype(np.float32)
    y = torch.empty(M, N2, dtype=torch.bfloat16, device="cuda")
    cbuf = dc.clone()                                          # published IN PLACE, as the forward does (c <- c + d)
    dW2, dbias2, dcs = to_dev(W2g, torch.bfloat16), to_dev(bias2), to_dev(colsum)
    ext2 = hip.GemmExt(hip.ptr(cbuf).value, None, 0, None, hip.ptr(part).value, slots, hip.ptr(dcs).value, hip.ptr(cbuf).value, D, 1e-5)
    keep2 = None
    rc = lib.uspace_gemm_bf16_ext(hip.ptr(xc), D, None, 0, D, hip.ptr(dW2), D, M, N2, D, hip.EPI_BIAS | hip.EPI_OUT_BF16 | hip.EPI_LN_IN,
                                  hip.ptr(dbias2), None, 0, None, 0, hip.ptr(y), N2, ctypes.byref(ext2), hip.stream_ptr())
    assert rc == 0
    yg = y.float().cpu().numpy()[rows]
    assert rel_l2(yg, y_ref) < 4e-3, rel_l2(yg, y_ref)
    np.testing.assert_allclose(cbuf.cpu().numpy(), xg.mean(axis=1), rtol=1e-4, atol=1e-4)
    del keep, keep2
