"""LAB test (round 5; the form was measured and not landed: profiles/r05_gemm4.md).  Needs the lab library:
    tools/lab/gemm4/build.sh && USPACE_HIP_LIB=tools/lab/_build/lib_gemm4.so python -m pytest tools/lab/gemm4/test_form4.py -m gpu -q
The four-wave GEMM form (gemm4.hip: 4 waves x 128 x 128, accumulators in AGPRs, K loop in assembly -- kloop4.inc) against the 8-wave
template on 256 x 256 launches, both through the C-ABI (uspace_lab_gemm_set_big_form), and against the oracle.
Reference operators: nn.Linear of libs/timm.py:106-112 (fc1 / fc2), libs/uvit.py:89,116 (qkv / proj), :158-159 (skip_linear).

Without a residual the two forms run the same MFMAs in the same order per output element: outputs must be BIT-equal.  With a
residual the four-wave form adds it behind the K loop instead of starting the accumulators from it: fp32 rounding only."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import _cops as C
from tests.util import bf16_round, rel_l2, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from uspace_amd import _hip
    lib = _hip.lib()
    if not hasattr(lib, "uspace_lab_gemm_set_big_form"):
        pytest.skip("the loaded library has no four-wave form: build tools/lab/gemm4 and point USPACE_HIP_LIB at it")
    yield _hip
    lib.uspace_lab_gemm_set_big_form(1)


def _rand(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _both_forms(hip, fn):
    """run fn() under the 8-wave forms the planner picks, then under the forced four-wave form; returns the two results"""
    lib = hip.lib()
    out = []
    for form in (1, 2):
        assert lib.uspace_lab_gemm_set_big_form(form) in (0, 1, 2)
        out.append(fn())
    lib.uspace_lab_gemm_set_big_form(1)
    return out


def _takes4(hip, *a):
    lib = hip.lib()
    lib.uspace_lab_gemm_set_big_form(2)
    r = lib.uspace_lab_gemm_takes_form4(*a)
    lib.uspace_lab_gemm_set_big_form(1)
    return r


# (M, N, K): whole tile rows | one strip per 16 tile rows (64 * 257 rows: the headline) | strips with a ragged last one | K = 128 .. 4096
SHAPES = [(512, 512, 256), (2048, 1024, 128), (4112, 512, 512), (64 * 257, 1024, 1024), (8 * 257 + 5 + 2048, 768, 384), (1024, 256, 4096)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_plain_epilogues_bit_equal_between_forms(hip, M, N, K):
    rng = np.random.default_rng(M + N + K)
    lib = hip.lib()
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    rows = np.unique(np.concatenate([rng.integers(0, M, 300), np.arange(min(M, 40)), np.arange(max(M - 300, 0), M)]))
    lin = C.linear(A[rows], W, b)
    for kind in ("bf16", "bias_bf16", "bias_gelu_bf16", "bias_f32", "bias_f32_bf16"):
        flags = (hip.EPI_BIAS if "bias" in kind else 0) | (hip.EPI_GELU if "gelu" in kind else 0) | \
                (hip.EPI_OUT_F32 if "f32" in kind else 0) | (hip.EPI_OUT_BF16 if "bf16" in kind else 0)
        assert _takes4(hip, M, N, K, K, flags) == 1, (M, N, K, kind)

        def run():
            o32 = torch.full((M, N), float("nan"), device="cuda") if "f32" in kind else None
            o16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda") if "bf16" in kind else None
            hip.gemm(dA, dW, bias=db if "bias" in kind else None, gelu="gelu" in kind, out_f32=o32, out_bf16=o16)
            return o32, o16
        (a32, a16), (b32, b16) = _both_forms(hip, run)
        if a32 is not None:
            assert torch.equal(a32, b32), (kind, M, N, K)
            ref = lin if "bias" in kind else lin - b[None, :]
            assert rel_l2(b32.cpu().numpy()[rows], ref) < 1e-5
        if a16 is not None:
            assert torch.equal(a16, b16), (kind, M, N, K)
            assert bool(torch.isfinite(b16.float()).all())
    # repeated launches of the four-wave form agree with themselves (barrier / LDS-DMA race screen)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm(dA, dW, out_bf16=o1)
    for _ in range(3):
        o2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.gemm(dA, dW, out_bf16=o2)
        assert torch.equal(o1, o2)


@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (64 * 257, 1024, 1024), (4112, 512, 4096)])
def test_residual_form_matches_the_oracle_and_the_8_wave_form(hip, M, N, K):
    rng = np.random.default_rng(M + N + K + 1)
    A = bf16_round(_rand(rng, M, K))
    W = bf16_round(_rand(rng, N, K) * 0.1)
    b = _rand(rng, N)
    R = (_rand(rng, M, N) * 2.0).astype(np.float32)
    dA, dW, db = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    assert _takes4(hip, M, N, K, K, hip.EPI_BIAS | hip.EPI_RESIDUAL | hip.EPI_OUT_F32) == 1

    def run():
        x = to_dev(R).clone()
        hip.gemm(dA, dW, bias=db, resid=x, out_f32=x)       # in place, as proj / fc2 run (libs/uvit.py:160-161)
        return x
    x8, x4 = _both_forms(hip, run)
    rows = np.unique(np.concatenate([rng.integers(0, M, 300), np.arange(min(M, 40)), np.arange(max(M - 300, 0), M)]))
    ref = C.linear(A[rows], W, b) + R[rows]
    assert rel_l2(x4.cpu().numpy()[rows], ref) < 1e-5
    d = (x4 - x8).abs().max().item()
    assert d <= 4e-6 * float(np.abs(ref).max()) + 1e-6, d


@pytest.mark.parametrize("M,D", [(2048, 512), (64 * 257, 1024), (4112, 1024)])
def test_two_slab_rank1_producer_between_forms(hip, M, D):
    """skip_linear with the centred skip slab (libs/uvit.py:158-159): two K slabs, rank-1 term, producer outputs."""
    rng = np.random.default_rng(M + D)
    lib = hip.lib()
    xcur = bf16_round(_rand(rng, M, D))
    skip_c = bf16_round(_rand(rng, M, D))
    c = _rand(rng, M)
    W = bf16_round(_rand(rng, D, 2 * D) * 0.05)
    b = _rand(rng, D)
    cs2 = W[:, D:].sum(axis=1).astype(np.float32)
    dx, ds, dW, db = to_dev(xcur, torch.bfloat16), to_dev(skip_c, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b)
    drc, dc, dcs2 = to_dev(_rand(rng, M) * 0.1), to_dev(c), to_dev(cs2)
    flags = hip.EPI_RANK1 | hip.EPI_CEN_OUT | hip.EPI_BIAS | hip.EPI_OUT_F32
    assert _takes4(hip, M, D, 2 * D, D, flags) == 1

    def run():
        slots = lib.uspace_gemm_part_slots_k(M, D, 2 * D)          # of the form in force
        out = torch.full((M, D), float("nan"), device="cuda")
        xc = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        part = torch.full((M, slots, 2), float("nan"), device="cuda")
        ext = hip.GemmExt(hip.ptr(drc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
        ext.row_add, ext.col_add = hip.ptr(dc).value, hip.ptr(dcs2).value
        rc = lib.uspace_gemm_bf16_ext(hip.ptr(dx), D, hip.ptr(ds), D, D, hip.ptr(dW), 2 * D, M, D, 2 * D, flags, hip.ptr(db), None, 0,
                                      hip.ptr(out), D, None, 0, ctypes.byref(ext), hip.stream_ptr())
        assert rc == 0
        return out, xc, part
    (o8, c8, p8), (o4, c4, p4) = _both_forms(hip, run)
    assert torch.equal(o8, o4) and torch.equal(c8, c4)          # no residual: same MFMAs, same epilogue arithmetic
    rows = np.unique(np.concatenate([rng.integers(0, M, 300), np.arange(max(M - 300, 0), M)]))
    ref = C.linear(np.concatenate([xcur[rows], skip_c[rows] + c[rows, None]], axis=1), W, b)
    np.testing.assert_allclose(o4.cpu().numpy()[rows], ref, rtol=2e-4, atol=2e-3)
    # partial sums: the same values added over two column groups per tile instead of four
    s8, s4 = p8.double().sum(1).cpu().numpy(), p4.double().sum(1).cpu().numpy()
    np.testing.assert_allclose(s4[:, 0], s8[:, 0], rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(s4[:, 1], s8[:, 1], rtol=1e-5)


@pytest.mark.parametrize("M,D,Kp,N2", [(2048, 512, 256, 1536), (64 * 257, 1024, 1024, 3072), (4112, 1024, 4096, 4096)])
def test_layernorm_fold_chain_between_forms(hip, M, D, Kp, N2):
    """producer (proj / fc2: + residual, centred copy, partial sums) -> consumer (norm -> qkv / fc1 + GELU) on both forms."""
    rng = np.random.default_rng(M + D + N2)
    lib = hip.lib()
    A = bf16_round(_rand(rng, M, Kp))
    W = bf16_round(_rand(rng, D, Kp) * 0.2)
    b = _rand(rng, D)
    R = (_rand(rng, M, D) * 1.5 + _rand(rng, M, 1) * 2.0).astype(np.float32)
    gam, bet = (_rand(rng, D) * 0.2 + 1.0).astype(np.float32), _rand(rng, D, scale=0.1)
    W2 = (_rand(rng, N2, D) * 0.05).astype(np.float32)
    b2 = _rand(rng, N2)
    c = R.mean(axis=1).astype(np.float32)
    dA, dW, db, dc = to_dev(A, torch.bfloat16), to_dev(W, torch.bfloat16), to_dev(b), to_dev(c)
    W2g = bf16_round(W2 * gam[None, :])
    bias2 = (b2 + W2 @ bet).astype(np.float32)
    colsum = W2g.sum(axis=1).astype(np.float32)
    dW2, dbias2, dcs = to_dev(W2g, torch.bfloat16), to_dev(bias2), to_dev(colsum)
    pflags = hip.EPI_BIAS | hip.EPI_RESIDUAL | hip.EPI_OUT_F32 | hip.EPI_CEN_OUT
    assert _takes4(hip, M, D, Kp, Kp, pflags) == 1

    def run():
        slots = lib.uspace_gemm_part_slots_k(M, D, Kp)             # of the form in force
        x = to_dev(R).clone()
        xc = torch.full((M, D), float("nan"), dtype=torch.bfloat16, device="cuda")
        part = torch.full((M, slots, 2), float("nan"), device="cuda")
        ext = hip.GemmExt(hip.ptr(dc).value, hip.ptr(xc).value, D, hip.ptr(part).value, None, 0, None, None, D, 1e-5)
        assert lib.uspace_gemm_bf16_ext(hip.ptr(dA), Kp, None, 0, Kp, hip.ptr(dW), Kp, M, D, Kp, pflags, hip.ptr(db), hip.ptr(x), D,
                                        hip.ptr(x), D, None, 0, ctypes.byref(ext), hip.stream_ptr()) == 0
        ys = []
        for gelu in (False, True):
            y = torch.full((M, N2), float("nan"), dtype=torch.bfloat16, device="cuda")
            cout = torch.full((M,), float("nan"), device="cuda")
            ext2 = hip.GemmExt(hip.ptr(dc).value, None, 0, None, hip.ptr(part).value, slots, hip.ptr(dcs).value, hip.ptr(cout).value, D, 1e-5)
            fl = hip.EPI_BIAS | hip.EPI_OUT_BF16 | hip.EPI_LN_IN | (hip.EPI_GELU if gelu else 0)
            assert lib.uspace_gemm_bf16_ext(hip.ptr(xc), D, None, 0, D, hip.ptr(dW2), D, M, N2, D, fl, hip.ptr(dbias2), None, 0, None, 0,
                                            hip.ptr(y), N2, ctypes.byref(ext2), hip.stream_ptr()) == 0
            ys.append((y, cout))
        return x, xc, part, ys
    (x8, c8, p8, y8), (x4, c4, p4, y4) = _both_forms(hip, run)
    rows = np.unique(np.concatenate([rng.integers(0, M, 300), np.arange(min(M, 40)), np.arange(max(M - 300, 0), M)]))
    x_ref = C.linear(A[rows], W, b) + R[rows]
    np.testing.assert_allclose(x4.cpu().numpy()[rows], x_ref, rtol=1e-3, atol=2e-3)
    assert (x4 - x8).abs().max().item() <= 1e-5 * float(np.abs(x_ref).max())
    assert torch.equal(c4, (x4 - dc[:, None]).to(torch.bfloat16))
    pg = p4.double().sum(1).cpu().numpy()
    cen = x4.double().cpu().numpy() - c[:, None]
    np.testing.assert_allclose(pg[:, 0], cen.sum(1), rtol=1e-4, atol=2e-2)
    np.testing.assert_allclose(pg[:, 1], (cen ** 2).sum(1), rtol=1e-4)
    y_ref = C.linear(C.layernorm(x_ref, gam, bet, eps=1e-5), W2, b2)
    for (ya, ca), (yb, cb), ref in zip(y8, y4, (y_ref, C.gelu(y_ref))):
        got = yb.float().cpu().numpy()[rows]
        assert rel_l2(got, ref) < 4e-3, rel_l2(got, ref)
        assert rel_l2(yb.float().cpu().numpy()[rows], ya.float().cpu().numpy()[rows]) < 3e-3
        np.testing.assert_allclose(cb.cpu().numpy(), x4.cpu().numpy().mean(axis=1), rtol=1e-4, atol=1e-4)
