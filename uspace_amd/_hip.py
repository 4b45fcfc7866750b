"""ctypes binding of libuspace_hip.so (include/uspace_hip.h).

There is NO fallback: if the shared library is missing or an entry point returns an error the
caller gets an exception.  torch is used only to own device memory and to name the stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (USPACE_HIP_LIB: another build of the same ABI, for A/B measurements of whole solves on one box -- tools/lab/.  Lab builds can be WRONG on
# purpose (ablations that skip stores, forms that were not landed): lib() says so on stderr and library_info() carries it into bench.py's line.)
_DEFAULT_LIB = os.path.join(_HERE, "libuspace_hip.so")
LIB_PATH = os.environ.get("USPACE_HIP_LIB") or _DEFAULT_LIB

EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_OUT_F32, EPI_OUT_BF16 = 1, 2, 4, 8, 16
EPI_CEN_OUT, EPI_LN_IN, EPI_RANK1 = 32, 64, 128          # uspace_gemm_bf16_ext only (LayerNorm folded through the GEMMs)

ABI_VERSION = 11

_ERR = {-1: "USPACE_ERR_ARG", -2: "USPACE_ERR_LAUNCH", -3: "USPACE_ERR_WORKSPACE"}


class UspaceHipError(RuntimeError):
    pass


class UvitConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "img_size", "patch_size", "in_chans", "embed_dim", "depth", "num_heads", "mlp_hidden",
        "n_extra", "clip_dim", "time_first")]


class UvitIO(ctypes.Structure):
    _fields_ = [
        ("x", ctypes.c_void_p), ("t", ctypes.c_void_p), ("t_stride", ctypes.c_int),
        ("context", ctypes.c_void_p), ("mid_delta", ctypes.c_void_p), ("mid_scale", ctypes.c_float),
        ("mid_tap", ctypes.c_void_p), ("key_scale", ctypes.c_void_p), ("out", ctypes.c_void_p),
        ("mid_row_scale", ctypes.c_void_p)]


class VaeConfig(ctypes.Structure):
    _fields_ = [("ch", ctypes.c_int), ("ch_mult", ctypes.c_int * 4), ("n_levels", ctypes.c_int),
                ("num_res_blocks", ctypes.c_int), ("resolution", ctypes.c_int)]


class ClipConfig(ctypes.Structure):
    _fields_ = [("vocab", ctypes.c_int), ("dim", ctypes.c_int), ("heads", ctypes.c_int), ("layers", ctypes.c_int),
                ("ffn", ctypes.c_int), ("max_pos", ctypes.c_int), ("eps", ctypes.c_float)]


class GemmExt(ctypes.Structure):
    _fields_ = [("row_c", ctypes.c_void_p), ("out_cen", ctypes.c_void_p), ("ld_cen", ctypes.c_int),
                ("part_out", ctypes.c_void_p), ("part_in", ctypes.c_void_p), ("np_in", ctypes.c_int),
                ("colsum", ctypes.c_void_p), ("c_out", ctypes.c_void_p), ("norm_dim", ctypes.c_int), ("eps", ctypes.c_float),
                ("row_add", ctypes.c_void_p), ("col_add", ctypes.c_void_p),
                ("split_ws", ctypes.c_void_p), ("split_ws_bytes", ctypes.c_size_t),
                ("sk_ws", ctypes.c_void_p), ("sk_ws_bytes", ctypes.c_size_t), ("sk_counters", ctypes.c_void_p)]


_P, _I, _L, _F, _SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/uspace_hip.h declares
SIGNATURES = {
    "uspace_abi_version": (_I, []),
    "uspace_gemm_bf16": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "uspace_gemm_bf16_ext": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P, _I,
                                  ctypes.POINTER(GemmExt), _P]),
    "uspace_gemm_part_slots": (_I, [_I, _I]),
    "uspace_gemm_part_slots_k": (_I, [_I, _I, _I]),
    "uspace_gemm_split_ws_bytes": (_SZ, [_I, _I, _I]),
    "uspace_gemm_sk_ws_bytes": (_SZ, [_I, _I, _I]),
    "uspace_gemm_set_sk": (_I, [_I]),
    "uspace_gemm_get_sk": (_I, []),
    "uspace_fold_layernorm": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "uspace_center_rows": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "uspace_uvit_set_ln_fold": (_I, [_I]),
    "uspace_uvit_get_ln_fold": (_I, []),
    "uspace_gemm_tile_choice": (_I, [_I, _I, ctypes.POINTER(_I)]),
    "uspace_gemm_plan": (_I, [_I, _I, ctypes.POINTER(_I)]),
    "uspace_gemm_plan_k": (_I, [_I, _I, _I, _I, ctypes.POINTER(_I)]),
    "uspace_gemm_slabs_bf16": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, ctypes.POINTER(_I), _I, _P, _P, _I, _P, _I, _P, _I, _P]),
    "uspace_layernorm_f32_bf16": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "uspace_attention_bf16": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "uspace_embed_tokens": (_I, [_P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "uspace_output_head": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "uspace_add_broadcast": (_I, [_P, _P, _P, _F, _I, _L, _P]),
    "uspace_add_broadcast_rows": (_I, [_P, _P, _P, _F, _P, _I, _L, _P]),
    "uspace_cast_f32_bf16": (_I, [_P, _P, _L, _P]),
    "uspace_direction_accumulate": (_I, [_P, _P, _P, _P, _I, _L, _I, _P]),
    "uspace_center_cols_f32": (_I, [_P, _P, _I, _L, _P]),
    "uspace_gram_f64": (_I, [_P, _P, _I, _L, _P]),
    "uspace_project_rows_f64": (_I, [_P, _P, _P, _I, _I, _L, _P]),
    "uspace_normalize_rows_signed": (_I, [_P, _I, _L, _P]),
    "uspace_ode_combine": (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _L, _P]),
    "uspace_ode_error_norm": (_I, [_P, _P, ctypes.POINTER(_P), ctypes.POINTER(_F), _I, _F, _F, _L, _P, _P, _P]),
    "uspace_uvit_num_params": (_I, [ctypes.POINTER(UvitConfig)]),
    "uspace_uvit_param_numel": (_L, [ctypes.POINTER(UvitConfig), _I]),
    "uspace_uvit_weight_bytes": (_SZ, [ctypes.POINTER(UvitConfig)]),
    "uspace_uvit_workspace_bytes": (_SZ, [ctypes.POINTER(UvitConfig), _I]),
    "uspace_uvit_pack_weights": (_I, [ctypes.POINTER(UvitConfig), ctypes.POINTER(_P), _I, _P, _SZ, _P]),
    "uspace_uvit_forward": (_I, [ctypes.POINTER(UvitConfig), _P, _P, _SZ, ctypes.POINTER(UvitIO), _I, _P]),
    "uspace_uvit_graph_create": (_I, [ctypes.POINTER(UvitConfig), _P, _P, _SZ, ctypes.POINTER(UvitIO), _I, _P,
                                      ctypes.POINTER(_P)]),
    "uspace_uvit_graph_launch": (_I, [_P, _P]),
    "uspace_uvit_graph_destroy": (_I, [_P]),
    "uspace_groupnorm_map_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "uspace_vae_num_params": (_I, [ctypes.POINTER(VaeConfig)]),
    "uspace_vae_param_numel": (_L, [ctypes.POINTER(VaeConfig), _I]),
    "uspace_vae_weight_bytes": (_SZ, [ctypes.POINTER(VaeConfig)]),
    "uspace_vae_workspace_bytes": (_SZ, [ctypes.POINTER(VaeConfig), _I]),
    "uspace_vae_pack_weights": (_I, [ctypes.POINTER(VaeConfig), ctypes.POINTER(_P), _I, _P, _SZ, _P]),
    "uspace_vae_decode": (_I, [ctypes.POINTER(VaeConfig), _P, _P, _SZ, _P, _F, _P, _I, _P]),
    "uspace_vae_decode_tap": (_I, [ctypes.POINTER(VaeConfig), _P, _P, _SZ, _P, _F, _I, _I, _P, ctypes.POINTER(_I), _P]),
    "uspace_clip_num_params": (_I, [ctypes.POINTER(ClipConfig)]),
    "uspace_clip_param_numel": (_L, [ctypes.POINTER(ClipConfig), _I]),
    "uspace_clip_weight_bytes": (_SZ, [ctypes.POINTER(ClipConfig)]),
    "uspace_clip_workspace_bytes": (_SZ, [ctypes.POINTER(ClipConfig), _I]),
    "uspace_clip_pack_weights": (_I, [ctypes.POINTER(ClipConfig), _P, _I, _P, _SZ, _P]),
    "uspace_clip_text_forward": (_I, [ctypes.POINTER(ClipConfig), _P, _P, _SZ, _P, _P, _I, _I, _I, _P]),
    "uspace_attention_causal_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "uspace_layernorm_f32": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "uspace_table_embed": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "uspace_quick_gelu_bf16": (_I, [_P, _L, _P]),
    "uspace_prof_gemm_begin": (_I, [_I, _I, _I, _I]),
    "uspace_prof_gemm_end": (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I)]),
    "uspace_prof_all_begin": (_I, [_I]),
    "uspace_prof_all_end": (_I, [ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_double), _I, ctypes.POINTER(_I)]),
    "uspace_prof_dropped": (_L, []),
    "uspace_prof_mfma_peak": (_I, [_I, ctypes.POINTER(ctypes.c_double)]),
    "uspace_prof_mfma_peak_clock": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "uspace_prof_mfma_peak_gemm_op": (_I, [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "uspace_prof_hbm_copy": (_I, [_SZ, _I, ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def lib():
    """Load libuspace_hip.so or raise -- the product path has no CPU / eager fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UspaceHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). uspace_amd has no fallback path.")
        L = ctypes.CDLL(LIB_PATH)
        if os.path.realpath(LIB_PATH) != os.path.realpath(_DEFAULT_LIB):
            import sys
            print(f"uspace_amd: USPACE_HIP_LIB overrides the product library: loading {LIB_PATH} -- a lab build, results and timings are not the "
                  "product's", file=sys.stderr)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if L.uspace_abi_version() != ABI_VERSION:
            raise UspaceHipError("libuspace_hip.so ABI version mismatch")
        # the library itself reads no environment: USPACE_LN_FOLD=0 selects the separate-LayerNorm path for A/B runs
        env = os.environ.get("USPACE_LN_FOLD")
        if env is not None and env != "":
            L.uspace_uvit_set_ln_fold(0 if env[0] == "0" else 1)
        env = os.environ.get("USPACE_GEMM_SK")   # ... USPACE_GEMM_SK=0 switches the GEMM's in-launch K-split tail off
        if env is not None and env != "":
            L.uspace_gemm_set_sk(0 if env[0] == "0" else 1)
        _lib = L
    return _lib


def library_info():
    """Which library is loaded: path, whether it is the in-tree product build, lab symbols it exports."""
    L = lib()
    lab = [s for s in ("uspace_lab_gemm_force_tile", "uspace_lab_gemm_set_big_form", "uspace_lab_gemm_trace") if hasattr(L, s)]
    return {"path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE)), "product_build": os.path.realpath(LIB_PATH) == os.path.realpath(_DEFAULT_LIB),
            "lab_symbols": lab}


def check(rc, what):
    if rc != 0:
        raise UspaceHipError(f"{what} failed: {_ERR.get(rc, rc)}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def sync_current_stream():
    """Wait for the work enqueued so far on the stream the kernels are launched on."""
    torch.cuda.current_stream().synchronize()


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def require_device(t, name="tensor"):
    if not t.is_cuda:
        raise UspaceHipError(
            f"{name} lives on {t.device}: uspace_amd runs only on a ROCm device (MI355X); there is no CPU path")


# ------------------------------------------------------------------------------------------
# thin operator wrappers (used by the parity tests and by the solver); tensors are torch CUDA
# ------------------------------------------------------------------------------------------
def gemm(A, W, *, A2=None, bias=None, resid=None, gelu=False, out_f32=None, out_bf16=None, split_ws=None, sk_ws=None):
    """acc = [A|A2] @ W^T with fused epilogue; A/A2/W are torch.bfloat16, returns (out_f32, out_bf16).  split_ws: an optional
    fp32 workspace tensor (uspace_gemm_split_ws_bytes) that allows the K-split form of small launches.  sk_ws: an optional
    workspace tensor (uspace_gemm_sk_ws_bytes) that allows the in-launch K-split tail; its 256 arrival counters are made here."""
    require_device(A, "A")
    M, K1 = A.shape
    N, K = W.shape
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
    if gelu:
        flags |= EPI_GELU
    if resid is not None:
        flags |= EPI_RESIDUAL
    if out_f32 is not None:
        flags |= EPI_OUT_F32
    if out_bf16 is not None:
        flags |= EPI_OUT_BF16
    args = (ptr(A), A.stride(0), ptr(A2), A2.stride(0) if A2 is not None else 0, K1, ptr(W), W.stride(0), M, N, K, flags,
            ptr(bias), ptr(resid), resid.stride(0) if resid is not None else 0,
            ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
            ptr(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0)
    if split_ws is not None or sk_ws is not None:
        ext = GemmExt()
        if split_ws is not None:
            require_device(split_ws, "split_ws")
            ext.split_ws = ptr(split_ws).value
            ext.split_ws_bytes = split_ws.numel() * split_ws.element_size()
        if sk_ws is not None:
            require_device(sk_ws, "sk_ws")
            counters = torch.zeros(256, dtype=torch.int32, device=sk_ws.device)
            ext.sk_ws = ptr(sk_ws).value
            ext.sk_ws_bytes = sk_ws.numel() * sk_ws.element_size()
            ext.sk_counters = ptr(counters).value
        rc = lib().uspace_gemm_bf16_ext(*args, ctypes.byref(ext), stream_ptr())
    else:
        rc = lib().uspace_gemm_bf16(*args, stream_ptr())
    check(rc, "uspace_gemm_bf16")
    return out_f32, out_bf16


def gemm_slabs(A, W, row_shift, *, bias=None, resid=None, out_f32=None, out_bf16=None, M=None):
    """acc[m] = sum_t A[m + row_shift[t]] @ W[:, t*K1:(t+1)*K1]^T.  ``A`` points at row 0 of the map (the caller's
    buffer has guard rows around it); M rows are produced."""
    require_device(A, "A")
    K1 = A.shape[1]
    N = W.shape[0]
    n = len(row_shift)
    assert W.shape[1] == n * K1 and A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16
    M = A.shape[0] if M is None else M
    flags = (EPI_BIAS if bias is not None else 0) | (EPI_RESIDUAL if resid is not None else 0) | \
            (EPI_OUT_F32 if out_f32 is not None else 0) | (EPI_OUT_BF16 if out_bf16 is not None else 0)
    shifts = (ctypes.c_int * n)(*[int(v) for v in row_shift])
    rc = lib().uspace_gemm_slabs_bf16(
        ptr(A), A.stride(0), ptr(W), W.stride(0), M, N, K1, n, shifts, flags, ptr(bias), ptr(resid),
        resid.stride(0) if resid is not None else 0, ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
        ptr(out_bf16), out_bf16.stride(0) if out_bf16 is not None else 0, stream_ptr())
    check(rc, "uspace_gemm_slabs_bf16")
    return out_f32, out_bf16


def layernorm(x, gamma, beta, eps=1e-5):
    require_device(x, "x")
    M, D = x.shape
    y = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    check(lib().uspace_layernorm_f32_bf16(ptr(x), ptr(gamma), ptr(beta), ptr(y), M, D, eps, stream_ptr()),
          "uspace_layernorm_f32_bf16")
    return y


def attention(qkv, B, L, H, key_scale=None):
    require_device(qkv, "qkv")
    out = torch.empty(B * L, H * 64, dtype=torch.bfloat16, device=qkv.device)
    check(lib().uspace_attention_bf16(ptr(qkv), ptr(key_scale), ptr(out), B, L, H, stream_ptr()),
          "uspace_attention_bf16")
    return out


def add_broadcast(x, delta, scale, x_bf16=None, row_scale=None):
    """x[b] += scale * (row_scale[b] if given) * delta, in place."""
    require_device(x, "x")
    B = x.shape[0]
    per = x.numel() // B
    assert delta.numel() == per and delta.dtype == torch.float32 and x.dtype == torch.float32
    if row_scale is not None:
        assert row_scale.numel() == B and row_scale.dtype == torch.float32 and row_scale.is_cuda
    check(lib().uspace_add_broadcast_rows(ptr(x), ptr(x_bf16), ptr(delta), float(scale), ptr(row_scale), B, per,
                                          stream_ptr()), "uspace_add_broadcast_rows")
    return x


def cast_bf16(src):
    require_device(src, "src")
    dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(lib().uspace_cast_f32_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "uspace_cast_f32_bf16")
    return dst


def ode_combine(out, y, ks, coefs):
    require_device(y, "y")
    n = len(ks)
    karr = (ctypes.c_void_p * max(n, 1))(*[k.data_ptr() for k in ks])
    carr = (ctypes.c_float * max(n, 1))(*[float(c) for c in coefs])
    check(lib().uspace_ode_combine(ptr(out), ptr(y), karr, carr, n, y.numel(), stream_ptr()), "uspace_ode_combine")
    return out


def ode_error_norm(y0, y1, ks, coefs, rtol, atol, scratch, result):
    """``result``: device float[2] = [rms, sum of squares]."""
    if result.numel() < 2:
        raise UspaceHipError("ode_error_norm: result must hold 2 floats (ABI 6)")
    n = len(ks)
    karr = (ctypes.c_void_p * n)(*[k.data_ptr() for k in ks])
    carr = (ctypes.c_float * n)(*[float(c) for c in coefs])
    check(lib().uspace_ode_error_norm(ptr(y0), ptr(y1), karr, carr, n, float(rtol), float(atol), y0.numel(),
                                      ptr(scratch), ptr(result), stream_ptr()), "uspace_ode_error_norm")
    return result


def prof_gemm_begin(epi_flags, N, K, max_launches=8192):
    check(lib().uspace_prof_gemm_begin(epi_flags, N, K, max_launches), "uspace_prof_gemm_begin")


def prof_peaks(mfma_iters=20000, copy_bytes=1 << 30, copy_reps=10):
    """(dense bf16 MFMA TFLOP/s of a v_mfma_f32_32x32x16_bf16-only loop, stream-copy GB/s read + write, sustained shader
    clock in GHz under the MFMA loop) measured on the current device."""
    tf, gb, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(lib().uspace_prof_mfma_peak_clock(mfma_iters, ctypes.byref(tf), ctypes.byref(ghz)), "uspace_prof_mfma_peak_clock")
    check(lib().uspace_prof_hbm_copy(copy_bytes, copy_reps, ctypes.byref(gb)), "uspace_prof_hbm_copy")
    return tf.value, gb.value, ghz.value


def prof_mfma_gemm_op(mfma_iters=20000):
    """(TFLOP/s, sustained shader GHz) of a v_mfma_f32_16x16x32_bf16-only loop on pseudo-random operands: the GEMM's instruction on
    data that toggles like real activations (prof_peaks() times v_mfma_f32_32x32x16_bf16 on near-constant operands)."""
    tf, ghz = ctypes.c_double(0.0), ctypes.c_double(0.0)
    check(lib().uspace_prof_mfma_peak_gemm_op(mfma_iters, ctypes.byref(tf), ctypes.byref(ghz)), "uspace_prof_mfma_peak_gemm_op")
    return tf.value, ghz.value


def prof_all_begin(max_launches=16384):
    check(lib().uspace_prof_all_begin(max_launches), "uspace_prof_all_begin")


def prof_dropped():
    """Launches the recorder could not take (it was full) since the last prof_*_begin()."""
    return int(lib().uspace_prof_dropped())


def prof_all_end(max_records=256):
    """-> list of dicts(kind, flags, M, N, K, launches, total_ms), one per distinct (kind, flags, M, N, K)."""
    keys = (ctypes.c_int * (6 * max_records))()
    ms = (ctypes.c_double * max_records)()
    n = ctypes.c_int(0)
    check(lib().uspace_prof_all_end(keys, ms, max_records, ctypes.byref(n)), "uspace_prof_all_end")
    return [dict(kind=keys[6 * i], flags=keys[6 * i + 1], M=keys[6 * i + 2], N=keys[6 * i + 3], K=keys[6 * i + 4],
                 launches=keys[6 * i + 5], total_ms=ms[i]) for i in range(n.value)]


def prof_gemm_end():
    ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
    check(lib().uspace_prof_gemm_end(ctypes.byref(ms), ctypes.byref(n)), "uspace_prof_gemm_end")
    return ms.value, n.value
