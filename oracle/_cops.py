"""ctypes loader for oracle/ops.c (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_lib = None


def build(force=False):
    """Compile ops.c into oracle/_build/ (gcc + make; works on CPU-only hosts)."""
    need = force or not all(
        os.path.exists(os.path.join(_BUILD, n)) for n in ("liboracle_v3.so", "liboracle_generic.so"))
    if need:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))


def _cpu_has_v3():
    try:
        flags = ""
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line + " "
                    break
        need = ("avx2", "fma", "bmi2", "movbe", "f16c")
        return all((" " + n + " ") in flags.replace("\n", " ") for n in need)
    except OSError:
        return False


def lib():
    global _lib
    if _lib is not None:
        return _lib
    name = "liboracle_v3.so" if _cpu_has_v3() else "liboracle_generic.so"
    path = os.path.join(_BUILD, name)
    if not os.path.exists(path):
        build()
    L = ctypes.CDLL(path)
    f32p = ctypes.POINTER(ctypes.c_float)
    lg = ctypes.c_long
    L.oracle_abi_version.restype = ctypes.c_int
    L.oracle_num_threads.restype = ctypes.c_int
    L.oracle_set_threads.argtypes = [ctypes.c_int]
    L.oracle_linear.argtypes = [f32p, f32p, f32p, f32p, lg, lg, lg, ctypes.c_int]
    L.oracle_layernorm.argtypes = [f32p, f32p, f32p, f32p, lg, lg, ctypes.c_float]
    L.oracle_gelu.argtypes = [f32p, lg]
    L.oracle_attention.argtypes = [f32p, f32p, f32p, lg, lg, lg, lg, ctypes.c_float]
    L.oracle_patch_embed.argtypes = [f32p, f32p, f32p, f32p, lg, lg, lg, lg, lg]
    L.oracle_timestep_embedding.argtypes = [f32p, f32p, lg, lg]
    L.oracle_unpatchify.argtypes = [f32p, f32p, lg, lg, lg, lg]
    L.oracle_conv3x3.argtypes = [f32p, f32p, f32p, f32p, lg, lg, lg]
    L.oracle_conv2d.argtypes = [f32p, f32p, f32p, f32p, lg, lg, lg, lg, lg, lg]
    L.oracle_groupnorm.argtypes = [f32p, f32p, f32p, f32p, lg, lg, lg, lg, ctypes.c_float]
    L.oracle_conv2d.restype = None
    L.oracle_groupnorm.restype = None
    for n in ("oracle_linear", "oracle_layernorm", "oracle_gelu", "oracle_attention", "oracle_patch_embed",
              "oracle_timestep_embedding", "oracle_unpatchify", "oracle_conv3x3", "oracle_set_threads"):
        getattr(L, n).restype = None
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def linear(x, W, b=None, out=None, accumulate=False):
    x = _f32(x)
    W = _f32(W)
    b = _f32(b) if b is not None else None
    lead = x.shape[:-1]
    K = x.shape[-1]
    N = W.shape[0]
    assert W.shape[1] == K
    M = int(np.prod(lead)) if lead else 1
    if out is None:
        assert not accumulate
        out = np.empty(lead + (N,), np.float32)
    lib().oracle_linear(_p(x), _p(W), _p(b), _p(out), M, N, K, 1 if accumulate else 0)
    return out


def layernorm(x, g, b, eps=1e-5):
    x = _f32(x)
    out = np.empty_like(x)
    D = x.shape[-1]
    lib().oracle_layernorm(_p(x), _p(_f32(g)), _p(_f32(b)), _p(out), x.size // D, D, eps)
    return out


def gelu(x):
    x = _f32(x).copy()
    lib().oracle_gelu(_p(x), x.size)
    return x


def attention(qkv, H, colscale=None):
    qkv = _f32(qkv)
    B, L, C3 = qkv.shape
    C = C3 // 3
    Dh = C // H
    out = np.empty((B, L, C), np.float32)
    cs = _f32(colscale) if colscale is not None else None
    lib().oracle_attention(_p(qkv), _p(cs), _p(out), B, L, H, Dh, float(Dh) ** -0.5)
    return out


def patch_embed(img, w, b):
    img = _f32(img)
    B, C, HW, _ = img.shape
    D, _, p, _ = w.shape
    out = np.empty((B, (HW // p) ** 2, D), np.float32)
    lib().oracle_patch_embed(_p(img), _p(_f32(w)), _p(_f32(b)), _p(out), B, C, HW, p, D)
    return out


def timestep_embedding(t, D):
    t = _f32(t)
    out = np.empty((t.shape[0], D), np.float32)
    lib().oracle_timestep_embedding(_p(t), _p(out), t.shape[0], D)
    return out


def unpatchify(tok, C):
    tok = _f32(tok)
    B, T, PD = tok.shape
    g = int(round(T ** 0.5))
    p = int(round((PD // C) ** 0.5))
    assert g * g == T and p * p * C == PD
    out = np.empty((B, C, g * p, g * p), np.float32)
    lib().oracle_unpatchify(_p(tok), _p(out), B, g, p, C)
    return out


def conv3x3(x, w, b):
    x = _f32(x)
    B, C, HW, _ = x.shape
    out = np.empty_like(x)
    lib().oracle_conv3x3(_p(x), _p(_f32(w)), _p(_f32(b)), _p(out), B, C, HW)
    return out


def conv2d(x, w, b):
    x = _f32(x)
    w = _f32(w)
    B, Ci, H, W = x.shape
    Co, _, k, _ = w.shape
    out = np.empty((B, Co, H, W), np.float32)
    lib().oracle_conv2d(_p(x), _p(w), _p(_f32(b)) if b is not None else None, _p(out), B, Ci, Co, H, W, k)
    return out


def groupnorm(x, g, b, groups=32, eps=1e-6):
    x = _f32(x)
    B, C = x.shape[:2]
    out = np.empty_like(x)
    lib().oracle_groupnorm(_p(x), _p(_f32(g)), _p(_f32(b)), _p(out), B, C, x.size // (B * C), groups, eps)
    return out
