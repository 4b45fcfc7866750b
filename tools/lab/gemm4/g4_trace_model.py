#!/usr/bin/env python3
"""Phase stamps of the four-wave GEMM form INSIDE a U-ViT forward (lab build, see g4_trace.py):
    USPACE_HIP_LIB=tools/lab/_build/lib_gemm4_trace.so python3 tools/lab/gemm4/g4_trace_model.py
For each model GEMM shape: the stamps of its last launch of a forward."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from bench import COMMON, MODELS  # noqa: E402
from uspace_amd import _hip  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402


def main():
    lib = _hip.lib()
    lib.uspace_lab_gemm_set_big_form(0)      # the four-wave form wherever it applies
    lib.uspace_lab_gemm4_trace.restype = ctypes.c_int
    lib.uspace_lab_gemm4_trace.argtypes = [ctypes.c_void_p]
    lib.uspace_lab_gemm4_trace_filter.restype = None
    cfg = dict(MODELS["L_u"])
    name = cfg.pop("name")
    torch.manual_seed(1234)
    net = get_nnet(name, **COMMON, **cfg).cuda().eval()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 4, 32, 32, generator=g).cuda()
    t = torch.tensor(0.35, device="cuda").expand(B)
    M = B * 257
    for nm, N, K in (("qkv", 3072, 1024), ("fc1", 4096, 1024), ("proj", 1024, 1024), ("fc2", 1024, 4096), ("skip", 1024, 2048)):
        lib.uspace_lab_gemm4_trace_filter(N, K)
        with torch.no_grad():
            for _ in range(3):
                net(x, t, None, edit_loc=None)
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
        nb = lib.uspace_lab_gemm4_trace(buf.ctypes.data_as(ctypes.c_void_p))
        nblk = min(nb, (M // 256) * (N // 256))
        tt = buf.reshape(1024, 4, 8)[:nblk].astype(np.int64)
        nk = K // 64
        med = lambda v: float(np.median(v))
        d = [tt[:, :, i + 1] - tt[:, :, i] for i in range(5)]
        span = (tt[:, :, 5].max() - tt[:, :, 0].min())
        print(f"{nm:5s} N={N} K={K} | prologue {med(d[0]):6.0f} | K loop {med(d[1]):7.0f} = {med(d[1]) / nk:5.0f}/tile (p90 {np.percentile(d[1], 90) / nk:5.0f}) | to rows {med(d[2]):5.0f} | rows {med(d[3]):6.0f} "
              f"(p90 {np.percentile(d[3], 90):6.0f}) | tail {med(d[4]):5.0f} | total {med(tt[:, :, 5] - tt[:, :, 0]):7.0f} | per K tile: counter waits {med(tt[:, :, 6]) / nk:5.0f}, barrier waits {med(tt[:, :, 7]) / nk:5.0f}", flush=True)


if __name__ == "__main__":
    main()
