#!/usr/bin/env python3
"""Phase stamps of the four-wave GEMM form (lab build of gemm4.hip with a TRACE=1 K loop):
    python3 tools/lab/gemm4/gen_kloop4.py --out=tools/lab/_build/var/kloop4_trace.inc TRACE=1
    tools/lab/build_variant.sh g4trace gemm4.hip '-DKLOOP4_INC="\"'$PWD'/tools/lab/_build/var/kloop4_trace.inc\""'
    USPACE_HIP_LIB=tools/lab/_build/lib_gemm4_trace.so python3 tools/lab/gemm4/g4_trace.py
Prints, per model GEMM, the median cycles (s_memtime) of: prologue | K loop (per K tile) | loop exit -> first epilogue row | the eight
rows | tail (strip, partial sums, store drain)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from uspace_amd import _hip  # noqa: E402

B, G, R, F, H, C, L, K1F = 1, 2, 4, 8, 16, 32, 64, 128


def main():
    lib = _hip.lib()
    lib.uspace_lab_gemm_set_big_form(0)      # the four-wave form wherever it applies
    tr_fn = lib.uspace_lab_gemm4_trace
    tr_fn.restype = ctypes.c_int
    tr_fn.argtypes = [ctypes.c_void_p]
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16448
    if len(sys.argv) > 2:      # residual prefetch under the K loop: 1 / 0
        lib.uspace_lab_gemm4_prefetch.restype = None
        lib.uspace_lab_gemm4_prefetch(int(sys.argv[2]))
    D = 1024
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, 4 * D, generator=g).to(dev).to(torch.bfloat16)
    A2 = torch.randn(M, D, generator=g).to(dev).to(torch.bfloat16)
    W = (torch.randn(4 * D, 4 * D, generator=g) * 0.02).to(dev).to(torch.bfloat16)
    bias = torch.randn(4 * D, generator=g).to(dev) * 0.1
    Rm = torch.randn(M, D, generator=g).to(dev)
    part_in = torch.stack([torch.randn(M, 4, generator=g) * 0.5, 128 + 10 * torch.randn(M, 4, generator=g)], dim=2).contiguous().to(dev)
    cvec = (torch.randn(M, generator=g) * 0.01).to(dev)
    shapes = [("qkv  L|B|H", 3 * D, D, L | B | H), ("proj C|B|R|F", D, D, C | B | R | F), ("fc1  L|B|G|H", 4 * D, D, L | B | G | H),
              ("fc2  C|B|R|F", D, 4 * D, C | B | R | F), ("fc2  B|R|F|H", D, 4 * D, B | R | F | H), ("skip K|C|B|F", D, 2 * D, K1F | C | B | F),
              ("plain B|H", 4 * D, D, B | H)]
    for name, N, K, fl in shapes:
        skip = K == 2 * D and N == D
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        o32 = torch.empty(M, D, device=dev)
        cen = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        pout = torch.empty(M, 8, 2, device=dev)
        cout = torch.empty(M, device=dev)
        ext = _hip.GemmExt()
        ext.norm_dim, ext.eps = D, 1e-5
        if fl & C:
            ext.row_c, ext.out_cen, ext.ld_cen, ext.part_out = _hip.ptr(cvec).value, _hip.ptr(cen).value, D, _hip.ptr(pout).value
        if fl & L:
            ext.part_in, ext.np_in, ext.colsum, ext.row_c, ext.c_out = _hip.ptr(part_in).value, 4, _hip.ptr(bias).value, _hip.ptr(cvec).value, _hip.ptr(cout).value
        if fl & K1F:
            ext.row_add, ext.col_add = _hip.ptr(cvec).value, _hip.ptr(bias).value

        def run():
            rc = lib.uspace_gemm_bf16_ext(_hip.ptr(A), D if skip else K, _hip.ptr(A2) if skip else None, D if skip else 0, D if skip else K, _hip.ptr(W), K, M, N, K, fl,
                                          _hip.ptr(bias), _hip.ptr(Rm) if fl & R else None, D, _hip.ptr(o32) if fl & F else None, D,
                                          _hip.ptr(o16) if fl & H else None, N, ctypes.byref(ext), _hip.stream_ptr())
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        if os.environ.get("G4_COLD"):       # operands out of L2 / Infinity Cache: 1 GiB written between launches; the stamps are the last launch's
            junk = torch.empty(256 << 20, device=dev)
            touch = os.environ.get("G4_COLD")       # "1": nothing touched; contains "w" / "a": that operand is read once in front of the launch
            for _ in range(3):
                junk.fill_(1.0)
                if "w" in touch:
                    W[:N, :K].view(torch.int16).max()
                if "a" in touch:
                    (A2 if skip else A)[:, :K].view(torch.int16).max()
                    if skip:
                        A[:, :D].view(torch.int16).max()
                run()
            torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if not os.environ.get("G4_COLD"):
            ev0.record()
            for _ in range(10):
                run()
            ev1.record()
            torch.cuda.synchronize()
        else:
            ev0.record(); ev1.record(); torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 100
        buf = np.zeros(1024 * 4 * 8, dtype=np.uint64)
        nb = tr_fn(buf.ctypes.data_as(ctypes.c_void_p))
        nblk = min(nb, (M // 256) * (N // 256))
        t = buf.reshape(1024, 4, 8)[:nblk].astype(np.int64)
        nk = K // 64
        med = lambda x: float(np.median(x))
        d = [t[:, :, i + 1] - t[:, :, i] for i in range(5)]
        print(f"{name:14s} {us:7.1f} us | prologue {med(d[0]):6.0f} | K loop {med(d[1]):7.0f} = {med(d[1]) / nk:5.0f}/tile | to rows {med(d[2]):5.0f} | rows {med(d[3]):6.0f} "
              f"(p90 {np.percentile(d[3], 90):6.0f}) | tail {med(d[4]):5.0f} | total {med(t[:, :, 5] - t[:, :, 0]):7.0f} | per K tile: counter waits {med(t[:, :, 6]) / nk:5.0f}, barrier waits {med(t[:, :, 7]) / nk:5.0f}", flush=True)


if __name__ == "__main__":
    main()
