#!/usr/bin/env python3
"""Warm against cold operands, both GEMM forms (round 5; lab library: USPACE_HIP_LIB=tools/lab/_build/lib_gemm4.so): the model GEMMs timed one launch at a time with HIP events, (a) back to back on
operands that stay in L2 / the Infinity Cache, (b) with 1 GiB written between launches (what a forward does to the caches).
    python3 tools/lab/gemm4/cold_ab.py [M]"""
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 16448
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
    junk = torch.empty(256 << 20, device=dev)
    shapes = [("qkv  L|B|H", 3 * D, D, L | B | H), ("fc1  L|B|G|H", 4 * D, D, L | B | G | H), ("proj C|B|R|F", D, D, C | B | R | F),
              ("fc2  C|B|R|F", D, 4 * D, C | B | R | F), ("skip K|C|B|F", D, 2 * D, K1F | C | B | F)]
    print(f"M = {M}; us per launch, median of 9: form 1 = 8-wave template, form 0 = four-wave form")
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
        res = {}
        for cold in (0, 1):
            for form in (1, 0):
                lib.uspace_lab_gemm_set_big_form(form)
                ts = []
                for it in range(11):
                    if cold:
                        junk.fill_(float(it))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    run()
                    e1.record()
                    torch.cuda.synchronize()
                    if it >= 2:
                        ts.append(e0.elapsed_time(e1) * 1e3)
                res[(cold, form)] = float(np.median(ts))
        lib.uspace_lab_gemm_set_big_form(0)
        print(f"  {name:14s} warm: form 1 {res[(0, 1)]:7.1f} | form 0 {res[(0, 0)]:7.1f} ({res[(0, 0)] / res[(0, 1)]:.3f}) || cold: form 1 {res[(1, 1)]:7.1f} | form 0 {res[(1, 0)]:7.1f} "
              f"({res[(1, 0)] / res[(1, 1)]:.3f}) || cold / warm: form 1 {res[(1, 1)] / res[(0, 1)]:.3f}, form 0 {res[(1, 0)] / res[(0, 0)]:.3f}", flush=True)


if __name__ == "__main__":
    main()
