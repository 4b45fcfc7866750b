#!/usr/bin/env python3
"""Kernel micro-benchmarks at the BASELINE shapes (run on the GPU box): GEMMs, attention, LayerNorm."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uspace_amd import _hip  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--L", type=int, default=257)
    ap.add_argument("--D", type=int, default=1024)
    a = ap.parse_args()
    M, D = a.B * a.L, a.D
    dev = "cuda"
    bf = torch.bfloat16
    x = torch.randn(M, D, device=dev).to(bf)
    x2 = torch.randn(M, D, device=dev).to(bf)
    f = torch.randn(M, 4 * D, device=dev).to(bf)
    res = torch.randn(M, D, device=dev)
    shapes = [("qkv", D, 3 * D), ("proj", D, D), ("fc1", D, 4 * D), ("fc2", 4 * D, D), ("skip", 2 * D, D)]
    tot_t, tot_f = 0.0, 0.0
    for name, K, N in shapes:
        W = (torch.randn(N, K, device=dev) * 0.02).to(bf)
        b = torch.zeros(N, device=dev)
        if name == "qkv":
            o = torch.empty(M, N, device=dev, dtype=bf)
            fn = lambda: _hip.gemm(x, W, out_bf16=o)
        elif name == "fc1":
            o = torch.empty(M, N, device=dev, dtype=bf)
            fn = lambda: _hip.gemm(x, W, bias=b, gelu=True, out_bf16=o)
        elif name == "fc2":
            o = torch.empty(M, N, device=dev, dtype=bf)
            fn = lambda: _hip.gemm(f, W, bias=b, resid=res, out_f32=res, out_bf16=o)
        elif name == "proj":
            fn = lambda: _hip.gemm(x, W, bias=b, resid=res, out_f32=res)
        else:
            fn = lambda: _hip.gemm(x, W, A2=x2, bias=b, out_f32=res)
        t = timeit(fn)
        fl = 2.0 * M * N * K
        cnt = {"qkv": 21, "proj": 21, "fc1": 21, "fc2": 21, "skip": 10}[name]
        tot_t += t * cnt
        tot_f += fl * cnt
        print(f"gemm {name:5s} M={M} N={N} K={K}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
    print(f"  -> all GEMMs of one forward: {tot_t*1e3:.2f} ms, {tot_f/tot_t/1e12:.1f} TFLOP/s aggregate")
    # ablation: the fc1 shape without the GELU in its epilogue (bias + bf16 store only)
    W = (torch.randn(4 * D, D, device=dev) * 0.02).to(bf)
    b = torch.zeros(4 * D, device=dev)
    o = torch.empty(M, 4 * D, device=dev, dtype=bf)
    t = timeit(lambda: _hip.gemm(x, W, bias=b, out_bf16=o))
    print(f"gemm fc1 without GELU (ablation):        {t*1e6:9.1f} us  {2.0*M*4*D*D/t/1e12:7.1f} TFLOP/s")
    H = D // 64
    qkv = torch.randn(M, 3 * D, device=dev).to(bf)
    t = timeit(lambda: _hip.attention(qkv, a.B, a.L, H))
    fl = 4.0 * a.L * a.L * D * a.B
    byts = M * 3 * D * 2 + M * D * 2
    print(f"attention B={a.B} L={a.L} H={H}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TFLOP/s  {byts/t/1e9:7.0f} GB/s")
    xs = torch.randn(M, D, device=dev)
    g, bb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    t = timeit(lambda: _hip.layernorm(xs, g, bb))
    print(f"layernorm M={M} D={D}: {t*1e6:9.1f} us  {M*D*6/t/1e9:7.0f} GB/s (algorithmic 6 B/elem)")


if __name__ == "__main__":
    main()
