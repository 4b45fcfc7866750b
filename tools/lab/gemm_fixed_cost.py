"""What a launch of 256x256 tiles costs besides its K loop: time against K (64 ... 4096) at fixed M x N, bias + bf16 output
(the qkv / fc1 store pattern), for one round (N = 1024), three (3072) and four (4096) rounds of tiles at 64 x 257 rows.
Intercept = launch + first-stage latency + last epilogue + drain; slope = one K tile per round.
    python tools/lab/gemm_fixed_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from uspace_amd import _hip  # noqa: E402


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return float(np.median(ts))


def main():
    dev = torch.device("cuda:0")
    M = 64 * 257
    for N in (1024, 3072, 4096):
        rows = []
        for K in (64, 128, 256, 512, 1024, 2048, 4096):
            A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            bias = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us = timed(lambda: _hip.gemm(A, W, bias=bias, out_bf16=out))
            rows.append((K, us))
        ks = np.array([r[0] / 64 for r in rows]); us = np.array([r[1] for r in rows])
        slope, icpt = np.polyfit(ks[3:], us[3:], 1)
        print(f"N={N} ({N // 256 * 64 // 256} rounds of 256 tiles): " + "  ".join(f"K={k}: {u:.1f}" for k, u in rows)
              + f" us | fit over K >= 512: {icpt:.1f} us + {slope:.2f} us per K tile", flush=True)


if __name__ == "__main__":
    main()
