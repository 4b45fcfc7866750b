"""Side-by-side timing of this repo's GEMM kernel and the vendor library (hipBLASLt through torch.nn.functional.linear)
on the U-ViT-L shapes and on long-K shapes (where prologue/epilogue cost is amortised).  Measurement aid only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from uspace_amd import _hip  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("qkv", 16448, 3072, 1024), ("proj", 16448, 1024, 1024), ("fc1", 16448, 4096, 1024), ("fc2", 16448, 1024, 4096),
          ("skip", 16448, 1024, 2048), ("4k^3", 4096, 4096, 4096), ("8k^3", 8192, 8192, 8192), ("fc1,K=8k", 16384, 4096, 8192)]
for n, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    o = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tv = timeit(lambda: torch.nn.functional.linear(a, w))
    to = timeit(lambda: _hip.gemm(a, w, out_bf16=o))
    ref = torch.nn.functional.linear(a, w).float()
    err = float((o.float() - ref).norm() / ref.norm())
    fl = 2.0 * M * N * K / 1e6
    print(f"{n:9s} M={M} N={N} K={K}: vendor {tv:8.1f} us {fl/tv:7.1f} TFLOP/s | ours (plain bf16 store) {to:8.1f} us {fl/to:7.1f} TFLOP/s | rel diff {err:.1e}")
