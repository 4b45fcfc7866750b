#!/usr/bin/env python3
"""Lab: uspace_embed_tokens / uspace_center_rows timings by batch and width (is the launch throughput- or latency-bound?).
Round 5 ran it on a build that also had a fused entry (token rows + the first norm's centring pass in one launch): profiles/r05_embed_lab.md."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uspace_amd import _hip as hip
lib = hip.lib()
S, p, C = 32, 2, 4
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for D in (1024, 512):
    for B in (8, 16, 32, 64, 128, 256):
        L = 257
        img = torch.randn(B, C, S, S, device="cuda"); t = torch.rand(B, device="cuda")
        pw = torch.randn(D, 16, device="cuda") * 0.2; pb = torch.randn(D, device="cuda") * 0.1; pos = torch.randn(L, D, device="cuda") * 0.02
        tok = torch.empty(B, L, D, device="cuda"); xc = torch.empty(B * L, D, device="cuda", dtype=torch.bfloat16)
        c = torch.empty(B * L, device="cuda"); part = torch.empty(B * L, 2, device="cuda")
        st = hip.stream_ptr()
        a = run(lambda: lib.uspace_embed_tokens(hip.ptr(img), hip.ptr(t), 1, None, 0, 0, hip.ptr(pw), hip.ptr(pb), hip.ptr(pos), hip.ptr(tok), None, B, C, S, p, D, st))
        b = run(lambda: lib.uspace_center_rows(hip.ptr(tok), hip.ptr(xc), hip.ptr(c), hip.ptr(part), B * L, D, st))
        mb = B * L * D * 4 / 1e6
        print(f"D {D:5d} B {B:4d}: embed {a:7.1f} us ({mb / a:5.2f} TB/s of rows)  center {b:7.1f} us", flush=True)
