#!/usr/bin/env python3
"""Lab: ms per U-ViT forward (median of chunks) with whatever library USPACE_HIP_LIB names -- run alternately
with two builds in one gpurun call to price a kernel change inside the model (boxes differ by +-3 %, one box by ~0.3 %)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import COMMON, MODELS
from uspace_amd.tools.utils_uvit import get_nnet

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="L_u"); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--chunks", type=int, default=9); ap.add_argument("--per", type=int, default=20)
a = ap.parse_args()
cfg = dict(MODELS[a.model]); name = cfg.pop("name")
torch.manual_seed(1234)
net = get_nnet(name, **COMMON, **cfg).cuda().eval()
g = torch.Generator().manual_seed(7)
x = torch.randn(a.batch, 4, 32, 32, generator=g).cuda()
ctx = torch.randn(a.batch, 77, 768, generator=g).cuda() if name == "uvit_t2i" else None
t = torch.tensor(0.35, device="cuda").expand(a.batch)
def fwd(): return net(x, t, context=ctx) if ctx is not None else net(x, t, None, edit_loc=None)
for _ in range(10): fwd()
torch.cuda.synchronize()
ts = []
for _ in range(a.chunks):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.per): out, _ = fwd()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / a.per)
ts.sort()
print(f"{os.environ.get('USPACE_HIP_LIB', 'product'):40s} forward ms: median {ts[len(ts)//2]:.4f} min {ts[0]:.4f} max {ts[-1]:.4f}  |out| {float(out.abs().mean()):.6f}", flush=True)
