#!/usr/bin/env python3
"""A few U-ViT forwards at a BASELINE shape -- the workload rocprofv3 counter passes are run over."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="L_u")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
cfg = dict(MODELS[a.model])
name = cfg.pop("name")
torch.manual_seed(1234)
net = get_nnet(name, **COMMON, **cfg).cuda().eval()
g = torch.Generator().manual_seed(7)
x = torch.randn(a.batch, 4, 32, 32, generator=g).cuda()
ctx = torch.randn(a.batch, 77, 768, generator=g).cuda() if name == "uvit_t2i" else None
t = torch.tensor(0.35, device="cuda").expand(a.batch)
for _ in range(a.reps):
    out, _ = net(x, t, context=ctx) if ctx is not None else net(x, t, None, edit_loc=None)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
