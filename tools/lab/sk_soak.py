#!/usr/bin/env python3
"""Soak of the GEMM's in-launch K-split tail inside the model (run on the GPU box):  python tools/lab/sk_soak.py [forwards=600]
U-ViT-L at 32, 16 and (T2I) 64 per GPU: every fc2 launch exchanges partial sums between 2 / 4 / 3 workgroups per shared tile.  One forward is
evaluated once, then `forwards` more times with other inputs in between, alternating two batch sizes (so the exchange workspace and the
counters are reused by launches of different geometry): every repeat must be BIT-IDENTICAL to the first.  A stale slab line (a missing
release / acquire somewhere) at a rate of 1e-6 per exchange would show: 600 forwards x 21 launches x 128 tiles x 2 parts = 3.2 M exchanges."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    g = torch.Generator().manual_seed(5)
    for key, batches in (("L_u", (32, 16)), ("L_t", (64, 16))):
        cfg = dict(MODELS[key])
        name = cfg.pop("name")
        torch.manual_seed(1234)
        net = get_nnet(name, **COMMON, **cfg).cuda().eval()
        t2i = name == "uvit_t2i"
        data = {}
        for B in batches:
            z = torch.randn(B, 4, 32, 32, generator=g).cuda()
            z2 = torch.randn(B, 4, 32, 32, generator=g).cuda()
            ctx = torch.randn(B, 77, 768, generator=g).cuda() if t2i else None
            data[B] = (z, z2, ctx)

        def fwd(B, x, tv):
            t = torch.tensor(float(tv), device="cuda").expand(B)
            with torch.no_grad():
                return (net(x, t, context=data[B][2]) if t2i else net(x, t, None, edit_loc=None))[0]
        first = {B: fwd(B, data[B][0], 0.35).clone() for B in batches}
        bad = 0
        t0 = time.time()
        for i in range(n):
            B = batches[i % len(batches)]
            fwd(B, data[B][1], 0.05 + 0.9 * (i % 17) / 17)          # other input, other time
            again = fwd(B, data[B][0], 0.35)
            if not torch.equal(again, first[B]):
                bad += 1
                print(f"{key} B={B} repeat {i}: differs, max abs {float((again - first[B]).abs().max()):.3e}", flush=True)
        torch.cuda.synchronize()
        print(f"{key} batches {batches}: {n} repeats ({2 * n} forwards) in {time.time() - t0:.1f} s, {bad} differing", flush=True)
        del net
        torch.cuda.empty_cache()
        if bad:
            raise SystemExit(1)


if __name__ == "__main__":
    main()
