"""Determinism stress for the small-batch path (64x64 ring tiles with counted waits, k-split head, query-split attention): the same
U-ViT-S / U-ViT-L evaluation repeated many times, eager and through the hipGraph, must give bit-identical outputs every time.
    python tools/lab/stress_small_batch.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    bad = 0
    for name, B in (("S_u", 4), ("S_u", 1), ("S_u", 8), ("L_u", 2), ("L_u", 1)):
        torch.manual_seed(3)
        cfg = dict(MODELS[name])
        net = get_nnet(cfg.pop("name"), **COMMON, **cfg).to(dev).eval()
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, 4, 32, 32, generator=g).to(dev)
        t = torch.full((), 0.37, device=dev).expand(B)
        for use_graph in (False, True):
            net.use_graph = use_graph
            ref = net(x, t)[0].clone()
            torch.cuda.synchronize()
            diff = 0
            for i in range(reps):
                out = net(x, t)[0]
                if not torch.equal(out, ref):
                    diff += 1
            torch.cuda.synchronize()
            bad += diff
            print(f"{name} B={B} graph={use_graph}: {reps} evaluations, {diff} differ from the first; finite={bool(torch.isfinite(ref).all())}", flush=True)
    print("STRESS", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
