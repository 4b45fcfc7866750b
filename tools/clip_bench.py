"""Throughput of the CLIP-L text transformer (77 tokens -> [77, 768] context) on the HIP path.
    python tools/clip_bench.py [--batch 64]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from uspace_amd.libs.clip import CLIPTextTransformer, CLIP_L_TEXT
    torch.manual_seed(0)
    m = CLIPTextTransformer(**CLIP_L_TEXT).cuda()
    ids = torch.randint(0, 49408, (a.batch, 77), device="cuda")
    m(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        m(ids)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    c = CLIP_L_TEXT
    D, F, Ln, L = c["hidden_size"], c["intermediate_size"], c["num_hidden_layers"], 77
    flops = Ln * (2 * L * D * (4 * D + 2 * F) + 4 * L * L * D)          # per prompt (full L x L attention counted)
    print(json.dumps({"workload": "CLIP-L text transformer, 77 tokens", "batch": a.batch, "ms_per_batch": dt * 1e3,
                      "prompts_per_s": a.batch / dt, "gflop_per_prompt": flops / 1e9, "tflops": flops * a.batch / dt / 1e12}))


if __name__ == "__main__":
    main()
