#!/usr/bin/env python3
"""In-model A/B of the two GEMM forms (lab library: tools/lab/gemm4/build.sh, run with USPACE_HIP_LIB=tools/lab/_build/lib_gemm4.so): U-ViT forwards at a BASELINE shape, the forms alternating round by
round inside ONE process (boxes differ by more than the forms do), every GEMM / attention launch timed with the library's recorder.

    python3 tools/lab/gemm4/forward_ab.py [--model L_u] [--batch 64] [--rounds 4] [--fwd 6] [--forms 1,0]
"""
import argparse
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd import _hip  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402

EPI = {1: "B", 2: "G", 4: "R", 8: "F", 16: "H", 32: "C", 64: "L", 128: "K"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="L_u")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--fwd", type=int, default=6)
    ap.add_argument("--forms", default="1,0")
    a = ap.parse_args()
    forms = [int(f) for f in a.forms.split(",")]
    cfg = dict(MODELS[a.model])
    name = cfg.pop("name")
    torch.manual_seed(1234)
    net = get_nnet(name, **COMMON, **cfg).cuda().eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(a.batch, 4, 32, 32, generator=g).cuda()
    ctx = torch.randn(a.batch, 77, 768, generator=g).cuda() if name == "uvit_t2i" else None
    t = torch.tensor(0.35, device="cuda").expand(a.batch)
    lib = _hip.lib()

    def fwd():
        with torch.no_grad():
            return net(x, t, context=ctx)[0] if ctx is not None else net(x, t, None, edit_loc=None)[0]
    outs = {}
    for f in forms:
        lib.uspace_lab_gemm_set_big_form(f)
        for _ in range(2):
            outs[f] = fwd()
    torch.cuda.synchronize()
    per = {f: defaultdict(lambda: [0, 0.0]) for f in forms}
    wall = {f: [] for f in forms}
    for r in range(a.rounds):
        for f in forms:
            lib.uspace_lab_gemm_set_big_form(f)
            fwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            _hip.prof_all_begin(a.fwd * 200)
            e0.record()
            for _ in range(a.fwd):
                fwd()
            e1.record()
            torch.cuda.synchronize()
            for rec in _hip.prof_all_end():
                k = (rec["kind"], rec["flags"], rec["M"], rec["N"], rec["K"])
                per[f][k][0] += rec["launches"]
                per[f][k][1] += rec["total_ms"]
            wall[f].append(e0.elapsed_time(e1) / a.fwd)
    lib.uspace_lab_gemm_set_big_form(0)
    keys = sorted(per[forms[0]].keys(), key=lambda k: -per[forms[0]][k][1])
    print(f"{a.model} batch {a.batch}: forms {forms} (1 = 8-wave template, 0 = four-wave form where it applies)")
    tot = {f: 0.0 for f in forms}
    for k in keys:
        kind, fl, M, N, K = k
        nm = "attention" if kind == 1 else "gemm " + "|".join(v for b, v in EPI.items() if fl & b)
        cells = []
        for f in forms:
            n, ms = per[f].get(k, [0, 0.0])
            cells.append(f"{ms / max(n, 1) * 1e3:8.1f} us x{n // (a.rounds * a.fwd):3d}")
            tot[f] += ms / (a.rounds * a.fwd)
        ratio = (per[forms[-1]][k][1] / max(per[forms[-1]][k][0], 1)) / (per[forms[0]][k][1] / max(per[forms[0]][k][0], 1))
        print(f"  {nm:22s} M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(cells) + f" | last/first {ratio:.3f}")
    print("  recorded kernels per forward: " + " | ".join(f"form {f}: {tot[f]:.3f} ms" for f in forms))
    print("  wall per forward (median):    " + " | ".join(f"form {f}: {sorted(wall[f])[len(wall[f]) // 2]:.3f} ms" for f in forms))
    d = (outs[forms[0]] - outs[forms[-1]]).abs().max().item() if len(forms) > 1 else 0.0
    print(f"  max |out(form {forms[0]}) - out(form {forms[-1]})| = {d:.3e} (outputs ~ {outs[forms[0]].abs().mean().item():.3f})")


if __name__ == "__main__":
    main()
