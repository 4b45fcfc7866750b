#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configurations on ONE MI355X (8-GPU configs: one GPU's share)."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402

CONFIGS = [
    dict(id=1, model="S_u", B=4, solver="euler", n=20, note="configs[0]: U-ViT-S-deep16, 20 Euler steps, batch 4"),
    dict(id=2, model="L_u", B=64, solver="dopri5", n=50, note="configs[1]: U-ViT-L, 50 dopri5 steps, batch 64"),
    dict(id=2, model="L_u", B=64, solver="euler", n=50, note="configs[1] with Euler-50"),
    dict(id=3, model="L_t", B=64, solver="euler", n=50, note="configs[2]: U-ViT-L T2I (77 ctx tokens), 50 steps, batch 64"),
    dict(id=4, model="S_t", B=64, solver="euler", n=50, note="configs[3]: U-ViT-S-deep16 T2I, 50 steps, 512/8 = 64 per GPU"),
    dict(id=5, model="L_u", B=32, solver="euler", n=50, hook=True, note="configs[4]: U-ViT-L mid-block u-space edit, 50 steps, 256/8 = 32 per GPU"),
    dict(id=2, model="L_u", B=64, solver="adaptive", n=0,
         note="configs[1] with the reference's default solver: adaptive dopri5, rtol = atol = 1e-5 (flow_matching.py:71-73); NFE is measured"),
]


def run(c, reps=2):
    cfg = dict(MODELS[c["model"]])
    name = cfg.pop("name")
    t2i = name == "uvit_t2i"
    torch.manual_seed(1234)
    net = get_nnet(name, **COMMON, **cfg).cuda().eval()
    if t2i:
        from uspace_amd.flow_matching_t2i import CNF
    else:
        from uspace_amd.flow_matching import CNF
    cnf = CNF(net)
    g = torch.Generator().manual_seed(7)
    B = c["B"]
    z = torch.randn(B, 4, 32, 32, generator=g).cuda()
    cond = torch.randn(B, 77, 768, generator=g).cuda() if t2i else None
    if c["solver"] == "adaptive":
        sk = dict(solver="adaptive", solver_fix="euler", solver_fix_step=0.02, solver_adaptive="dopri5", solver_adaptive_prec=1e-5)
        kw = dict(edit_loc=None, solver_kwargs=sk)          # not a dissection run -> dopri5 at 1e-5, as the reference
    else:
        sk = dict(solver="adaptive" if c["solver"] == "dopri5" else "fixed", solver_fix="euler", solver_fix_step=1.0 / c["n"],
                  solver_adaptive="dopri5", solver_adaptive_prec=0.01, n_steps=c["n"])
        kw = dict(dissect_name="bench", edit_loc=None, solver_kwargs=sk)
    tmp = None
    if c.get("hook"):
        tmp = tempfile.mkdtemp()
        rng = np.random.default_rng(11)
        table = (rng.standard_normal((40, net.seq_len, net.embed_dim)) * 0.01).astype(np.float32)
        for k in range(1, 101):
            np.save(os.path.join(tmp, f"delta_{k / 100:.2f}.npy"), table)
        kw.update(dissect_task="uspace_uvit", dissect_name="write_attr", edit_loc="mid", t_edit=0.4, write_scale=1.0,
                  ith_attr="31_39_20", write_path_root=tmp)
    with torch.no_grad():
        cnf.decode(z, cond, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = cnf.decode(z, cond, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    assert bool(torch.isfinite(out).all())
    res = dict(c, nfe=cnf.last_stats.nfe, seconds_per_solve=dt, images_per_sec_per_gpu=B / dt,
               ms_per_nfe=1e3 * dt / cnf.last_stats.nfe)
    del net, cnf
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    out = []
    only = os.environ.get("USPACE_BENCH_ONLY")           # e.g. "4" or "2,5": restrict to these config ids
    only = {int(v) for v in only.split(",")} if only else None
    for c in CONFIGS:
        if only and c["id"] not in only:
            continue
        r = run(c)
        out.append(r)
        print(json.dumps(r))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)
