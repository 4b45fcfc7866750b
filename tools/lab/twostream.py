"""Experiment: one batch-64 solve vs two concurrent batch-32 solves on two HIP streams (two copies of the network's
workspaces), so that one half's epilogue store bursts / prologues overlap the other half's K loops.
    python tools/lab/twostream.py [steps] [solver]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from bench import COMMON, MODELS  # noqa: E402
from uspace_amd.flow_matching import CNF  # noqa: E402
from uspace_amd.tools.utils_uvit import get_nnet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
solver = sys.argv[2] if len(sys.argv) > 2 else "euler"
nsplit = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
cfg = dict(MODELS["L_u"])
name = cfg.pop("name")
nets = []
for i in range(nsplit):
    torch.manual_seed(1234)
    nets.append(get_nnet(name, **COMMON, **cfg).to(dev).eval())
cnfs = [CNF(n) for n in nets]
B = 64
z = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(7)).to(dev)
sk = dict(solver_fix="euler", solver_fix_step=1.0 / 50, solver_adaptive="dopri5", solver_adaptive_prec=0.01, n_steps=50,
          solver="adaptive" if solver == "dopri5" else "fixed")
kw = dict(dissect_name="bench", edit_loc=None, solver_kwargs=sk)


def full():
    return cnfs[0].decode(z, None, **kw)


streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
outs = [None] * nsplit


def part(i):
    torch.cuda.set_device(dev)
    h = B // nsplit
    with torch.no_grad(), torch.cuda.stream(streams[i]):
        outs[i] = cnfs[i].decode(z[i * h:(i + 1) * h].contiguous(), None, **kw)
        streams[i].synchronize()


def split():
    th = [threading.Thread(target=part, args=(i,)) for i in range(nsplit)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return torch.cat(outs)


with torch.no_grad():
    r0 = full()
    r1 = split()
    torch.cuda.synchronize()
    print("max |full - split| =", float((r0 - r1).abs().max()), " rel", float((r0 - r1).norm() / r0.norm()))
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            full()
        torch.cuda.synchronize()
        t_full = (time.perf_counter() - t0) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            split()
        torch.cuda.synchronize()
        t_split = (time.perf_counter() - t0) / steps
        print(f"round {rnd}: {solver}-50: one batch-64 solve {t_full*1e3:.1f} ms = {B/t_full:.2f} img/s | {nsplit} concurrent batch-{B//nsplit} solves {t_split*1e3:.1f} ms = {B/t_split:.2f} img/s", flush=True)
