"""Lab: can a whole fixed-step solve be captured as ONE hipGraph (torch.cuda.graph around CNF.decode)?  Config 1 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import COMMON, MODELS
from uspace_amd.tools.utils_uvit import get_nnet
from uspace_amd.flow_matching import CNF

model, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = dict(MODELS[model]); name = cfg.pop("name")
torch.manual_seed(1234)
net = get_nnet(name, **COMMON, **cfg).cuda().eval()
cnf = CNF(net)
z = torch.randn(B, 4, 32, 32, generator=torch.Generator().manual_seed(7)).cuda()
kw = dict(dissect_name="bench", edit_loc=None, solver_kwargs=dict(solver="fixed", solver_fix="euler", solver_fix_step=1.0 / steps,
                                                               solver_adaptive="dopri5", solver_adaptive_prec=0.01, n_steps=steps))
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    ref = cnf.decode(z, None, **kw)
    base = t(lambda: cnf.decode(z, None, **kw))
    print(f"per-evaluation graph replay: {base*1e3:.2f} ms per solve = {B/base:.1f} images/s")
    net.use_graph = False
    eager = t(lambda: cnf.decode(z, None, **kw))
    print(f"eager launches:              {eager*1e3:.2f} ms per solve")
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            cnf.decode(z, None, **kw)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = cnf.decode(z, None, **kw)
        whole = t(lambda: g.replay())
        print(f"whole solve as one graph:    {whole*1e3:.2f} ms per solve = {B/whole:.1f} images/s; max |diff| vs per-evaluation graphs {float((out-ref).abs().max()):.3e}")
    except Exception as ex:
        print("capture failed:", repr(ex)[:400])
