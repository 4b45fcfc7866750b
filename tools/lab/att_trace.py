"""Lab: phase cycle stamps of the L = 257 attention launch (library built with -DUSPACE_ATT_TRACE=1):
   python tools/lab/att_trace.py tools/lab/_build/lib_att_trace.so
Prints, for workgroups 0 (first round: staging contended) and 700 (second round), per wave: staging, then per query tile the cycles
of Q.K^T | row maximum | exponentials + P.V | normalise + store + next Q."""
import ctypes, sys
import numpy as np
import torch

lib = ctypes.CDLL(sys.argv[1])
lib.uspace_attention_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B, L, H = 64, 257, 16
qkv = torch.randn(B * L, 3 * H * 64, device="cuda").to(torch.bfloat16)
out = torch.empty(B * L, H * 64, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    lib.uspace_attention_bf16(qkv.data_ptr(), None, out.data_ptr(), B, L, H, st)
torch.cuda.synchronize()
buf = np.zeros(2 * 4 * 64, dtype=np.uint64)
assert lib.uspace_lab_att_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(2, 4, 64).astype(np.int64)
for blk, name in enumerate(("workgroup 0", "workgroup 700")):
    t0 = t[blk, :, 0].min()
    print(name)
    for w in range(4):
        r = t[blk, w]
        n_tiles = 5 if w == 0 else 4
        print(f"  wave {w}: start +{r[0] - t0}, staged after {r[1] - r[0]} cycles")
        tot = []
        for i in range(n_tiles):
            s = r[2 + 4 * i: 6 + 4 * i]
            prev = r[1 + 4 * i]
            d = [s[0] - prev, s[1] - s[0], s[2] - s[1], s[3] - s[2]]
            tot.append(s[3] - prev)
            print(f"    tile {i}: QK {d[0]:5d} | max {d[1]:5d} | exp+PV {d[2]:5d} | store {d[3]:5d} | sum {s[3] - prev:5d}")
        print(f"    tiles: {sum(tot)} cycles, end at +{r[1 + 4 * n_tiles] - t0}")
