"""Lab: cycle stamps around the K-loop barriers of the 256x256 GEMM.  The stamps (K_STAMP / K_PHASE, switch USPACE_KTRACE) left the product
source in round 5: re-apply them to a scratch copy first --
   patch -o /tmp/gemm_ktrace.hip uspace_amd/csrc/gemm.hip tools/lab/dropped/gemm_ktrace_fulllines_r04.patch
and build that copy with -DUSPACE_LAB=1 -DUSPACE_KTRACE=1 the way tools/lab/build_variant.sh builds gemm.hip -- then
   python tools/lab/gemm_trace.py tools/lab/_build/lib_ktrace.so [N K]
fc1-shaped launch (M = 64*257, +bias +GELU -> bf16) by default.  Workgroups 10 (first round of tiles) and 600 (third), every wave:
slot 0 kernel start, 1 first tile visible, 2+2kt / 3+2kt before / after the barrier that ends K tile kt, 62 K loop done, 63 kernel end."""
import ctypes, sys
import numpy as np
import torch

lib = ctypes.CDLL(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
M = 64 * 257
vp, ci = ctypes.c_void_p, ctypes.c_int
lib.uspace_gemm_bf16.argtypes = [vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, ci, vp, ci, vp]
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
EPI = 1 | 2 | 16      # bias, GELU, bf16 out
for _ in range(5):
    rc = lib.uspace_gemm_bf16(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, M, N, K, EPI, bias.data_ptr(), None, 0, None, 0, out.data_ptr(), N, st)
    assert rc == 0, rc
torch.cuda.synchronize()
buf = np.zeros(2 * 8 * 64, dtype=np.uint32)
assert lib.uspace_lab_gemm_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(2, 8, 64).astype(np.int64)
nk = K // 64
for blk, name in enumerate(("workgroup 10", "workgroup 600")):
    r = t[blk]
    t0 = r[:, 0].min()
    d = lambda a, b: (a - b) & 0xFFFFFFFF
    print(f"{name}: start skew {[int(d(x, t0)) for x in r[:, 0]]}")
    print(f"  prologue (start -> first tile visible): {[int(d(r[w, 1], r[w, 0])) for w in range(8)]}")
    wait_tot = np.zeros(8, dtype=np.int64); work_tot = np.zeros(8, dtype=np.int64)
    for kt in range(nk - 1):
        before, after = r[:, 2 + 2 * kt], r[:, 3 + 2 * kt]
        prev = r[:, 1] if kt == 0 else r[:, 3 + 2 * (kt - 1)]
        work = [int(d(before[w], prev[w])) for w in range(8)]
        wait = [int(d(after[w], before[w])) for w in range(8)]
        wait_tot += wait; work_tot += work
        print(f"  K tile {kt:2d}: work {work} | barrier wait {wait}")
    if len(sys.argv) > 4 and sys.argv[4] == "vm":
        print("  wait for the wave's own LDS-DMA in front of the barrier | barrier after that:")
        for kt in range(nk - 1):
            vm = [int(d(r[w, 32 + kt], r[w, 2 + 2 * kt])) for w in range(8)]
            ba = [int(d(r[w, 3 + 2 * kt], r[w, 32 + kt])) for w in range(8)]
            print(f"    K tile {kt:2d}: vmcnt wait {vm} | barrier {ba}")
    elif r[:, 32:56].any():
        print("  phases of K tiles 4..11 (after barrier -> 8 left-over MFMAs + 16 of phase 0 | phase 1 (16) | phase 2 (16) | phase 3 first 8 -> at barrier | wait):")
        for kt in range(4, min(12, nk - 1)):
            for w in (0, 4, 1, 5):
                a = r[w, 3 + 2 * (kt - 1)]; p = [r[w, 32 + 3 * (kt - 4) + i] for i in range(3)]; b = r[w, 2 + 2 * kt]; c = r[w, 3 + 2 * kt]
                print(f"    kt {kt:2d} wave {w}: {int(d(p[0], a)):5d} | {int(d(p[1], p[0])):5d} | {int(d(p[2], p[1])):5d} | {int(d(b, p[2])):5d} | {int(d(c, b)):5d}")
    last = [int(d(r[w, 62], r[w, 3 + 2 * (nk - 2)])) for w in range(8)]
    print(f"  last K tile (no barrier): {last}")
    print(f"  epilogue: {[int(d(r[w, 63], r[w, 62])) for w in range(8)]}")
    print(f"  per wave over {nk - 1} tiles: work {list(map(int, work_tot))}  wait {list(map(int, wait_tot))}; whole kernel {[int(d(r[w, 63], r[w, 0])) for w in range(8)]}")
