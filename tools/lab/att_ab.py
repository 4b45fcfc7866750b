"""Lab: interleaved A/B timing of the attention launch across library builds (ctypes): python tools/lab/att_ab.py libA.so libB.so ..."""
import ctypes, sys, torch
names = sys.argv[1:]
libs = [ctypes.CDLL(p) for p in names]
for lib in libs:
    lib.uspace_attention_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for B, L, H in [(64, 257, 16), (64, 334, 16), (32, 257, 16), (64, 257, 8), (4, 257, 8)]:
    qkv = torch.randn(B * L, 3 * H * 64, device='cuda').to(torch.bfloat16)
    outs = [torch.empty(B * L, H * 64, device='cuda', dtype=torch.bfloat16) for _ in libs]
    st = torch.cuda.current_stream().cuda_stream
    runs = [(lambda f=lib.uspace_attention_bf16, o=o: f(qkv.data_ptr(), None, o.data_ptr(), B, L, H, st)) for lib, o in zip(libs, outs)]
    for r in runs:
        for _ in range(3): r()
    torch.cuda.synchronize()
    res = [[] for _ in libs]
    for rnd in range(9):
        for i, r in enumerate(runs):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): r()
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 1e3 / 30)
    eq = [bool(torch.equal(outs[0], o)) for o in outs]
    print(f"B={B} L={L} H={H}: " + " | ".join(f"{sorted(r)[len(r)//2]:.1f} ({min(r):.1f}) us" for r in res) + f" | bit-equal to first: {eq}", flush=True)
