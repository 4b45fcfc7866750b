"""Lab: attention launch time of two library builds (ctypes), L and batch from the command line pairs below."""
import ctypes, sys, torch
libs = [ctypes.CDLL(p) for p in sys.argv[1:3]]
for B, L, H in [(4, 334, 16), (4, 334, 8), (8, 334, 16), (2, 334, 8), (16, 334, 16), (64, 334, 16)]:
    qkv = torch.randn(B * L, 3 * H * 64, device='cuda').to(torch.bfloat16)
    out = torch.empty(B * L, H * 64, device='cuda', dtype=torch.bfloat16)
    res = []
    for lib in libs:
        f = lib.uspace_attention_bf16
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        run = lambda: f(qkv.data_ptr(), None, out.data_ptr(), B, L, H, torch.cuda.current_stream().cuda_stream)
        for _ in range(3): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 50)
    print(f"B={B} L={L} H={H}: " + " | ".join(f"{r:.1f} us" for r in res))
