"""Lab: interleaved A/B of uspace_gemm_bf16 (plain epilogues, no LayerNorm fold) across library builds:
   python tools/lab/gemm_plain_ab.py libA.so libB.so ..."""
import ctypes, sys
import torch

names = sys.argv[1:]
libs = [ctypes.CDLL(p) for p in names]
vp, ci = ctypes.c_void_p, ctypes.c_int
for lib in libs:
    lib.uspace_gemm_bf16.argtypes = [vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, ci, vp, ci, vp]
B_, G_, R_, F_, H_ = 1, 2, 4, 8, 16
CASES = [("qkv  -> bf16", 16448, 3072, 1024, H_), ("fc1  +bias +GELU -> bf16", 16448, 4096, 1024, B_ | G_ | H_), ("proj +bias +residual -> fp32", 16448, 1024, 1024, B_ | R_ | F_),
         ("fc2  +bias +residual -> fp32 + bf16", 16448, 1024, 4096, B_ | R_ | F_ | H_), ("4096^3 -> bf16", 4096, 4096, 4096, H_), ("8192^3 -> bf16", 8192, 8192, 8192, H_)]
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K, fl in CASES:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda") * 0.1
    outs = []
    for lib in libs:
        x = torch.randn(M, N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) if fl & (R_ | F_) else None
        ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if fl & H_ else None
        outs.append((x, ob))
    def call(lib, x, ob):
        rc = lib.uspace_gemm_bf16(A.data_ptr(), K, None, 0, K, W.data_ptr(), K, M, N, K, fl, bias.data_ptr(), x.data_ptr() if (x is not None and fl & R_) else None, N,
                                  x.data_ptr() if x is not None else None, N, ob.data_ptr() if ob is not None else None, N, st)
        assert rc == 0, rc
    for lib, (x, ob) in zip(libs, outs):
        call(lib, x, ob)
    torch.cuda.synchronize()
    ref = outs[0]
    eq = [bool((ref[0] is None or torch.equal(ref[0], x)) and (ref[1] is None or torch.equal(ref[1], ob))) for x, ob in outs]
    res = [[] for _ in libs]
    reps = 10 if M * N * K < 2e11 else 4
    for rnd in range(9):
        for i, (lib, (x, ob)) in enumerate(zip(libs, outs)):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): call(lib, x, ob)
            e1.record(); torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) * 1e3 / reps)
    fl_ = 2.0 * M * N * K
    print(f"{name:38s} " + " | ".join(f"{sorted(r)[len(r)//2]:8.1f} us (min {min(r):8.1f}, {fl_ / sorted(r)[len(r)//2] * 1e-6:6.0f} TF)" for r in res) + f" | first call bit-equal to first: {eq}", flush=True)
