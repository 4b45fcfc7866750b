#!/usr/bin/env python3
"""Headline benchmark: images/sec of U-ViT-L 256^2-latent flow-matching sampling, batch 64 per GPU,
50 Dormand-Prince steps (BASELINE.json configs[1]; "dopri5-50" of BASELINE.md = 301 network
evaluations per solve), on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full latent -> latent solve of a batch of 64 synthetic latents per GPU (weak scaling:
independent trajectories, no data-path collective; the final latents are all-gathered once per solve).
Prints ONE JSON line on rank 0.  Extra fields: NFE, the Euler-50 rate, the roofline of the dominant
kernel (fc1 GEMM) from live HIP events, and the CPU oracle timed on this host (cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODELS = {
    "L_u": dict(name="uvit", embed_dim=1024, depth=20, num_heads=16, num_classes=-1),
    "L_t": dict(name="uvit_t2i", embed_dim=1024, depth=20, num_heads=16, clip_dim=768, num_clip_token=77),
    "S_u": dict(name="uvit", embed_dim=512, depth=16, num_heads=8, num_classes=-1),
    "S_t": dict(name="uvit_t2i", embed_dim=512, depth=16, num_heads=8, clip_dim=768, num_clip_token=77),
}
COMMON = dict(img_size=32, patch_size=2, in_chans=4, mlp_ratio=4, qkv_bias=False, mlp_time_embed=False)
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flops_per_sample(D, depth, L, t2i):
    nb, ns = depth + 1, depth // 2
    f = 2 * L * D * D * (12 * nb + 2 * ns) + 4 * L * L * D * nb + 2 * 256 * 16 * D + 2 * L * D * 16 + 2 * 4 * 4 * 9 * 1024
    if t2i:
        f += 2 * 77 * 768 * D
    return f


def cpu_baseline(model_key, nfe, budget_s=12.0):
    """The CPU oracle (oracle/, a port of the reference forward) timed on this host on a bounded sample."""
    from oracle import _cops
    from oracle import uvit_oracle as O
    cfg = MODELS[model_key]
    t2i = cfg["name"] == "uvit_t2i"
    spec = O.UViTSpec(img_size=32, patch_size=2, in_chans=4, embed_dim=cfg["embed_dim"], depth=cfg["depth"],
                      num_heads=cfg["num_heads"], t2i=t2i)
    rng = np.random.default_rng(1234)
    D, Hd = spec.D, spec.hidden
    sd = {"pos_embed": rng.standard_normal((1, spec.L, D), dtype=np.float32) * 0.02,
          "patch_embed.proj.weight": rng.standard_normal((D, 4, 2, 2), dtype=np.float32) * 0.1,
          "patch_embed.proj.bias": np.zeros(D, np.float32)}
    if t2i:
        sd["context_embed.weight"] = rng.standard_normal((D, 768), dtype=np.float32) * 0.02
        sd["context_embed.bias"] = np.zeros(D, np.float32)
    for b in spec.block_names():
        for n, shp in (("norm1.weight", (D,)), ("norm2.weight", (D,))):
            sd[f"{b}.{n}"] = np.ones(shp, np.float32)
        for n in ("norm1.bias", "norm2.bias", "attn.proj.bias", "mlp.fc2.bias"):
            sd[f"{b}.{n}"] = np.zeros(D, np.float32)
        sd[f"{b}.mlp.fc1.bias"] = np.zeros(Hd, np.float32)
        sd[f"{b}.attn.qkv.weight"] = rng.standard_normal((3 * D, D), dtype=np.float32) * 0.02
        sd[f"{b}.attn.proj.weight"] = rng.standard_normal((D, D), dtype=np.float32) * 0.02
        sd[f"{b}.mlp.fc1.weight"] = rng.standard_normal((Hd, D), dtype=np.float32) * 0.02
        sd[f"{b}.mlp.fc2.weight"] = rng.standard_normal((D, Hd), dtype=np.float32) * 0.02
        if b.startswith("out_blocks"):
            sd[f"{b}.skip_linear.weight"] = rng.standard_normal((D, 2 * D), dtype=np.float32) * 0.02
            sd[f"{b}.skip_linear.bias"] = np.zeros(D, np.float32)
    sd["norm.weight"], sd["norm.bias"] = np.ones(D, np.float32), np.zeros(D, np.float32)
    sd["decoder_pred.weight"] = rng.standard_normal((16, D), dtype=np.float32) * 0.02
    sd["decoder_pred.bias"] = np.zeros(16, np.float32)
    sd["final_layer.weight"] = rng.standard_normal((4, 4, 3, 3), dtype=np.float32) * 0.1
    sd["final_layer.bias"] = np.zeros(4, np.float32)
    Bs = 2
    x = rng.standard_normal((Bs, 4, 32, 32), dtype=np.float32)
    ctx = rng.standard_normal((Bs, 77, 768), dtype=np.float32) if t2i else None
    lib = _cops.lib()
    max_threads = lib.oracle_num_threads()
    fwd = lambda xb, cb: O.uvit_forward(spec, sd, xb, 0.5, context=cb, edit_loc=None)
    fwd(x[:1], None if ctx is None else ctx[:1])                                                      # warm-up
    # many-core hosts: the OpenMP port stops scaling well before all hardware threads; pick the best count
    best = None
    for n in sorted({min(max_threads, c) for c in (8, 16, 32, 64, max_threads)}):
        lib.oracle_set_threads(n)
        t0 = time.perf_counter()
        fwd(x[:1], None if ctx is None else ctx[:1])
        dt1 = time.perf_counter() - t0
        if best is None or dt1 < best[1]:
            best = (n, dt1)
    threads = best[0]
    lib.oracle_set_threads(threads)
    t0 = time.perf_counter()
    reps = 0
    while True:
        fwd(x, ctx)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 8:
            break
    per_fwd = el / reps
    lib.oracle_set_threads(max_threads)
    return dict(value=Bs / (per_fwd * nfe), unit="images/sec", cores=int(threads), kind="port",
                note="unoptimised C/OpenMP restatement (oracle/ops.c): a correctness checker timed for context, not a tuned CPU "
                     "implementation; the reference's own PyTorch-CPU path is faster per core (reference_pytorch_cpu below)",
                sample=f"{reps} fp32 forwards of batch {Bs} of the C/OpenMP oracle port ({per_fwd:.2f} s each), "
                       f"extrapolated linearly to {nfe} NFE per solve; host {os.cpu_count()} logical CPUs")


def _gemm_role(flags, N, K, D):
    if K == 2 * D:
        return "skip_linear"
    if K == 4 * D:
        return "fc2"
    if N == 3 * D:
        return "qkv"
    if N == 4 * D:
        return "fc1"
    if N == D and K == D:
        return "proj"
    return "other"


def roofline_rows(recs, D):
    """bench.py's roofline_all: one row per distinct (kernel kind, epilogue flags, M, N, K) of the recorded eager solve, priced
    against BOTH roofs (dense bf16 MFMA peak; HBM peak with the launch's ALGORITHMIC bytes: operands once + outputs once)."""
    from uspace_amd import _hip
    rows = []
    for r in recs:
        n, avg_s = r["launches"], r["total_ms"] / max(r["launches"], 1) / 1e3
        if r["kind"] == 0:
            M, N, K, f = r["M"], r["N"], r["K"], r["flags"]
            flops = 2.0 * M * N * K
            byts = 2.0 * M * K + 2.0 * N * K
            byts += 4.0 * M * N * (bool(f & _hip.EPI_RESIDUAL) + bool(f & _hip.EPI_OUT_F32))
            byts += 2.0 * M * N * (bool(f & _hip.EPI_OUT_BF16) + bool(f & _hip.EPI_CEN_OUT))
            names = [nm for nm, bit in (("LN_IN", _hip.EPI_LN_IN), ("BIAS", _hip.EPI_BIAS), ("GELU", _hip.EPI_GELU), ("RESIDUAL", _hip.EPI_RESIDUAL),
                                        ("OUT_F32", _hip.EPI_OUT_F32), ("OUT_BF16", _hip.EPI_OUT_BF16), ("CEN_OUT", _hip.EPI_CEN_OUT),
                                        ("RANK1", _hip.EPI_RANK1)) if f & bit]
            row = dict(kernel="gemm_kernel", role=_gemm_role(f, N, K, D), epi="|".join(names), epi_flags=f, M=M, N=N, K=K)
        else:
            BH, L, hd = r["M"], r["N"], r["K"]
            flops = 4.0 * L * L * hd * BH
            byts = 2.0 * BH * L * hd * 4                      # q, k, v read + output written, bf16
            row = dict(kernel="attention_kernel", role="attention", epi="key_scale" if r["flags"] else "", epi_flags=r["flags"], M=BH, N=L, K=hd)
        tf, gbs = flops / avg_s / 1e12, byts / avg_s / 1e9
        row.update(launches=n, avg_us=1e6 * avg_s, flops_per_launch=flops, algorithmic_bytes_per_launch=byts, tflops=tf, gbs=gbs,
                   frac_mfma=tf / MFMA_BF16_PEAK_TFLOPS, frac_hbm=gbs / HBM_PEAK_GBS,
                   bound="mfma" if tf / MFMA_BF16_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS else "hbm")
        rows.append(row)
    rows.sort(key=lambda q: -q["avg_us"] * q["launches"])
    return rows


GEMM_SOURCES = ("gemm.hip", "gemm_args.h", "common.h")      # (tools/lab/gemm_chain.h is compiled only into lab variants)


def fc1_traffic(model, B, tile):
    """HBM bytes per fc1 launch from the committed PMC pass (profiles/fc1_traffic.json) -- only when that pass was taken on the
    GEMM sources that are being benchmarked (sha256 over gemm.hip + gemm_args.h + common.h) and on the same workload and tile form; otherwise null."""
    import hashlib
    tp = os.path.join(ROOT, "profiles", "fc1_traffic.json")
    if not os.path.exists(tp):
        return None, "no committed PMC pass"
    t = json.load(open(tp))
    h = hashlib.sha256()
    for f in GEMM_SOURCES:                   # everything the GEMM kernels are compiled from
        h.update(open(os.path.join(ROOT, "uspace_amd", "csrc", f), "rb").read())
    sha = h.hexdigest()
    if t.get("gemm_source_sha256") != sha:
        return None, (f"profiles/fc1_traffic.json was measured on other GEMM sources ({str(t.get('gemm_source_sha256'))[:12]} != {sha[:12]}, "
                      f"sha256 over {' + '.join(GEMM_SOURCES)}): dropped")
    if t.get("model") != model or t.get("batch") != B or list(t.get("tile", [])) != list(tile):
        return None, "profiles/fc1_traffic.json was measured on another workload / tile form: dropped"
    return t.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {t.get('source')}, {' + '.join(GEMM_SOURCES)} {sha[:12]}"


# BASELINE.json configs[i-1] -> workload; "dopri5" = 50 fixed Dormand-Prince steps (301 NFE), "euler" = 50 Euler steps
CONFIGS = {
    1: dict(model="S_u", batch=4, solver="euler", ode_steps=20, hook=False),
    2: dict(model="L_u", batch=64, solver="dopri5", ode_steps=50, hook=False),      # the headline
    3: dict(model="L_t", batch=64, solver="euler", ode_steps=50, hook=False),
    4: dict(model="S_t", batch=64, solver="euler", ode_steps=50, hook=False),       # 512 over 8 GPUs
    5: dict(model="L_u", batch=32, solver="euler", ode_steps=50, hook=True),        # 256 over 8 GPUs, mid-block u-space write hook
}


def hook_kwargs(net, tmpdir):
    """Synthetic direction tables of SURVEY.md 8(d): delta_{t:.2f}.npy [40, L, D] ~ N(0, 0.01^2), seed 11."""
    rng = np.random.default_rng(11)
    table = (rng.standard_normal((40, net.seq_len, net.embed_dim)) * 0.01).astype(np.float32)
    # the hook reads delta_{t:.2f}.npy for t <= t_edit only: one file on disk, the other names are hard links to it
    # (round 2 wrote 100 copies of 42 MB per rank, often into RAM-backed tmpfs)
    first = os.path.join(tmpdir, "delta_0.01.npy")
    np.save(first, table)
    for k in range(2, 42):
        dst = os.path.join(tmpdir, f"delta_{k / 100:.2f}.npy")
        try:
            os.link(first, dst)
        except OSError:
            np.save(dst, table)
    return dict(dissect_task="uspace_uvit", dissect_name="write_attr", edit_loc="mid", t_edit=0.4, write_scale=1.0,
                ith_attr="31_39_20", write_path_root=tmpdir)


class Runtime:
    """The device side of the benchmark: one ROCm device per rank, RCCL between them (backend "nccl" IS RCCL on ROCm), HIP events
    on the launching stream.  tests/test_multigpu_readiness.py substitutes a CPU + gloo stand-in (and stubs the kernel launches at
    uspace_amd._hip.lib()) so that main() -- rank discovery, process group, the sharded solves, the gather, the self-check of the
    gathered batch and the JSON line -- runs end to end on two ranks without GPUs."""
    backend = "nccl"

    def __init__(self, local_rank):
        assert torch.cuda.is_available(), "bench.py needs ROCm devices"
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)

    def init_group(self, rank, world):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(self.backend, rank=rank, world_size=world, device_id=self.device, timeout=group_timeout())

    def synchronize(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def device_name(self):
        p = torch.cuda.get_device_properties(self.device)
        return f"{p.name} ({getattr(p, 'gcnArchName', '?')}, {p.multi_processor_count} CUs)"


RUNTIME = Runtime


def group_timeout():
    """Bound on every collective of the run (default 10 minutes; USPACE_BENCH_PG_TIMEOUT_S): a rank that dies mid-solve must not leave the
    others waiting in the gather for ever.  (Under `python -m torch.distributed.run` the agent also ends the other ranks as soon as one exits
    with an error; this bound is for every other launch form and for a rank that hangs instead of dying.)"""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("USPACE_BENCH_PG_TIMEOUT_S", "600")))


def relaunch_argv(gpus, port, argv):
    """The one-rank-per-GPU launch of this script on one node (the form the driver uses for N > 1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def rank_env(gpus, env=None):
    """(world, rank, local_rank) from the launcher's environment; the world size must be the --gpus that was asked for."""
    env = os.environ if env is None else env
    world, rank, local_rank = int(env.get("WORLD_SIZE", "1")), int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
    if world != gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {gpus})")
    if not (0 <= rank < world and 0 <= local_rank < world):
        raise SystemExit(f"bench.py: RANK={rank} LOCAL_RANK={local_rank} outside WORLD_SIZE={world}")
    return world, rank, local_rank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[N-1]; 2 = the headline")
    ap.add_argument("--model", default=None, choices=sorted(MODELS))
    ap.add_argument("--batch", type=int, default=None, help="latents per GPU")
    ap.add_argument("--solver", default=None, choices=["dopri5", "euler"])
    ap.add_argument("--ode-steps", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the auxiliary Euler-50 / VAE / roofline measurements")
    args = ap.parse_args()
    wl = dict(CONFIGS[args.config])
    for k in ("model", "batch", "solver", "ode_steps"):
        if getattr(args, k) is not None:
            wl[k] = getattr(args, k)
    args.model, args.batch, args.solver, args.ode_steps = wl["model"], wl["batch"], wl["solver"], wl["ode_steps"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch as one rank per GPU (what the driver does itself)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        argv = relaunch_argv(args.gpus, port, sys.argv[1:])
        if os.environ.get("USPACE_BENCH_PRINT_RELAUNCH") == "1":      # tests: show the command instead of running it
            print(json.dumps(argv))
            return
        os.execvp(argv[0], argv)
    world, rank, local_rank = rank_env(args.gpus)
    rt = RUNTIME(local_rank)
    dev = rt.device
    ranks_seen = [dict(rank=rank, local_rank=local_rank, device=rt.device_name(), pid=os.getpid())]
    if world > 1:
        rt.init_group(rank, world)
        _w = torch.full((8,), float(rank), device=dev)        # create the communicator outside the timed region ...
        got = [torch.empty_like(_w) for _ in range(world)]
        dist.all_gather(got, _w)
        rt.synchronize()
        # ... and prove it: every rank of the launch answered, in rank order, each from its own process and device
        assert [int(g[0].item()) for g in got] == list(range(world)), "all_gather did not return the ranks in order"
        objs = [None] * world
        dist.all_gather_object(objs, ranks_seen[0])
        ranks_seen = objs
        assert [o["rank"] for o in ranks_seen] == list(range(world)) and len({o["pid"] for o in ranks_seen}) == world

    from uspace_amd import _hip
    from uspace_amd.sampling import gather_batch
    from uspace_amd.tools.utils_uvit import get_nnet

    cfg = dict(MODELS[args.model])
    name = cfg.pop("name")
    t2i = name == "uvit_t2i"
    torch.manual_seed(1234)                                    # reference init, SURVEY.md §8(d)
    net = get_nnet(name, **COMMON, **cfg).to(dev).eval()
    if t2i:
        from uspace_amd.flow_matching_t2i import CNF
    else:
        from uspace_amd.flow_matching import CNF
    cnf = CNF(net)                                             # product defaults: eager launches (USPACE_UVIT_GRAPH=1: hipGraph replay)
    if os.environ.get("USPACE_BENCH_EAGER") == "1":            # profiling aid: rocprofv3's kernel trace crashes on graph replays
        net.use_graph = False
    B = args.batch
    g = torch.Generator().manual_seed(7 + rank)
    z = torch.randn(B, 4, 32, 32, generator=g).to(dev)
    cond = torch.randn(B, 77, 768, generator=g).to(dev) if t2i else None
    import tempfile
    tmp = tempfile.TemporaryDirectory()
    hk = hook_kwargs(net, tmp.name) if wl["hook"] else None

    def solver_kwargs(kind):
        sk = dict(solver_fix="euler", solver_fix_step=1.0 / args.ode_steps, solver_adaptive="dopri5",
                  solver_adaptive_prec=0.01, n_steps=args.ode_steps)
        sk["solver"] = "adaptive" if kind in ("dopri5", "dopri5_adaptive") else "fixed"
        if kind == "dopri5_adaptive":          # the reference's default sampler: error-controlled dopri5 at rtol = atol = 1e-5
            del sk["n_steps"]                  # (flow_matching.py:78-84); the NFE is data-dependent
        return sk

    last_local = [None]

    def solve(kind, gather=True):
        kw = dict(dissect_name="bench", edit_loc=None, solver_kwargs=solver_kwargs(kind))
        if hk:
            kw.update(hk)
        out = cnf.decode(z, cond, **kw) if t2i else cnf.decode(z, None, **kw)
        last_local[0] = out
        return gather_batch(out, B * world) if gather else out   # the one collective of the sampling path

    def fence():
        rt.synchronize()
        if world > 1:
            dist.barrier()
        rt.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            solve(args.solver)
        # ---- the timed region: exactly `steps` solves between two fences; every solve also bracketed by HIP events on
        #      the stream the kernels run on (torch's current stream is the one handed to the C-ABI)
        ev = [(rt.event(), rt.event()) for _ in range(args.steps)]
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ev[i][0].record()
            res = solve(args.solver)
            ev[i][1].record()
            # every timed solve hands back the WHOLE job's batch (a shape check: no device work, no sync)
            assert res.shape[0] == B * world, f"gathered {res.shape[0]} rows, expected {B} x {world}"
        fence()
        dt = time.perf_counter() - t0
        nfe = cnf.last_stats.nfe
        per_solve_ms = sorted(a.elapsed_time(b) for a, b in ev)
        median_ms = per_solve_ms[len(per_solve_ms) // 2]
        assert bool(torch.isfinite(res).all())
        # rank r's rows of the gathered batch are rank r's own solve (bit-equal), on every rank
        mine_ok = bool(torch.equal(res[rank * B:(rank + 1) * B], last_local[0]))
        rank_dt, rank_median = dt, median_ms
        tt = torch.tensor([dt, median_ms, 0.0 if mine_ok else 1.0], dtype=torch.float64, device=dev)
        per_rank = [[rank_dt, rank_median]]
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mine = torch.tensor([rank_dt, rank_median], dtype=torch.float64, device=dev)
            allr = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [[float(v) for v in a.tolist()] for a in allr]
        dt, median_ms = float(tt[0].item()), float(tt[1].item())
        rows_ok = float(tt[2].item()) == 0.0        # MAX over ranks of "my rows of the LAST timed solve's gathered batch differ from my own solve"
        assert rows_ok, "a rank's rows of the gathered batch differ from its own solve"

        # ---- outside the timed region: rooflines.  One more solve with eager launches and HIP events recorded (by the
        #      library, on the launching stream) around every GEMM and attention launch
        D, Hd = cfg["embed_dim"], 4 * cfg["embed_dim"]
        fold = _hip.lib().uspace_uvit_get_ln_fold() != 0       # norm2 folded into fc1 (default) or a separate launch
        fc1_flags = _hip.EPI_BIAS | _hip.EPI_GELU | _hip.EPI_OUT_BF16 | (_hip.EPI_LN_IN if fold else 0)
        recs, peaks, gemm_op = [], None, None
        if rank == 0 and not args.no_extra:
            was = net.use_graph
            net.use_graph = False
            # one event pair per GEMM / attention launch of the whole solve: 5 per block + one skip_linear per out-block, per NFE
            rec_cap = (nfe + 2) * (5 * (cfg["depth"] + 1) + cfg["depth"] // 2 + 4)
            _hip.prof_all_begin(rec_cap)
            solve(args.solver, gather=False)                  # rank 0 only: no collective in here
            rt.synchronize()
            rec_dropped = _hip.prof_dropped()
            recs = _hip.prof_all_end()
            if rec_dropped:
                raise SystemExit(f"bench.py: the launch recorder was full ({rec_cap} launches) and dropped {rec_dropped}: roofline would cover a truncated solve")
            net.use_graph = was
            peaks = _hip.prof_peaks()
            gemm_op = _hip.prof_mfma_gemm_op()
        if world > 1:
            dist.barrier()

        extra = {}
        if not args.no_extra and args.solver == "dopri5":
            solve("euler")
            fence()
            t1 = time.perf_counter()
            solve("euler")
            fence()
            e = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(e, op=dist.ReduceOp.MAX)
            extra = dict(euler50_images_per_sec=B * world / float(e.item()), euler50_nfe=cnf.last_stats.nfe)
            # the reference-default sampler (adaptive dopri5, rtol = atol = 1e-5), outside the timed region.  Per-rank step
            # control (the reference under `accelerate launch`; every rank its own NFE) and, for N > 1, the group-controlled form
            # (CNF.norm_group: one all-reduced error norm per step attempt, the step sequence of the unsharded solve on every rank)
            def adaptive_reading(tag, group):
                prev = cnf.norm_group
                cnf.norm_group = group
                try:
                    solve("dopri5_adaptive")
                    fence()
                    t3 = time.perf_counter()
                    solve("dopri5_adaptive")
                    fence()
                    st = cnf.last_stats
                    a = torch.tensor([time.perf_counter() - t3, st.nfe, st.accepted, st.rejected], dtype=torch.float64, device=dev)
                    rows = [a]
                    if world > 1:
                        rows = [torch.empty_like(a) for _ in range(world)]
                        dist.all_gather(rows, a)
                    rows = [[float(v) for v in r.tolist()] for r in rows]
                    wall = max(r[0] for r in rows)
                    extra[f"{tag}_images_per_sec"] = B * world / wall
                    extra[f"{tag}_nfe"] = [int(r[1]) for r in rows] if world > 1 else int(rows[0][1])
                    extra[f"{tag}_steps_accepted"] = [int(r[2]) for r in rows] if world > 1 else int(rows[0][2])
                    extra[f"{tag}_steps_rejected"] = [int(r[3]) for r in rows] if world > 1 else int(rows[0][3])
                finally:
                    cnf.norm_group = prev
            try:                                   # auxiliary figures, never fatal (an error here is the same on every rank: host-side code)
                adaptive_reading("dopri5_adaptive", None)
                extra["dopri5_adaptive_note"] = ("reference default: dopri5, rtol = atol = 1e-5 (flow_matching.py:78-84), NFE data-dependent; per-rank step control; "
                                                 "one warm-up solve, then one timed solve (max over ranks)")
                if world > 1:
                    adaptive_reading("dopri5_adaptive_norm_group", True)
            except Exception as ex:
                extra["dopri5_adaptive_error"] = repr(ex)
            if world == 1 and dev.type == "cuda":
                # latents -> 256^2 images through the VAE decoder; outside the timed region and
                # outside `value`, reported so the latent->latent figure can be read as an end-to-end one
                try:
                    from uspace_amd.libs.autoencoder import get_model
                    torch.manual_seed(4321)
                    vae = get_model(None).to(dev)
                    lat = res[:32].contiguous()
                    vae.decode(lat, chunk=8)
                    fence()
                    t2 = time.perf_counter()
                    img = vae.decode(lat, chunk=8)
                    fence()
                    extra["vae_decode_images_per_sec"] = lat.shape[0] / (time.perf_counter() - t2)
                    assert img.shape == (lat.shape[0], 3, 256, 256) and bool(torch.isfinite(img).all())
                    del vae, img
                except Exception as ex:  # auxiliary figure, never fatal
                    extra["vae_decode_images_per_sec"] = repr(ex)

    if rank == 0:
        L = net.seq_len
        M = B * L
        fps = flops_per_sample(D, cfg["depth"], L, t2i)
        value = B * world * args.steps / dt
        line = {
            "metric": "images/sec, U-ViT-L 256 latent FM sampling (50 ODE steps, bs64) @1/2/4/8 GPU",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config - 1}]: {args.model} (U-ViT D={D} depth={cfg['depth']} L={L}), "
                                   f"batch {B}/GPU, {args.solver}-{args.ode_steps} fixed steps"
                                   + (", mid-block u-space write hook (t <= 0.4)" if wl["hook"] else "")
                                   + ", seeded random-init weights, latent->latent (VAE excluded), "
                                   + ("hipGraph replay per evaluation" if net.use_graph else "eager launches"),
                       "global_batch": B * world, "nfe_per_solve": nfe, "parallelism": f"batch-sharded x{world}"},
            # what ran, from the process group itself (not from the command line): ranks in the group, who they were, what each of
            # them measured, and that the gathered batch of the timed solves held every rank's rows (asserted above)
            "multi_gpu": {"backend": (dist.get_backend() if world > 1 else None), "rccl_world_size": (dist.get_world_size() if world > 1 else 1),
                          "ranks_seen": ranks_seen, "per_rank_wall_s": [p[0] for p in per_rank],
                          "per_rank_median_ms": [p[1] for p in per_rank], "gathered_rows": int(res.shape[0]),
                          "gathered_rows_expected": B * world,
                          # measured (all-reduced over the ranks), on the last timed solve; the row COUNT is asserted after every timed solve
                          "each_ranks_rows_equal_its_own_solve": rows_ok, "rows_checked_on": "the last timed solve"},
            "nfe": nfe,
            "library": _hip.library_info(),
            "median_ms_per_step": median_ms, "per_step_ms_rank0": per_solve_ms,
            "images_per_sec_median": B * world / (median_ms * 1e-3),
            "sample_nfe_per_sec": B * world * nfe * args.steps / dt,
            "model_tflops_per_gpu": fps * B * nfe * args.steps / dt / 1e12,
            "mfma_util_whole_solve": fps * B * nfe * args.steps / dt / 1e12 / MFMA_BF16_PEAK_TFLOPS,
        }
        line.update(extra)
        if recs:
            all_rows = roofline_rows(recs, D)
            line["roofline_all"] = all_rows
            fc1 = [r for r in all_rows if r.get("epi_flags") == fc1_flags and r["role"] == "fc1"]
            if fc1:
                r = max(fc1, key=lambda q: q["launches"])
                import ctypes
                out = (ctypes.c_int * 8)()
                _hip.check(_hip.lib().uspace_gemm_plan_k(r["M"], r["N"], r["K"], 0, out), "uspace_gemm_plan_k")
                tile = f"{out[2]},{out[3]}"
                waves = {(256, 256): "2,4", (192, 256): "2,4", (256, 128): "4,2", (128, 128): "2,2", (64, 64): "2,2"}[(out[2], out[3])]
                traffic, tnote = fc1_traffic(args.model, B, (out[2], out[3]))
                line["roofline"] = {"bound": "mfma", "kernel": f"gemm_kernel<{tile},{waves}," + ("LN_IN|" if fold else "") + "BIAS|GELU|OUT_BF16> (fc1)",
                                    "achieved": r["tflops"], "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": r["tflops"] / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_note": tnote,
                                    "launches": r["launches"], "avg_us": r["avg_us"], "flops_per_launch": r["flops_per_launch"],
                                    "recorder": {"capacity": rec_cap, "recorded": sum(q["launches"] for q in all_rows), "dropped": 0},
                                    "timing": "HIP events around EVERY GEMM and attention launch of one extra eager solve after the timed "
                                              "region (the recorder fails the run if it drops a launch)"}
                if peaks:
                    line["roofline"]["peak_measured"] = {
                        "mfma_bf16_tflops": peaks[0], "hbm_copy_gbs": peaks[1], "shader_ghz_under_mfma_loop": peaks[2],
                        "mfma_bf16_tflops_instruction": "v_mfma_f32_32x32x16_bf16, near-constant operands (burst figure)",
                        "frac_of_measured_mfma": r["tflops"] / peaks[0],
                        "gemm_instruction": {
                            "instruction": "v_mfma_f32_16x16x32_bf16 (the GEMM's), pseudo-random operands in [-1, 1)",
                            "tflops": gemm_op[0], "shader_ghz": gemm_op[1], "frac_of_it": r["tflops"] / gemm_op[0]} if gemm_op else None,
                        "loop": "20000 iterations x 32 v_mfma_f32_32x32x16_bf16 per wave, 1024 waves (one per SIMD), about 10 ms: long enough for the "
                                "clock to settle at its sustained value (the guide's 2495 TFLOP/s is a short burst at 2.4 GHz)",
                        "how": "v_mfma_f32_32x32x16_bf16-only loop, one wave on every SIMD; the chip clocks to its power budget, so the "
                               "measured rate = 2.5 PF x sustained clock / 2.4 GHz; 1 GiB device-to-device float4 copy (read + write bytes)"}
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(args.model, nfe)
                rp = os.path.join(ROOT, "tests", "golden", "ref_cpu_timing.json")
                if os.path.exists(rp) and args.config in (1, 2):
                    r = json.load(open(rp))
                    key = {1: "cfg1_images_per_s", 2: "cfg2_L_u_B64_dopri5_50_images_per_s_extrapolated" if args.solver == "dopri5"
                           else "cfg2_L_u_B64_euler50_images_per_s_extrapolated"}[args.config]
                    line["cpu_baseline"]["reference_pytorch_cpu"] = {
                        "value": r[key], "unit": "images/sec", "cores": r["threads"], "kind": "reference",
                        "host": f"build container, {r['cpu']}, {r['nproc']} CPUs (NOT this GPU host; the reference cannot travel)",
                        "sample": "the reference's own PyTorch-CPU fp32 forward timed by tests/golden/make_golden.py"
                                  + (" at batch 8 and extrapolated linearly in batch and NFE" if args.config == 2 else "")}
            except Exception as ex:  # the baseline is reported context, never fatal
                line["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(line))
    tmp.cleanup()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as ex:   # name the rank: with N ranks the first error is the one that matters, the others fail in the gather after it
        if not isinstance(ex, SystemExit) or ex.code not in (0, None):
            print(f"bench.py: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} failed: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)
        raise
