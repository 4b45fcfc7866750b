"""N > 1 readiness without multi-GPU hardware: the launch form of bench.py, and the two multi-process
sampling loops of the reference (tools/utils_uvit.py:264-277 sample2dir, tools/utils_vis.py:168-241 the u-space sweep of BASELINE
config 5) over DistAccelerator on world_size-2 gloo processes, with the hook keywords of config 5 (per-rank ``batch_id``,
direction tables shared on disk, one DeltaCache per process)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ------------------------------------------------------------------------------------------- bench.py launch form
def test_bench_relaunch_argv_and_world_size_check():
    env = dict(os.environ, USPACE_BENCH_PRINT_RELAUNCH="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-400:]
    argv = json.loads(out.stdout.strip().splitlines()[-1])
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and int(argv[argv.index("--master-port") + 1]) > 0
    i = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]          # the ranks run the same command line
    # a rank started with the wrong world size stops before touching a device
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--gpus 2 but WORLD_SIZE=3" in (bad.stderr + bad.stdout)
    sys.path.insert(0, ROOT)
    import bench
    assert bench.rank_env(8, dict(WORLD_SIZE="8", RANK="5", LOCAL_RANK="5")) == (8, 5, 5)
    assert bench.rank_env(1, {}) == (1, 0, 0)
    with pytest.raises(SystemExit):
        bench.rank_env(2, dict(WORLD_SIZE="2", RANK="2", LOCAL_RANK="0"))


# ------------------------------------------------------------------------------------------- the two sampling loops, 2 ranks
L_TOK, D_EMB, N_ATTR = 5, 8, 40
STEPS = [k / 10 for k in range(10)]                # t = 0.0 .. 0.9 (Euler-10 grid)


def _hook_kwargs(root):
    return dict(dissect_task="uspace_uvit", dissect_name="write_attr", edit_loc="mid", t_edit=0.4, ith_attr="31_39_20",
                write_path_root=root, has_attr=False, dataset_name="celeba256", seed=3)


def _write_tables(root):
    rng = np.random.default_rng(11)
    for k in range(0, 10):
        np.save(os.path.join(root, f"delta_{k / 10:.2f}.npy"), (rng.standard_normal((N_ATTR, L_TOK, D_EMB)) * 0.1).astype(np.float32))


class _StandInSolve:
    """What a hooked solve does on the host side, with the network replaced by x <- 0.9 x + 0.01 (batch_id + 1): the product's
    plan_uspace_hook / DeltaCache decide and fetch the edit of every step exactly as libs/uvit.py does around the mid block."""

    def __init__(self):
        from uspace_amd.libs.dissection import DeltaCache
        self.cache = DeltaCache()
        self.loads = 0
        self.batch_ids = []

    def __call__(self, input_z, write_scale, batch_id, **kwargs):
        from uspace_amd.libs.dissection import plan_uspace_hook
        self.batch_ids.append(batch_id)
        x = input_z.reshape(input_z.shape[0], -1).clone()
        for t in STEPS:
            plan = plan_uspace_hook(f"{t:.2f}", dict(kwargs, write_scale=write_scale, batch_id=batch_id))
            x = 0.9 * x + 0.01 * (batch_id + 1)
            if plan is not None:
                before = len(self.cache._store)
                d = self.cache.get(plan.path, plan.ith, torch.device("cpu"), L_TOK * D_EMB)
                self.loads += len(self.cache._store) - before
                x = x + d[None] * plan.scale
        return x.reshape(input_z.shape[0], 1, L_TOK, D_EMB)


def _expected_rows(rank, world, mini, scales, root, rounds):
    """Single-process recomputation of what rank `rank` contributes: [rounds][mini * n_scales, 1, L, D]."""
    g = torch.Generator().manual_seed(100 + rank)
    fn = _StandInSolve()
    out = []
    for b in range(rounds):
        z = torch.randn(mini, 1, L_TOK, D_EMB, generator=g)
        cols = [fn(input_z=z, write_scale=s, batch_id=b, **_hook_kwargs(root)) for s in scales]
        out.append(torch.stack(cols, dim=1).reshape(-1, 1, L_TOK, D_EMB))
    return out


def _vis_worker(rank, world, port, root, outdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uspace_amd.tools.utils_uvit import DistAccelerator, sample2dir
    from uspace_amd.tools.utils_vis import sample_for_hspace_vis
    acc = DistAccelerator()
    assert acc.num_processes == world and acc.process_index == rank and acc.is_main_process == (rank == 0)
    scales = [-1.0, 0.0, 2.0]
    mini, n_samples = 2, 8                                           # 2 rounds of 2 x 2 latents
    grids = []
    fn = _StandInSolve()
    written = sample_for_hspace_vis(acc, outdir, fn, z_shape=(1, L_TOK, D_EMB), device=torch.device("cpu"), n_samples=n_samples,
                                    mini_batch_size=mini, write_scales=scales, generator=torch.Generator().manual_seed(100 + rank),
                                    attr_name_fn=lambda ith, ds: f"attr{ith}", padding=0,
                                    save_grid_fn=lambda grid, path: grids.append(grid.clone()), **_hook_kwargs(root))
    ok = fn.batch_ids == [0, 0, 0, 1, 1, 1]                          # every rank numbers its rounds itself (batch_id of the reference loop)
    ok = ok and fn.loads == 4                                        # t = 0.10 .. 0.40 edit ("0.00" never does): 4 tables, read once per process
    if rank == 0:
        ok = ok and len(written) == 2 and len(grids) == 2
        exp = [_expected_rows(r, world, mini, scales, root, 2) for r in range(world)]
        for b in range(2):
            rows = torch.cat([exp[r][b] for r in range(world)])      # rank order, then (sample, scale) order inside a rank
            n = rows.shape[0]
            assert n == world * mini * len(scales)
            got = grids[b]                                            # make_grid: nrow = len(scales), padding 0, 1 -> 3 channels
            tiles = got[0].reshape(n // len(scales), L_TOK, len(scales), D_EMB).permute(0, 2, 1, 3).reshape(n, L_TOK, D_EMB)
            ok = ok and torch.allclose(tiles, rows[:, 0], atol=0, rtol=0)
    else:
        ok = ok and written == []
    # sample2dir: 7 samples, 2 per rank and round -> rounds of 4 and 3; the main process writes 0.png .. 6.png in rank order
    saved = []
    cnt = [0]

    def sample_fn(n):
        cnt[0] += 1
        return torch.full((n, 1, 2, 2), float(10 * rank + cnt[0]))
    nw = sample2dir(acc, os.path.join(outdir, "s2d"), 7, 2, sample_fn, save_fn=lambda s, f: saved.append((os.path.basename(f), float(s.mean()))))
    if rank == 0:
        ok = ok and nw == 7 and saved == [("0.png", 1.0), ("1.png", 1.0), ("2.png", 11.0), ("3.png", 11.0), ("4.png", 2.0), ("5.png", 2.0), ("6.png", 12.0)]
    else:
        ok = ok and nw == 0 and saved == []
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_uspace_sweep_and_sample2dir(tmp_path):
    root = str(tmp_path / "tables")
    os.makedirs(root)
    _write_tables(root)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vis_worker, args=(r, 2, port, root, str(tmp_path / "out"), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(2))
    assert got == {0: True, 1: True}


# ------------------------------------------------------------------------------------------- bench.main() end to end, 2 gloo ranks
class _FakeKernels:
    """Stands in for libuspace_hip.so at uspace_amd._hip.lib(): every host-side query goes to the real library (it loads without a
    GPU); the entry points that would LAUNCH kernels act on the CPU buffers behind the pointers instead -- the network evaluation
    is v(x, t) = -x (so the trajectories are independent per row, like the real network), the ODE state update is the real
    arithmetic.  Everything above the C-ABI -- CNF, odeint, the sharding, the gather, bench.py -- is the product's own code."""

    def __init__(self, real):
        self._real = real
        self.forwards = 0

    def __getattr__(self, name):
        return getattr(self._real, name)

    @staticmethod
    def _arr(p, n):
        import ctypes
        addr = p if isinstance(p, int) else (p.value if hasattr(p, "value") else ctypes.cast(p, ctypes.c_void_p).value)
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(addr))

    def uspace_uvit_pack_weights(self, *a):
        return 0

    def uspace_uvit_forward(self, cfg, blob, ws, ws_bytes, io, B, stream):
        c, o = cfg._obj, io._obj
        n = B * c.in_chans * c.img_size * c.img_size
        self._arr(o.out, n)[:] = -self._arr(o.x, n)
        self.forwards += 1
        return 0

    def uspace_ode_combine(self, out, y, ks, coefs, n, numel, stream):
        acc = self._arr(y, numel).astype(np.float32).copy()
        for i in range(n):
            acc += np.float32(coefs[i]) * self._arr(ks[i], numel)
        self._arr(out, numel)[:] = acc
        return 0

    def uspace_ode_error_norm(self, y0, y1, ks, coefs, n, rtol, atol, numel, scratch, result, stream):
        a0, a1 = self._arr(y0, numel), self._arr(y1, numel)
        e = np.zeros(numel, np.float32)
        for i in range(n):
            e += np.float32(coefs[i]) * self._arr(ks[i], numel)
        r = e / (atol + rtol * np.maximum(np.abs(a0), np.abs(a1)))
        out = self._arr(result, 2)
        out[1] = float((r.astype(np.float64) ** 2).sum())
        out[0] = float(np.sqrt(out[1] / numel))
        return 0

    def uspace_prof_all_begin(self, n):
        return 0

    def uspace_prof_dropped(self):
        return 0

    def uspace_prof_all_end(self, keys, ms, max_records, n):
        n._obj.value = 0
        return 0

    def uspace_prof_mfma_peak_clock(self, iters, tf, ghz):
        tf._obj.value, ghz._obj.value = 1.0, 1.0
        return 0

    def uspace_prof_mfma_peak_gemm_op(self, iters, tf, ghz):
        tf._obj.value, ghz._obj.value = 1.0, 1.0
        return 0

    def uspace_prof_hbm_copy(self, nbytes, reps, gb):
        gb._obj.value = 1.0
        return 0


class _CpuEvent:
    def record(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return 1e3 * (other.t - self.t)


class _CpuRuntime:
    backend = "gloo"

    def __init__(self, local_rank):
        self.device = torch.device("cpu")

    def init_group(self, rank, world):
        dist.init_process_group(self.backend, rank=rank, world_size=world)

    def synchronize(self):
        pass

    def event(self):
        return _CpuEvent()

    def device_name(self):
        return "cpu stand-in"


def _bench_main_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import contextlib
    import io

    import bench
    from uspace_amd import _hip
    fake = _FakeKernels(_hip.lib())
    _hip.lib = lambda: fake
    _hip.require_device = lambda t, name="tensor": None
    _hip.stream_ptr = lambda: None
    _hip.sync_current_stream = lambda: None
    bench.RUNTIME = _CpuRuntime
    bench.MODELS["tiny_u"] = dict(name="uvit", embed_dim=64, depth=2, num_heads=1, num_classes=-1)
    bench.CONFIGS[2] = dict(model="tiny_u", batch=3, solver="dopri5", ode_steps=4, hook=False)
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
        f.write(buf.getvalue())
    with open(os.path.join(outdir, f"forwards{rank}.txt"), "w") as f:
        f.write(str(fake.forwards))


def test_bench_main_runs_end_to_end_on_two_gloo_ranks(tmp_path):
    """bench.py's whole main() under the driver's launch form (WORLD_SIZE / RANK / LOCAL_RANK from the environment, 127.0.0.1
    rendezvous), two CPU processes over gloo, kernels stubbed at _hip.lib() only: the process group forms, every rank solves its
    own shard (dopri5 on a fixed grid: 1 + 6 n evaluations), the timed solves gather 2 x B rows in rank order, rank 0 prints ONE
    JSON line whose multi_gpu object names both ranks, their own timings and the gathered row count."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_bench_main_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    out0 = open(tmp_path / "rank0.txt").read().strip().splitlines()
    assert len(out0) == 1 and open(tmp_path / "rank1.txt").read().strip() == ""           # ONE line, from rank 0
    line = json.loads(out0[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2 and line["nfe"] == 1 + 6 * 4
    assert line["config"]["global_batch"] == 6 and line["value"] == pytest.approx(6 * 2 / (line["ms_per_step"] * 2e-3), rel=1e-6)
    mg = line["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["rccl_world_size"] == 2 and mg["gathered_rows"] == mg["gathered_rows_expected"] == 6
    assert [r["rank"] for r in mg["ranks_seen"]] == [0, 1] and [r["local_rank"] for r in mg["ranks_seen"]] == [0, 1]
    assert len({r["pid"] for r in mg["ranks_seen"]}) == 2 and len(mg["per_rank_median_ms"]) == 2 and len(mg["per_rank_wall_s"]) == 2
    assert line["ms_per_step"] * 2e-3 == pytest.approx(max(mg["per_rank_wall_s"]), rel=1e-9)             # MAX over ranks
    # the reference-default sampler's reading (adaptive dopri5, rtol = atol = 1e-5): per-rank step control and the group-controlled form,
    # one entry per rank each; under group control every rank takes the same steps
    for tag in ("dopri5_adaptive", "dopri5_adaptive_norm_group"):
        nfe, acc, rej = line[f"{tag}_nfe"], line[f"{tag}_steps_accepted"], line[f"{tag}_steps_rejected"]
        assert len(nfe) == len(acc) == len(rej) == 2 and line[f"{tag}_images_per_sec"] > 0
        assert all(n == 2 + 6 * (a + r) for n, a, r in zip(nfe, acc, rej)) and min(acc) >= 2      # 2 evaluations pick the first step
    assert len(set(line["dopri5_adaptive_norm_group_nfe"])) == 1
    # every rank ran its own solves: warm-up + 2 timed + (rank 0: the recorded extra solve) + 2 Euler solves of 4 steps + 2 x 2 adaptive solves
    f0, f1 = int(open(tmp_path / "forwards0.txt").read()), int(open(tmp_path / "forwards1.txt").read())
    adaptive1 = 2 * line["dopri5_adaptive_nfe"][1] + 2 * line["dopri5_adaptive_norm_group_nfe"][1]
    adaptive0 = 2 * line["dopri5_adaptive_nfe"][0] + 2 * line["dopri5_adaptive_norm_group_nfe"][0]
    assert f1 == 3 * 25 + 2 * 4 + adaptive1 and f0 == 4 * 25 + 2 * 4 + adaptive0
