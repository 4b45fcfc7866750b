"""The reference's sampling loops (tools/utils_uvit.py:264-277 sample2dir, tools/utils_vis.py:138-255 sample_for_hspace_vis)
over two gloo processes: rank-ordered gather, main process writes, the batched write_scales sweep gives the image of the
sequential sweep."""
import glob
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from PIL import Image


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_decode(z, scale):
    # stand-in for VAE-decode(score_model.decode(z, write_scale=scale)): [B,4,8,8] -> [B,3,8,8] in [0,1]
    return torch.sigmoid(z[:, :3] + scale * z[:, 3:4])


def _worker(rank, world, port, out_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uspace_amd.tools.utils_uvit import DistAccelerator, sample2dir
    from uspace_amd.tools.utils_vis import sample_for_hspace_vis
    acc = DistAccelerator()
    assert acc.num_processes == world and acc.is_main_process == (rank == 0)

    # ---- sample2dir: 7 samples, mini batch 2 per process -> rounds of 4, 3
    calls = []

    def sample_fn(n):
        calls.append(n)
        k = len(calls) - 1
        return torch.stack([torch.full((3, 4, 4), (100 * k + 10 * rank + i) / 255.0) for i in range(n)])

    n_written = sample2dir(acc, os.path.join(out_dir, "s2d"), 7, 2, sample_fn, unpreprocess_fn=lambda v: v)
    ok = calls == [2, 2] and (n_written == 7 if rank == 0 else n_written == 0)

    # ---- sample_for_hspace_vis: sequential sweep vs one batched sweep, same latents (seeded per rank)
    scales = [-1.0, 0.0, 2.0]
    kw = dict(dissect_name="write_attr", ith_attr=3, seed=5, dataset_name="celeba256")
    files = {}
    for mode in ("seq", "sweep"):
        g = torch.Generator().manual_seed(100 + rank)
        seen = []

        def one(input_z, write_scale, batch_id, **k):
            seen.append((batch_id, float(write_scale)))
            return _fake_decode(input_z, write_scale)

        def sweep(input_z, write_scales, batch_id, **k):
            seen.append((batch_id, tuple(write_scales)))
            return torch.stack([_fake_decode(input_z, s) for s in write_scales])

        files[mode] = sample_for_hspace_vis(acc, os.path.join(out_dir, mode), one, z_shape=(4, 8, 8), device="cpu", n_samples=8,
                                            mini_batch_size=2, write_scales=scales, sweep_fn=sweep if mode == "sweep" else None,
                                            generator=g, **kw)
        ok = ok and (len(seen) == (6 if mode == "seq" else 2))       # 2 rounds x 3 scales, or 2 rounds x 1 sweep
    if rank == 0:
        ok = ok and len(files["seq"]) == 2 and len(files["sweep"]) == 2
    with pytest.raises(NotImplementedError):
        sample_for_hspace_vis(acc, out_dir, None, z_shape=(4, 8, 8), n_samples=2, mini_batch_size=1, write_scales=scales,
                              dissect_name="bogus")
    with pytest.raises(ValueError):     # unknown dataset name, as tools/utils_attr.py:104-111
        sample_for_hspace_vis(acc, out_dir, lambda **k: torch.zeros(1, 3, 8, 8), z_shape=(4, 8, 8), device="cpu", n_samples=1,
                              mini_batch_size=1, write_scales=[0.0], dissect_name="write_attr", ith_attr=1, dataset_name="x")
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_loops_over_two_processes(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
    # sample2dir: files 0..6 in gather order: round k -> rank 0's two samples, then rank 1's
    vals = [int(np.asarray(Image.open(tmp_path / "s2d" / f"{i}.png"))[0, 0, 0]) for i in range(7)]
    assert vals == [0, 1, 10, 11, 100, 101, 110] and not (tmp_path / "s2d" / "7.png").exists()
    # the batched sweep writes the same grids as the sequential sweep
    seq = sorted(glob.glob(str(tmp_path / "seq" / "*.png")))
    swp = sorted(glob.glob(str(tmp_path / "sweep" / "*.png")))
    assert len(seq) == len(swp) == 2 and all("_seed5_" in f and f.endswith("_Bags_Under_Eyes-1.00_0.00_2.00.png") for f in seq)
    for a, b in zip(seq, swp):
        ia, ib = np.asarray(Image.open(a)), np.asarray(Image.open(b))
        # 4 rows (2 processes x 2 latents) of 3 scales, 8x8 images, padding 8 (tools/utils_vis.py:18)
        assert ia.shape == (4 * 16 + 8, 3 * 16 + 8, 3)
        np.testing.assert_array_equal(ia, ib)
        assert (ia[0] == 255).all() and (ia[:, 0] == 255).all()          # pad_value 1.0 border


def test_make_grid_layout():
    from uspace_amd.tools.utils_vis import make_grid
    imgs = torch.arange(5, dtype=torch.float32).view(5, 1, 1, 1).expand(5, 3, 2, 2)
    g = make_grid(imgs, nrow=3, padding=1, pad_value=9.0)
    assert g.shape == (3, 2 * 3 + 1, 3 * 3 + 1)
    want = np.full((7, 10), 9.0, np.float32)
    for k in range(5):
        y, x = divmod(k, 3)
        want[1 + 3 * y:3 + 3 * y, 1 + 3 * x:3 + 3 * x] = k
    np.testing.assert_array_equal(g[0].numpy(), want)


class _OneProc:
    num_processes = 1
    is_main_process = True

    def gather(self, t):
        return t


def test_hspace_vis_reference_calling_convention_with_fixed_latents(tmp_path):
    """every reference config calls with has_attr=True and fixed_z_path=<root>, which means the latents are in
    ``<root>.npz`` under "latent" (tools/utils_vis.py:25-35); the grid is written like ToPILImage: mul(255).byte() = truncation
    (tools/utils_vis.py:249-251); file names carry the attribute names (tools/utils_attr.py:113-121)."""
    from uspace_amd.tools.utils_attr import get_attr_name_from_attr_id
    from uspace_amd.tools.utils_vis import load_z_from_dir, sample_for_hspace_vis
    lat = np.random.default_rng(3).standard_normal((3, 4, 8, 8)).astype(np.float32)
    root = str(tmp_path / "latents")
    np.savez(root + ".npz", latent=lat, attr=np.zeros((3, 40), np.int64))
    np.save(str(tmp_path / "plain.npy"), lat)
    np.testing.assert_array_equal(load_z_from_dir(root, True, "cpu").numpy(), lat)
    np.testing.assert_array_equal(load_z_from_dir(root + ".npz", True, "cpu").numpy(), lat)      # documented leniency
    np.testing.assert_array_equal(load_z_from_dir(str(tmp_path / "plain.npy"), False, "cpu").numpy(), lat)
    with pytest.raises(FileNotFoundError):
        load_z_from_dir(str(tmp_path / "nothing"), True, "cpu")
    seen = []

    def one(input_z, write_scale, batch_id, **k):
        seen.append(input_z.clone())
        return torch.full((input_z.shape[0], 3, 4, 4), 0.5)      # 127.5 -> 127 by truncation, 128 by rounding

    kw = dict(dissect_name="write_attr", ith_attr="31_39_20", seed=None, dataset_name="celeba256", has_attr=True)
    files = sample_for_hspace_vis(_OneProc(), str(tmp_path / "o"), one, z_shape=(4, 8, 8), device="cpu", n_samples=99,
                                  mini_batch_size=2, write_scales=[0.0, 1.0], fixed_z_path=root, **kw)
    assert len(files) == 2                                                 # n_samples overridden by len(latents) = 3
    np.testing.assert_array_equal(torch.cat(seen[::2]).numpy(), lat)       # rows in order, one slice per round
    assert files[0].endswith("_seedNone_0_Smiling_Young_Male0.00_1.00.png")
    img = np.asarray(Image.open(files[0]))
    assert img.shape == (2 * 12 + 8, 2 * 12 + 8, 3) and img[8, 8, 0] == 127 and img[0, 0, 0] == 255
    with pytest.raises(KeyError):        # the reference indexes kwargs["has_attr"]
        sample_for_hspace_vis(_OneProc(), str(tmp_path / "o"), one, z_shape=(4, 8, 8), device="cpu", mini_batch_size=2,
                              write_scales=[0.0], fixed_z_path=root, dissect_name="write_attr", ith_attr=1, dataset_name="ffhq")
    assert get_attr_name_from_attr_id(1, "ffhq256") == "smile" and get_attr_name_from_attr_id("0_39", "celebamask") == "5_o_Clock_Shadow_Young"
    with pytest.raises(ValueError):
        get_attr_name_from_attr_id(1.5, "ffhq")
