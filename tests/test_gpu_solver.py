"""ODE driver on the GPU: HIP state kernels + CNF entry points against the numpy oracle solver and the
committed Euler-20 trajectory of the reference network (BASELINE.json configs[0])."""
import os

import numpy as np
import pytest
import torch

from oracle import odeint_oracle as OO
from oracle import uvit_oracle as O
from tests.util import load_sd, rel_l2

pytestmark = pytest.mark.gpu

TINY = dict(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4,
            qkv_bias=False, mlp_time_embed=False)


def _field_np(t, y):
    return (-0.8 * y + np.float32(np.sin(3.0 * t)) + 0.3 * np.tanh(y)).astype(np.float32)


def _field_torch(t, y):
    return -0.8 * y + float(np.sin(3.0 * t)) + 0.3 * torch.tanh(y)


@pytest.mark.parametrize("method,kw", [
    ("euler", dict(step_size=0.01)), ("midpoint", dict(step_size=0.02)), ("rk4", dict(step_size=0.05)),
    ("euler", dict(n_steps=50)), ("dopri5", dict(n_steps=50)), ("dopri5", {}), ("bosh3", {}), ("adaptive_heun", dict(rtol=1e-3, atol=1e-3)),
])
@pytest.mark.parametrize("span", [(0.0, 1.0), (1.0, 0.0), (0.0, 0.4)])
def test_hip_odeint_matches_oracle_solver(method, kw, span):
    from uspace_amd.odeint import Stats, odeint
    rng = np.random.default_rng(1)
    y0 = rng.standard_normal((3, 4, 8, 8)).astype(np.float32)
    cnt = {}
    ref = OO.solve(_field_np, y0, span[0], span[1], method=method, counters=cnt, **kw)
    st = Stats()
    got = odeint(_field_torch, torch.from_numpy(y0).cuda(), span[0], span[1], method=method, stats=st, **kw)
    assert rel_l2(got.cpu().numpy(), ref) < 2e-5
    assert st.nfe == cnt["nfe"]
    if method == "dopri5" and "n_steps" in kw:
        assert st.nfe == 301            # "dopri5-50" of BASELINE.md
    if method == "euler" and "n_steps" in kw:
        assert st.nfe == 50


def _solver_kwargs(**over):
    sk = dict(solver="fixed", solver_fix="euler", solver_fix_step=0.05, solver_adaptive="dopri5",
              solver_adaptive_prec=0.01)
    sk.update(over)
    return sk


def test_cnf_euler20_matches_reference_trajectory(golden_dir):
    """configs[0]: S-deep16, B=4, 20 Euler steps; end state vs the reference network's own trajectory."""
    from uspace_amd.flow_matching import CNF
    from uspace_amd.tools.utils_uvit import get_nnet
    z = np.load(os.path.join(golden_dir, "euler20_S_u.npz"))
    torch.manual_seed(1234)
    net = get_nnet("uvit", img_size=32, patch_size=2, in_chans=4, embed_dim=512, depth=16, num_heads=8, mlp_ratio=4,
                   qkv_bias=False, mlp_time_embed=False, num_classes=-1).cuda().eval()
    cnf = CNF(net)
    x1 = cnf.decode(torch.from_numpy(z["z"]).cuda(), None, dissect_name="none", edit_loc=None,
                    solver_kwargs=_solver_kwargs(solver_fix_step=0.05))
    assert cnf.last_stats.nfe == 20
    r = rel_l2(x1.cpu().numpy(), z["x1"])
    assert r < 5e-3, r                   # 20-step trajectory gate


def test_cnf_modes_on_tiny_net_match_oracle(golden_dir):
    from uspace_amd.flow_matching import CNF
    from uspace_amd.tools.utils_uvit import get_nnet
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = get_nnet("uvit", num_classes=-1, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    spec = O.UViTSpec(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1)
    f_ref = lambda t, y: O.uvit_forward(spec, sd, y, np.float32(t), edit_loc=None)
    x0 = torch.from_numpy(z["x"]).cuda()

    # with pytest.raises(KeyError): the reference's decode() indexes kwargs["solver_kwargs"]
    with pytest.raises(KeyError):
        cnf.decode(x0, None)

    # non-dissection default: adaptive dopri5 rtol=atol=1e-5 (flow_matching.py:77-84)
    got = cnf.decode(x0, None, edit_loc=None, solver_kwargs=_solver_kwargs(solver="adaptive"))
    cnt = {}
    ref = OO.solve(f_ref, z["x"], 0.0, 1.0, method="dopri5", counters=cnt)
    assert rel_l2(got.cpu().numpy(), ref) < 1e-2
    assert abs(cnf.last_stats.nfe - cnt["nfe"]) <= 12      # step sequences may differ by a step or two in bf16

    # fixadp: Euler to t_edit, dopri5 after (flow_matching.py:153-180)
    got = cnf.decode(x0, None, edit_loc=None, dissect_name="none", t_edit=0.4,
                     solver_kwargs=_solver_kwargs(solver="fixadp", solver_fix_step=0.1))
    mid = OO.solve(f_ref, z["x"], 0.0, 0.4, method="euler", step_size=0.1)
    ref = OO.solve(f_ref, mid, 0.4, 1.0, method="dopri5")
    assert rel_l2(got.cpu().numpy(), ref) < 1e-2

    # encode (1 -> 0, fixed solver) then decode back ~ identity (dissect_lfm.py:171-178 "vis_reversible")
    kw = dict(edit_loc=None, dissect_name="none", solver_kwargs=_solver_kwargs(solver_fix="rk4", solver_fix_step=0.05))
    zz = cnf.encode(x0, None, **kw)
    back = cnf.decode(zz, None, **kw)
    assert rel_l2(back.cpu().numpy(), z["x"]) < 5e-3
    ref = OO.solve(f_ref, z["x"], 1.0, 0.0, method="rk4", step_size=0.05)
    assert rel_l2(zz.cpu().numpy(), ref) < 1e-2

    with pytest.raises(NotImplementedError):
        cnf.decode(x0, None, edit_loc=None, dissect_name="none", solver_kwargs=_solver_kwargs(solver="magic"))
    assert CNF.sample_ode is CNF.decode


def test_cnf_t2i_sets_direction_and_edits(golden_dir):
    from uspace_amd.flow_matching_t2i import CNF
    from uspace_amd.tools.utils_uvit import get_nnet
    z, sd = load_sd(golden_dir, "tiny_t2i.npz")
    net = get_nnet("uvit_t2i", clip_dim=64, num_clip_token=77, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    x0, ctx = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["ctx"]).cuda()
    ids = [np.array([3, 5]), np.array([], dtype=np.int64), np.array([0, 76])]
    base = dict(dissect_name="p2p", t_edit=0.5, block_id="all", target_context_ids=ids,
                token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=8.0),
                solver_kwargs=_solver_kwargs(solver_fix_step=0.1))
    plain = cnf.decode(x0, ctx, dissect_name="p2p", t_edit=0.5, block_id="all", target_context_ids=ids,
                       token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=1.0),
                       solver_kwargs=_solver_kwargs(solver_fix_step=0.1))
    edited = cnf.decode(x0, ctx, **base)
    assert cnf.last_stats.nfe == 10
    assert rel_l2(edited.cpu().numpy(), plain.cpu().numpy()) > 1e-4      # decode edits while t <= t_edit
    enc_a = cnf.encode(x0, ctx, **base)                                    # encode never edits
    enc_b = cnf.encode(x0, ctx, dissect_name="p2p", t_edit=0.5, block_id="all", target_context_ids=ids,
                       token_kwargs=dict(token_dissect="p2p_rescale", p2p_multiplier=1.0),
                       solver_kwargs=_solver_kwargs(solver_fix_step=0.1))
    assert torch.equal(enc_a, enc_b)
    # oracle trajectory with the same edit
    spec = O.UViTSpec(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, t2i=True,
                      clip_dim=64, num_clip_token=77)
    okw = dict(base, fm_direction="decode")
    okw.pop("solver_kwargs")
    f_ref = lambda t, y: O.uvit_forward(spec, sd, y, np.float32(t), context=z["ctx"], **okw)
    ref = OO.solve(f_ref, z["x"], 0.0, 1.0, method="euler", step_size=0.1)
    assert rel_l2(edited.cpu().numpy(), ref) < 1e-2


def test_batched_write_scales_sweep_equals_sequential_solves(golden_dir):
    """tools/utils_vis.py:189-198 runs one full solve per write_scale; decode_write_scales() runs them as one
    solve with a per-row hook scale.  Same results (up to GEMM tile-shape summation grouping) at every hook
    location, and scale 0 rows equal the un-hooked solve."""
    import os
    import tempfile
    from uspace_amd.flow_matching import CNF
    from uspace_amd.tools.utils_uvit import get_nnet
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    hz = np.load(os.path.join(golden_dir, "hooks_u.npz"))
    net = get_nnet("uvit", num_classes=-1, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    x0 = torch.from_numpy(z["x"]).cuda()
    spec = O.UViTSpec(img_size=16, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1)
    scales = [-2.0, -0.5, 0.0, 1.0, 3.0]
    sk = _solver_kwargs(solver_fix_step=0.1)
    with tempfile.TemporaryDirectory() as d:
        for loc, table in (("head", hz["img_attr"]), ("tail", hz["img_attr"]), ("mid", hz["tok_attr"])):
            root = os.path.join(d, loc)
            os.makedirs(root)
            for k in range(0, 11):
                np.save(os.path.join(root, f"delta_{k / 10:.2f}.npy"), table)
            kw = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=root,
                      edit_loc=loc, ith_attr="1_3", solver_kwargs=sk)
            seq = torch.stack([cnf.decode(x0, None, write_scale=s, **kw) for s in scales])
            bat = cnf.decode_write_scales(x0, None, scales, **kw)
            assert bat.shape == seq.shape
            assert rel_l2(bat.cpu().numpy(), seq.cpu().numpy()) < 2e-3, loc
            plain = cnf.decode(x0, None, dissect_name="none", edit_loc=None, solver_kwargs=sk)
            assert rel_l2(bat[2].cpu().numpy(), plain.cpu().numpy()) < 2e-3
            assert rel_l2(bat[4].cpu().numpy(), plain.cpu().numpy()) > 1e-3
            # ... and both equal the ORACLE's hooked trajectory, one sequential solve per scale as the reference loop
            # does (tools/utils_vis.py:189-198 around libs/dissection.py:115-186): the batched form is compared with an
            # independent statement of the hook, not only with the product's own sequential solves
            okw = {k: v for k, v in kw.items() if k != "solver_kwargs"}
            for i, sc in enumerate(scales):
                f_ref = lambda t, y, sc=sc: O.uvit_forward(spec, sd, y, np.float32(t), write_scale=sc, **okw)
                ref = OO.solve(f_ref, z["x"], 0.0, 1.0, method="euler", step_size=0.1)
                assert rel_l2(bat[i].cpu().numpy(), ref) < 1e-2, (loc, sc)
                assert rel_l2(seq[i].cpu().numpy(), ref) < 1e-2, (loc, sc)
            ref_plain = OO.solve(lambda t, y: O.uvit_forward(spec, sd, y, np.float32(t), edit_loc=None), z["x"], 0.0, 1.0,
                                 method="euler", step_size=0.1)
            edit_size = rel_l2(OO.solve(lambda t, y: O.uvit_forward(spec, sd, y, np.float32(t), write_scale=3.0, **okw),
                                        z["x"], 0.0, 1.0, method="euler", step_size=0.1), ref_plain)
            assert edit_size > 1e-3, (loc, edit_size)      # the comparison above is not vacuous: the hook moves the result
    with pytest.raises(ValueError):
        net(x0, torch.tensor(0.2, device="cuda").expand(3), None, dissect_task="uspace_uvit", dissect_name="write_attr",
            t_edit=0.4, write_path_root="/nonexistent", edit_loc="head", ith_attr=1, write_scale=[1.0, 2.0])


def test_device_resident_attribute_directions(golden_dir):
    """tools/utils_attr.py:124-207 on the GPU: activations streamed batch by batch into running sums,
    against the reference's own result (golden) and the oracle; then the files drive the write hook."""
    import os
    import tempfile
    from oracle import attr_oracle as A
    from uspace_amd.tools.utils_attr import DirectionAccumulator
    z = np.load(os.path.join(golden_dir, "attr_directions.npz"))
    for tag, adim in (("celeba", 40), ("ffhq", 11)):
        attrs, feats, want = z[f"{tag}_attrs"], z[f"{tag}_feats"], z[f"{tag}_delta"]
        acc = DirectionAccumulator(adim)
        N, T = feats.shape[:2]
        for lo in range(0, N, 10):                     # ragged last batch (37 = 10+10+10+7)
            for ti in range(T):
                acc.update(f"0.{ti}0", torch.from_numpy(feats[lo:lo + 10, ti]).cuda(), attrs[lo:lo + 10])
        with tempfile.TemporaryDirectory() as d:
            ts = acc.finalize(d)
            assert ts == ["0.00", "0.10", "0.20"]
            for ti, t in enumerate(ts):
                got = np.load(os.path.join(d, f"delta_{t}.npy"))
                assert got.shape == want[:, ti].shape and got.dtype == np.float32
                np.testing.assert_allclose(got, want[:, ti], rtol=2e-5, atol=2e-6, equal_nan=True)
                np.testing.assert_allclose(got, A.delta_directions(attrs, feats[:, ti]), rtol=2e-5, atol=2e-6, equal_nan=True)
    with pytest.raises(ValueError):
        DirectionAccumulator(7)


def test_read_hook_feeds_accumulator_then_write_hook_uses_it(golden_dir):
    """README workflow on the device: encode with the read hook (mid block) accumulating directions, finalize,
    then decode with the write hook reading those files."""
    import os
    import tempfile
    from uspace_amd.flow_matching import CNF
    from uspace_amd.tools.utils_attr import DirectionAccumulator
    from uspace_amd.tools.utils_uvit import get_nnet
    z, sd = load_sd(golden_dir, "tiny_u.npz")
    net = get_nnet("uvit", num_classes=-1, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    rng = np.random.default_rng(0)
    attrs = (rng.random((3, 40)) < 0.5).astype(np.int64)
    attrs[:, 5] = [1, 0, 1]
    x0 = torch.from_numpy(z["x"]).cuda()
    acc = DirectionAccumulator(40)
    sk = _solver_kwargs(solver_fix_step=0.25)
    cnf.encode(x0, None, dissect_task="uspace_uvit", dissect_name="read", edit_loc="mid", read_path_root="/unused",
               batch_id=0, direction_accumulator=acc, attrs=attrs, solver_kwargs=sk)
    with tempfile.TemporaryDirectory() as d:
        ts = acc.finalize(d)
        assert ts == ["0.25", "0.50", "0.75", "1.00"]
        delta = np.load(os.path.join(d, "delta_0.25.npy"))
        assert delta.shape == (40, 65, 64) and np.isfinite(delta[5]).all()
        kw = dict(dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.5, write_path_root=d, edit_loc="mid",
                  ith_attr=5, solver_kwargs=sk)
        edited = cnf.decode(x0, None, write_scale=2.0, **kw)
        plain = cnf.decode(x0, None, write_scale=0.0, **kw)
        assert bool(torch.isfinite(edited).all()) and rel_l2(edited.cpu().numpy(), plain.cpu().numpy()) > 1e-4


def test_pca_directions_on_device_match_reference_golden(golden_dir, tmp_path):
    """tools/utils_pca.py:13-50 / tools/utils_vis.py:80-118: principal directions of tapped activations (Gram-matrix
    route on the device) against the reference function's output, from tensors, from the accumulator the read hook can
    feed, and from ``{batch_id}_{t}.npy`` files; the result is what the write_pca hook loads."""
    from uspace_amd.tools.utils_pca import PcaAccumulator, extract_hspace_feat_unet_by_pca, pca_components
    z = np.load(os.path.join(golden_dir, "pca_components.npz"))
    feats, ref, n = z["feats"], z["components"], int(z["n_components"])

    def same_up_to_sign(got):
        assert got.shape == ref.shape
        for g, r in zip(got.reshape(n, -1), ref.reshape(n, -1)):
            assert abs(abs(float(np.dot(g, r))) - 1.0) < 1e-4 and np.abs(g * np.sign(np.dot(g, r)) - r).max() < 2e-4
    x = torch.from_numpy(feats).cuda()
    got = pca_components(x, n)
    same_up_to_sign(got.cpu().numpy())
    gram = got.reshape(n, -1) @ got.reshape(n, -1).t()
    assert float((gram - torch.eye(n, device="cuda")).abs().max()) < 1e-4          # orthonormal
    acc = PcaAccumulator()
    for lo in range(0, len(feats), 16):
        acc.update("0.30", x[lo:lo + 16])
    assert acc.finalize(str(tmp_path), n) == ["0.30"]
    same_up_to_sign(np.load(tmp_path / f"pca{n}_0.30.npy"))
    for b, lo in enumerate(range(0, len(feats), 16)):
        np.save(tmp_path / f"{b}_0.50.npy", feats[lo:lo + 16])
    assert "0.50" in extract_hspace_feat_unet_by_pca(str(tmp_path), n_components=n, batch_num=3)
    same_up_to_sign(np.load(tmp_path / f"pca{n}_0.50.npy"))
    with pytest.raises(ValueError):
        pca_components(x, len(feats) + 1)


def test_group_controlled_error_norm_on_the_device():
    """HipStateOps(group=...) all-reduces (sum of squares, count) over RCCL instead of reading the local RMS: with one
    rank the solve must take exactly the steps of the plain solve (the 2-rank behaviour is covered on CPU by
    tests/test_sharded_sampling.py); CNF.norm_group routes it."""
    import torch.distributed as dist

    from uspace_amd.odeint import HipStateOps, Stats, odeint
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        rng = np.random.default_rng(3)
        y0 = torch.from_numpy(rng.standard_normal((5, 4, 8, 8)).astype(np.float32)).cuda()
        s0, s1 = Stats(), Stats()
        a = odeint(_field_torch, y0, 0.0, 1.0, method="dopri5", stats=s0)
        b = odeint(_field_torch, y0, 0.0, 1.0, method="dopri5", stats=s1, ops=HipStateOps(y0, group=True))
        assert (s0.nfe, s0.accepted, s0.rejected) == (s1.nfe, s1.accepted, s1.rejected)
        assert float((a - b).abs().max()) < 1e-5
        # an empty shard still walks the control loop (it owes the group its share of every norm)
        s2 = Stats()
        e = odeint(_field_torch, y0[:0], 0.0, 1.0, method="dopri5", stats=s2, ops=HipStateOps(y0, group=True))
        assert e.shape[0] == 0 and s2.nfe > 0
    finally:
        if created:
            dist.destroy_process_group()


def test_pca_small_variance_components_survive_the_gram_matrix():
    """Directions whose singular value is 1e-5 of the largest: an fp32 Gram matrix (eigenvalue ratio 1e-10, below fp32's 1e-7)
    loses them; the fp64-matrix-core Gram of csrc/pca.hip keeps them.  Compared with the fp64-SVD oracle on the
    same fp32 data, up to sign; odd feature count exercises the padding and the feature tail."""
    from oracle import pca_oracle as PO
    from uspace_amd.tools.utils_pca import pca_components
    rng = np.random.default_rng(5)
    for shape in ((4, 8, 8), (3, 7, 5)):
        N, F = 48, int(np.prod(shape))
        sv = np.array([10.0, 3.0, 1.0, 1e-2, 1e-4], np.float64)
        u, _ = np.linalg.qr(rng.standard_normal((N, len(sv))))
        v, _ = np.linalg.qr(rng.standard_normal((F, len(sv))))
        x = ((u * sv) @ v.T + 0.5).astype(np.float32).reshape((N,) + shape)       # + a mean the centring must remove
        want = PO.pca_components(x, len(sv))
        got = pca_components(torch.from_numpy(x).cuda(), len(sv)).cpu().numpy()
        assert got.shape == want.shape
        for i in range(len(sv)):
            a, b = got[i].ravel().astype(np.float64), want[i].ravel().astype(np.float64)
            assert abs(np.linalg.norm(a) - 1.0) < 1e-5
            cos = float(a @ b)
            assert cos > 1.0 - 2e-3, (shape, i, cos)            # same direction AND same sign convention
        # orthonormal set
        gmat = got.reshape(len(sv), -1).astype(np.float64)
        assert np.abs(gmat @ gmat.T - np.eye(len(sv))).max() < 1e-3


# --------------------------------------------------------------------------- hooked solves against the REFERENCE's own trajectories
def test_cnf_hooked_uncond_solves_match_reference_trajectories(golden_dir, tmp_path, monkeypatch):
    """hooked_traj.npz (a): the reference network + the reference's dissect_helper_uvit over 100 Euler steps (make_golden.py::
    make_hooked_traj).  The same solve through CNF.decode: end state within the trajectory tolerance, and the product edits in exactly
    the evaluations the reference did -- 40 of 100: grid points k * 0.01 (fp32) with "0.01" <= "{t:.2f}" <= "0.40", never "0.00"."""
    from tests.test_oracle_golden import HOOKED_MID, HOOKED_U, write_hooked_tables
    from uspace_amd import _hip
    from uspace_amd.flow_matching import CNF
    from uspace_amd.libs import dissection
    from uspace_amd.tools.utils_uvit import get_nnet
    z = np.load(os.path.join(golden_dir, "hooked_traj.npz"))
    zt, sd = load_sd(golden_dir, "tiny_u.npz")
    net = get_nnet("uvit", num_classes=-1, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    write_hooked_tables(str(tmp_path))
    mid_dir = tmp_path / "mid"
    mid_dir.mkdir()
    write_hooked_tables(str(mid_dir), (5, 65, 64))
    planned, adds = [], []
    real_plan, real_add = dissection.plan_uspace_hook, _hip.add_broadcast
    monkeypatch.setattr(dissection, "plan_uspace_hook", lambda digit, kw: (lambda p: (planned.append((digit, None if p is None else os.path.basename(p.path))), p)[1])(real_plan(digit, kw)))
    monkeypatch.setattr(_hip, "add_broadcast", lambda *a, **k: (adds.append(1), real_add(*a, **k))[1])
    x0 = torch.from_numpy(zt["x"]).cuda()
    import json
    for tag, kw in HOOKED_U + (("mid", HOOKED_MID),):
        planned.clear(), adds.clear()
        root = str(mid_dir if tag == "mid" else tmp_path)
        x1 = cnf.decode(x0, None, dissect_task="uspace_uvit", dissect_name="write_attr", t_edit=0.4, write_path_root=root,
                        solver_kwargs=_solver_kwargs(solver_fix_step=0.01), **kw)
        assert cnf.last_stats.nfe == 100 and len(planned) == 100
        fired = [f for _d, f in planned if f is not None]
        # (the mid-block add happens inside the C call: 40 plans, no separate add launch; its fixture is documented semantics)
        assert fired == [f"delta_{k / 100:.2f}.npy" for k in range(1, 41)] and len(adds) == (0 if tag == "mid" else 40)
        if tag != "mid":
            assert fired == json.loads(bytes(z[f"u_{tag}_files"]).decode())
        assert [d for d, f in planned if f is None][:1] == ["0.00"] and planned[30] == ("0.30", "delta_0.30.npy") and planned[41][1] is None
        r = rel_l2(x1.cpu().numpy(), z[f"u_{tag}_x1"])
        assert r < 5e-3, (tag, r)
        shift = rel_l2(z["u_plain_x1"], z[f"u_{tag}_x1"])
        assert r < 0.25 * shift, (tag, r, shift)          # far closer to the hooked reference than the unhooked solve is


def test_cnf_t2i_encode_then_decode_matches_reference_trajectories(golden_dir, monkeypatch):
    """hooked_traj.npz (b): reference U-ViT T2I + editing_attention_map_vit (p2p_rescale on block 1, multiplier 40): encode 1 -> 0
    never edits (fm_direction "encode"), decode 0 -> 1 of the encoded latent edits in 11 of its 20 evaluations ("0.00" ... "0.50")."""
    from tests.test_oracle_golden import HOOKED_T, HOOKED_T_IDS
    from uspace_amd.flow_matching_t2i import CNF
    from uspace_amd.tools import utils_t2i
    from uspace_amd.tools.utils_uvit import get_nnet
    z = np.load(os.path.join(golden_dir, "hooked_traj.npz"))
    zt, sd = load_sd(golden_dir, "tiny_t2i.npz")
    net = get_nnet("uvit_t2i", clip_dim=64, num_clip_token=77, **TINY)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.cuda().eval()
    cnf = CNF(net)
    tables = []
    real = utils_t2i.key_scale_table
    monkeypatch.setattr(utils_t2i, "key_scale_table", lambda *a, **k: (lambda t: (tables.append(None if t is None else [int(b) for b in range(t.shape[0]) if (t[b] != 1).any()]), t)[1])(real(*a, **k)))
    x0, ctx = torch.from_numpy(zt["x"]).cuda(), torch.from_numpy(zt["ctx"]).cuda()
    kw = dict(HOOKED_T, target_context_ids=HOOKED_T_IDS, solver_kwargs=_solver_kwargs(solver_fix_step=0.05))
    z_enc = cnf.encode(x0, ctx, **dict(kw))
    assert cnf.last_stats.nfe == 20 and tables == [None] * 20
    tables.clear()
    x_dec = cnf.decode(torch.from_numpy(z["t_z_enc"]).cuda(), ctx, **dict(kw))
    assert cnf.last_stats.nfe == 20
    assert [t is not None for t in tables] == [True] * 11 + [False] * 9 and all(t == [1] for t in tables[:11])      # block 1 only
    assert [0, 11] == z["t_edit_calls"].tolist()
    assert rel_l2(z_enc.cpu().numpy(), z["t_z_enc"]) < 5e-3
    r = rel_l2(x_dec.cpu().numpy(), z["t_x_dec"])
    assert r < 5e-3, r


def test_cnf_euler50_L_t_matches_reference_trajectory(golden_dir):
    """BASELINE configs[2] and configs[3] end to end at B = 2: U-ViT-L T2I and U-ViT-S-deep16 T2I (weights rebuilt from the seed), 50 Euler steps."""
    from uspace_amd.flow_matching_t2i import CNF
    from uspace_amd.tools.utils_uvit import get_nnet
    z = np.load(os.path.join(golden_dir, "hooked_traj.npz"))
    torch.manual_seed(1234)
    net = get_nnet("uvit_t2i", img_size=32, patch_size=2, in_chans=4, embed_dim=1024, depth=20, num_heads=16, mlp_ratio=4, qkv_bias=False,
                   mlp_time_embed=False, clip_dim=768, num_clip_token=77)
    # (the seeded init reproduces the reference's weights bit for bit on the host that generated the fixture --
    # tests/test_host_logic.py pins that by sha256 --; another CPU's erfinv may differ in the last bit, so no hash here)
    cnf = CNF(net.cuda().eval())
    x1 = cnf.decode(torch.from_numpy(z["Lt_z"]).cuda(), torch.from_numpy(z["Lt_ctx"]).cuda(), dissect_name="none",
                    solver_kwargs=_solver_kwargs(solver_fix_step=0.02))
    assert cnf.last_stats.nfe == 50
    r = rel_l2(x1.cpu().numpy(), z["Lt_x1_euler50"])
    assert r < 5e-3, r
    # BASELINE configs[3]: U-ViT-S-deep16 T2I, same latents and context
    del cnf, net
    torch.manual_seed(1234)
    net_s = get_nnet("uvit_t2i", img_size=32, patch_size=2, in_chans=4, embed_dim=512, depth=16, num_heads=8, mlp_ratio=4, qkv_bias=False,
                     mlp_time_embed=False, clip_dim=768, num_clip_token=77)
    cnf_s = CNF(net_s.cuda().eval())
    x1s = cnf_s.decode(torch.from_numpy(z["Lt_z"]).cuda(), torch.from_numpy(z["Lt_ctx"]).cuda(), dissect_name="none",
                       solver_kwargs=_solver_kwargs(solver_fix_step=0.02))
    rs = rel_l2(x1s.cpu().numpy(), z["St_x1_euler50"])
    assert cnf_s.last_stats.nfe == 50 and rs < 5e-3, rs
