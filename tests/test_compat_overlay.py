"""compat/: the reference's scripts must find this package under the reference's own import names
(flow_matching.CNF dissect_lfm.py:2, flow_matching_t2i.CNF dissect_lfm_t2i.py:11, tools.utils_uvit.get_nnet
dissect_lfm.py:67 -> libs.uvit / libs.uvit_t2i tools/utils_uvit.py:27-41) while everything else keeps coming from the
reference tree.  A miniature reference tree stands in for /root/reference (which does not exist on the GPU box)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "compat")

SCRIPT = """
import json, sys
from flow_matching import CNF
import tools.utils_uvit as utils_uvit
from tools.utils_uvit import get_nnet
from tools.utils_misc import MARK as misc_mark
import libs.autoencoder
from flow_matching_t2i import CNF as CNF_T
cfg = dict(img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4, qkv_bias=False,
           mlp_time_embed=False, num_classes=-1)
net = get_nnet("uvit", **cfg)
net_t = get_nnet("uvit_t2i", clip_dim=768, num_clip_token=77, **{k: v for k, v in cfg.items() if k != "num_classes"})
other = get_nnet("unet_t2i")
print(json.dumps(dict(
    cnf=CNF.__module__, cnf_t=CNF_T.__module__, net=type(net).__module__, net_t=type(net_t).__module__,
    other=other, set_logger=utils_uvit.set_logger(), misc=misc_mark, ae=libs.autoencoder.MARK,
    score=type(CNF(net)).__module__, libs_uvit=__import__("libs.uvit", fromlist=["UViT"]).UViT.__module__)))
"""


def make_fake_reference(root):
    """Same layout and import statements as the reference; its hot-path modules would fail if they were ever imported."""
    files = {
        "flow_matching.py": "import torchdiffeq_that_is_not_installed\nclass CNF: pass\n",
        "flow_matching_t2i.py": "import torchdiffeq_that_is_not_installed\nclass CNF: pass\n",
        "libs/__init__.py": "",
        "libs/uvit.py": "raise ImportError('the reference U-ViT must not be imported')\n",
        "libs/uvit_t2i.py": "raise ImportError('the reference U-ViT must not be imported')\n",
        "libs/autoencoder.py": "MARK = 'reference autoencoder'\n",
        "tools/__init__.py": "",
        "tools/utils_misc.py": "MARK = 'reference utils_misc'\n",
        "tools/utils_uvit.py": textwrap.dedent("""
            def set_logger(*a, **k):
                return 'reference set_logger'
            def get_nnet(name, **kwargs):
                if name == 'uvit':
                    from libs.uvit import UViT
                    return UViT(**kwargs)
                elif name == 'uvit_t2i':
                    from libs.uvit_t2i import UViT
                    return UViT(**kwargs)
                elif name == 'unet_t2i':
                    return 'reference unet'
                raise NotImplementedError(name)
        """),
        "dissect_fake.py": SCRIPT,
    }
    for rel, text in files.items():
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(text)


def check(out):
    r = json.loads(out.strip().splitlines()[-1])
    assert r["cnf"] == "uspace_amd.flow_matching" and r["cnf_t"] == "uspace_amd.flow_matching_t2i"
    assert r["net"] == "uspace_amd.libs.uvit" and r["net_t"] == "uspace_amd.libs.uvit_t2i"
    assert r["libs_uvit"] == "uspace_amd.libs.uvit" and r["score"] == "uspace_amd.flow_matching"
    # the rest of the reference is untouched
    assert r["other"] == "reference unet" and r["set_logger"] == "reference set_logger"
    assert r["misc"] == "reference utils_misc" and r["ae"] == "reference autoencoder"


def run(cmd, cwd, env_extra):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env.update(env_extra)
    p = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stdout


def test_pythonpath_only(tmp_path):
    """`python dissect_lfm.py` with nothing but PYTHONPATH changed: the script directory precedes PYTHONPATH on sys.path,
    so the names are answered by the import hook compat/sitecustomize.py installs."""
    make_fake_reference(str(tmp_path))
    out = run([sys.executable, "dissect_fake.py"], str(tmp_path), {"PYTHONPATH": COMPAT + os.pathsep + ROOT})
    check(out)


def test_launcher_puts_compat_first(tmp_path):
    make_fake_reference(str(tmp_path))
    out = run([sys.executable, os.path.join(COMPAT, "run_ref.py"), "dissect_fake.py"], str(tmp_path), {})
    check(out)


def test_compat_first_on_sys_path_resolves_to_this_package(tmp_path):
    """compat/ first on sys.path, no hook, no reference tree at all."""
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from flow_matching import CNF\nfrom tools.utils_uvit import get_nnet, amortize\n"
            "from libs.uvit_t2i import UViT\n"
            "assert CNF.__module__ == 'uspace_amd.flow_matching', CNF.__module__\n"
            "assert get_nnet.__module__ == 'tools.utils_uvit' and UViT.__module__ == 'uspace_amd.libs.uvit_t2i'\n"
            "n = get_nnet('uvit', img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1)\n"
            "assert type(n).__module__ == 'uspace_amd.libs.uvit' and amortize(10, 4) == [4, 4, 2]\n"
            "try:\n    get_nnet('unet_t2i')\nexcept NotImplementedError:\n    print('ok')\n" % (COMPAT, ROOT))
    out = run([sys.executable, "-c", code], str(tmp_path), {})
    assert out.strip().endswith("ok")


@pytest.mark.skipif(not os.path.isdir("/root/reference/tools"), reason="reference tree not present on this machine")
def test_against_the_real_reference_tree(tmp_path):
    """In the build container: the real reference's `tools.utils_uvit` is shadowed and re-exported (its own imports need the
    leaf-library shims of tests/golden/_refshim.py), `get_nnet` answers from this package."""
    code = textwrap.dedent("""
        import sys
        sys.dont_write_bytecode = True
        sys.path.insert(0, %r)
        from tests.golden import _refshim
        _refshim.install()          # leaf-library stand-ins; also appends /root/reference to sys.path
        for m in [k for k in sys.modules if k.split('.')[0] in ('tools', 'libs')]:
            del sys.modules[m]
        sys.path[:0] = [%r, %r]
        import _overlay; _overlay.install()
        import tools.utils_uvit as u
        from flow_matching import CNF
        net = u.get_nnet('uvit', img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1)
        assert type(net).__module__ == 'uspace_amd.libs.uvit', type(net).__module__
        assert CNF.__module__ == 'uspace_amd.flow_matching'
        assert hasattr(u, 'TrainState') and hasattr(u, 'sample2dir') and u.amortize(5, 2) == [2, 2, 1]
        # ---- train_lfm.py:154-183 / train_lfm_t2i.py:190-204: `score_model.training_losses(...)`, `.mean().backward()`, on CPU.
        # The loss is the reference's own forward (libs/uvit.py:306-351) over THIS module's parameters (compat/_training.py).
        import torch
        torch.manual_seed(3)
        score = CNF(net=net)
        x = torch.randn(3, 4, 32, 32)
        torch.manual_seed(11)
        loss = score.training_losses(x, y=None, sigma_min=1e-4)
        assert loss.shape == (3,) and bool(torch.isfinite(loss).all())
        loss.mean().backward()
        grads = {n: p.grad for n, p in net.named_parameters()}
        assert all(g is not None for g in grads.values()) and sum(float(g.abs().sum()) for g in grads.values()) > 0
        # the same numbers as the reference computes by itself: its own UViT with this state_dict, its formula (flow_matching.py:88-100)
        import libs._ref_uvit as R
        ref = R.UViT(img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1)
        ref.load_state_dict(net.state_dict(), strict=True)
        torch.manual_seed(11)
        noise = torch.randn_like(x); t = torch.rand(len(x)); t_ = t[:, None, None, None]
        want = ((ref(t_ * x + (1 - (1 - 1e-4) * t_) * noise, t, None, edit_loc=None)[0] - (x - (1 - 1e-4) * noise)).square().mean(dim=(1, 2, 3)))
        assert torch.equal(loss.detach(), want.detach()), (loss, want)
        # an optimizer step on nnet.parameters() is seen by the twin (one set of tensors), and nothing is listed twice
        assert len(list(net.parameters())) == len(net.state_dict())
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        opt.step()
        torch.manual_seed(11)
        loss2 = score.training_losses(x, y=None, sigma_min=1e-4)
        assert not torch.equal(loss2.detach(), loss.detach())
        # T2I twin
        from flow_matching_t2i import CNF as CNF_T
        net_t = u.get_nnet('uvit_t2i', img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, clip_dim=64, num_clip_token=5)
        lt = CNF_T(net=net_t).training_losses(x, context=torch.randn(3, 5, 64), sigma_min=1e-4)
        lt.mean().backward()
        assert lt.shape == (3,) and net_t.context_embed.weight.grad is not None
        # ---- ADVICE r5: (a) a wrapped network is refused (the loss would bypass the wrapper's forward: no gradient all-reduce under DDP)
        class Wrapper(torch.nn.Module):
            def __init__(self, m):
                super().__init__(); self.module = m
        try:
            CNF(net=Wrapper(net)).training_losses(x, y=None, sigma_min=1e-4)
            raise SystemExit('a wrapped network was accepted')
        except NotImplementedError as e:
            assert 'single-process' in str(e)
        # (b) the twin follows the module's use_checkpoint; nnet and a second network (nnet_ema) get twins of ONE class object
        net_ck = u.get_nnet('uvit', img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, use_checkpoint=True)
        CNF(net=net_ck).training_losses(x, y=None, sigma_min=1e-4).mean().backward()
        assert net_ck._reference_twin.in_blocks[0].use_checkpoint is True and net._reference_twin.in_blocks[0].use_checkpoint is False
        assert type(net_ck._reference_twin) is type(net._reference_twin)
        # (c) weights edited through .data (the EMA update, tools/utils_uvit.py:109) are invisible to the packed blob: the overlay's decode
        # forgets the blob of any network after a training step of the process
        import _training
        calls = []
        net_ck.invalidate_packed = lambda: calls.append(1)
        _training.refresh(net_ck); _training.refresh(net_ck)
        assert calls == [1]
        CNF(net=net).training_losses(x, y=None, sigma_min=1e-4)
        _training.refresh(net_ck)
        assert calls == [1, 1]
        print('ok')
    """ % (ROOT, COMPAT, ROOT))
    out = run([sys.executable, "-c", code], str(tmp_path), {"PYTHONDONTWRITEBYTECODE": "1"})
    assert out.strip().endswith("ok")


EVAL_SCRIPT = """
# the eval path of train_lfm_t2i.py:210-253 in miniature: sample_fn -> score_model.decode(noise, context=..., **config.dissection)
# with dissect_name=None (=> the reference's default sampler, adaptive dopri5 1e-5, flow_matching_t2i.py:77-83) -> sample2dir
import json, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, %(root)r)
from tests.test_multigpu_readiness import _FakeKernels
from uspace_amd import _hip
Fake = _FakeKernels
fake = Fake(_hip.lib())
_hip.lib = lambda: fake
_hip.require_device = lambda t, name="tensor": None
_hip.stream_ptr = lambda: None
_hip.sync_current_stream = lambda: None
from flow_matching_t2i import CNF
from tools.utils_uvit import get_nnet, sample2dir
class Acc:
    is_main_process, num_processes = True, 1
    def gather(self, t): return t
net = get_nnet("uvit_t2i", img_size=32, patch_size=2, in_chans=4, embed_dim=64, depth=2, num_heads=1, clip_dim=64, num_clip_token=5)
score_model = CNF(net=net)
dissection = dict(dissect_name=None, edit_loc=None, solver_kwargs=dict(solver="adaptive", solver_adaptive="dopri5"))
def sample_fn(n):
    z = score_model.decode(torch.randn(n, 4, 32, 32), context=torch.randn(n, 5, 64), **dissection)
    return z[:, :3]
with tempfile.TemporaryDirectory() as d:
    sample2dir(Acc(), d, 5, 2, sample_fn, unpreprocess_fn=lambda v: v)
    saved = sorted(os.listdir(d))
st = score_model.last_stats
try:
    score_model.training_losses(torch.randn(2, 4, 32, 32), context=torch.randn(2, 5, 64), sigma_min=1e-4)
    tl = "no error"
except NotImplementedError as e:
    tl = "NotImplementedError"
print(json.dumps(dict(saved=saved, nfe=st.nfe, accepted=st.accepted, forwards=fake.forwards, cnf=CNF.__module__, tl=tl)))
"""


def test_eval_path_of_the_training_script_on_a_fake_tree(tmp_path):
    """`train_lfm_t2i.py`'s evaluation (its only use of this package besides the loss): `sample2dir` -> `sample_fn` -> `decode` with
    `dissect_name=None`, i.e. adaptive Dormand-Prince, through the overlay's import names; kernels stubbed at `_hip.lib()` (v = -x: the
    exact solution e^-1 z is reached with few accepted steps).  The loss itself has nothing to delegate to here -- the fake tree's
    U-ViT refuses to import -- and says so."""
    make_fake_reference(str(tmp_path))
    with open(tmp_path / "tools" / "utils_uvit.py", "a") as f:
        f.write(textwrap.dedent("""
            import os
            def sample2dir(accelerator, path, n_samples, mini_batch_size, sample_fn, unpreprocess_fn=None):
                os.makedirs(path, exist_ok=True)
                idx, bs = 0, mini_batch_size * accelerator.num_processes
                k, r = divmod(n_samples, bs)
                for b in k * [bs] + ([r] if r else []):
                    samples = accelerator.gather(unpreprocess_fn(sample_fn(mini_batch_size)).contiguous())[:b]
                    for s in samples:
                        open(os.path.join(path, f"{idx}.png"), "w").write(str(tuple(s.shape)))
                        idx += 1
        """))
    with open(tmp_path / "train_fake.py", "w") as f:
        f.write(EVAL_SCRIPT % dict(root=ROOT))
    out = run([sys.executable, "train_fake.py"], str(tmp_path), {"PYTHONPATH": COMPAT + os.pathsep + ROOT})
    r = json.loads(out.strip().splitlines()[-1])
    assert r["saved"] == [f"{i}.png" for i in range(5)] and r["cnf"] == "uspace_amd.flow_matching_t2i"
    assert r["accepted"] >= 1 and r["nfe"] >= 2 + 6 * r["accepted"] and r["forwards"] >= 3 * r["nfe"] - 12
    assert r["tl"] == "NotImplementedError"
