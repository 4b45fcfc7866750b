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
        print('ok')
    """ % (ROOT, COMPAT, ROOT))
    out = run([sys.executable, "-c", code], str(tmp_path), {"PYTHONDONTWRITEBYTECODE": "1"})
    assert out.strip().endswith("ok")
