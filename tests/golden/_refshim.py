"""Import shims for loading the *reference* (read-only at /root/reference) in this
container, used ONLY by make_golden.py to generate fixture vectors.

The reference hot-path modules import leaf visualisation / logging libraries that are
not installed here (absl, torchvision, cv2, IPython, tqdm.notebook, ml_collections).
None of them is called on the U-ViT forward path, so inert stand-in
modules are registered in ``sys.modules`` before the import.  Nothing from the
reference is copied: it is imported from where it lies and never travels.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("USPACE_REFERENCE_ROOT", "/root/reference")


class _Inert(types.ModuleType):
    """Module whose every attribute is a no-op callable / nested inert object."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def _noop(*a, **k):
            return None

        _noop.__name__ = name
        return _noop


def _stub(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _Inert(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave like a package so "import a.b" works
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(_stub(parent), child, m)
    return m


def install():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must be imported before torchvision is stubbed)

    for name in (
        "absl", "absl.logging", "absl.flags", "absl.app",
        "torchvision", "torchvision.io", "torchvision.utils", "torchvision.transforms",
        "torchvision.transforms.functional", "torchvision.datasets", "torchvision.models",
        "cv2", "IPython", "IPython.display", "tqdm.notebook", "ml_collections",
        "h5py", "wandb", "timm", "xformers",
    ):
        try:
            __import__(name)
        except Exception:
            _stub(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference():
    """Return (uvit_module, uvit_t2i_module) of the reference."""
    install()
    import importlib

    uvit = importlib.import_module("libs.uvit")
    uvit_t2i = importlib.import_module("libs.uvit_t2i")
    return uvit, uvit_t2i
