"""Names of the reference that the overlay answers, and the import hook for the PYTHONPATH-only form."""
import importlib.abc
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

# reference module name -> file of this directory
OVERLAY = {
    "flow_matching": os.path.join(HERE, "flow_matching.py"),
    "flow_matching_t2i": os.path.join(HERE, "flow_matching_t2i.py"),
    "libs.uvit": os.path.join(HERE, "libs", "uvit.py"),
    "libs.uvit_t2i": os.path.join(HERE, "libs", "uvit_t2i.py"),
    "tools.utils_uvit": os.path.join(HERE, "tools", "utils_uvit.py"),
}


class OverlayFinder(importlib.abc.MetaPathFinder):
    """Answers the five overlaid names before sys.path is consulted (the script directory precedes PYTHONPATH)."""

    def find_spec(self, fullname, path=None, target=None):
        f = OVERLAY.get(fullname)
        if f is None:
            return None
        return importlib.util.spec_from_file_location(fullname, f)


def install():
    if REPO not in sys.path:
        sys.path.append(REPO)                      # `import uspace_amd`
    if not any(isinstance(m, OverlayFinder) for m in sys.meta_path):
        sys.meta_path.insert(0, OverlayFinder())


def load_shadowed(fullname, own_file):
    """The reference's module of the same name (the next `<pkg>/<mod>.py` on the package path that is not `own_file`),
    executed under a private name; None if there is none (e.g. the overlay is used without the reference tree)."""
    pkg, _, mod = fullname.rpartition(".")
    paths = list(sys.modules[pkg].__path__) if pkg and pkg in sys.modules else list(sys.path)
    own = os.path.realpath(own_file)
    for p in paths:
        cand = os.path.join(p or ".", mod + ".py")
        if os.path.isfile(cand) and os.path.realpath(cand) != own:
            name = (pkg + "." if pkg else "") + "_ref_" + mod
            if name in sys.modules:                 # one execution per process: nnet and nnet_ema get twins of ONE class object
                return sys.modules[name]
            spec = importlib.util.spec_from_file_location(name, cand)
            m = importlib.util.module_from_spec(spec)
            sys.modules[name] = m
            try:
                spec.loader.exec_module(m)
            except BaseException:
                sys.modules.pop(name, None)         # no half-initialised module left behind
                raise
            return m
    return None
