# `tools` stays the reference's package: every module not overlaid here (utils_vis, utils_t2i, fid_score, ...) is found in the
# same-named directories further down sys.path.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
