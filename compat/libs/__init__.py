# `libs` stays the reference's package: every module not overlaid here (autoencoder, clip, timm, sd, ...) is found in the
# same-named directories further down sys.path.
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
