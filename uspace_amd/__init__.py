"""uspace_amd -- MI355X (gfx950) implementation of uspace's flow-matching sampling hot path.

Drop-in surface (mirrors dongzhuoyao/uspace):
    uspace_amd.tools.utils_uvit.get_nnet(name, **kwargs)      tools/utils_uvit.py:27
    uspace_amd.libs.uvit.UViT / uspace_amd.libs.uvit_t2i.UViT  nnet(x, timesteps, ...) -> (pred, None)
    uspace_amd.flow_matching.CNF / flow_matching_t2i.CNF       decode / encode / decode_fixadp
All arithmetic runs in hand-written HIP kernels behind include/uspace_hip.h; there is no
CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
