"""Principal directions of tapped activations for the ``write_pca`` hook (reference: tools/utils_pca.py:13-50 on top of
tools/utils_vis.py:62-118 -- faiss ``PCAMatrix`` / sklearn ``PCA(svd_solver="full")`` over ``[N, C*W*H]`` features that
the read hook saved as ``{batch_id}_{t:.2f}.npy``; output ``pca{n}_{t}.npy`` of shape ``[n, C, W, H]``).

The features stay on the device: N (a few thousand samples) is far below the feature size (4 096 for the latent,
263 168 for the mid block), so the directions come from the N x N Gram matrix of the centred data instead of an SVD of the
N x F matrix on the host.  Centring, the Gram matrix (fp64 matrix cores: products of fp32 data are exact, so small trailing
components are not drowned by the squared condition number of an fp32 Gram matrix), the projection U^T x and the
normalisation are kernels of libuspace_hip.so (csrc/pca.hip); only the N x N symmetric eigen-decomposition is a library
call.  Signs follow sklearn's convention (largest-magnitude entry of each direction positive); faiss' are arbitrary.
"""
import os

import numpy as np
import torch

from .. import _hip


def should_ignore(name):
    """Files in a feature directory that are not ``{batch_id}_{t}.npy`` taps (tools/utils_attr.py ``should_ignore``:
    the direction / component files written next to them)."""
    return name.startswith(("delta_", "pca", "latents", "attr")) or not name.endswith(".npy")


def pca_components(feats, n_components):
    """feats [N, ...] (CUDA tensor) -> [n_components, ...] fp32 on the same device: unit-norm principal directions,
    largest variance first."""
    _hip.require_device(feats, "feats")
    N = feats.shape[0]
    shape = tuple(feats.shape[1:])
    if not (0 < n_components <= min(N, int(np.prod(shape)))):
        raise ValueError(f"n_components={n_components} must be between 1 and min(n_samples, n_features)")
    x = feats.detach().to(torch.float32).reshape(N, -1).contiguous()
    F = x.shape[1]
    pad = (-F) % 4                                       # the kernels read 16-byte feature groups
    if pad:
        x = torch.cat([x, x.new_zeros(N, pad)], dim=1)
    Fp = F + pad
    L, st = _hip.lib(), _hip.stream_ptr()
    xc = torch.empty_like(x)
    _hip.check(L.uspace_center_cols_f32(_hip.ptr(x), _hip.ptr(xc), N, Fp, st), "uspace_center_cols_f32")
    gram = torch.empty(N, N, dtype=torch.float64, device=x.device)
    _hip.check(L.uspace_gram_f64(_hip.ptr(xc), _hip.ptr(gram), N, Fp, st), "uspace_gram_f64")
    evals, evecs = torch.linalg.eigh(gram)               # N x N, ascending: its eigenvectors are the left singular vectors of xc
    idx = torch.argsort(evals, descending=True)[:n_components]
    ut = evecs[:, idx].t().contiguous()                  # [n, N] fp64
    comps = torch.empty(n_components, Fp, dtype=torch.float32, device=x.device)
    _hip.check(L.uspace_project_rows_f64(_hip.ptr(ut), _hip.ptr(xc), _hip.ptr(comps), n_components, N, Fp, st),
               "uspace_project_rows_f64")                # sigma_i * v_i
    _hip.check(L.uspace_normalize_rows_signed(_hip.ptr(comps), n_components, Fp, st), "uspace_normalize_rows_signed")
    if pad:
        comps = comps[:, :F].contiguous()
    return comps.reshape((n_components,) + shape)


class PcaAccumulator:
    """Collects the activations the read hook taps, per timestep, in HBM (``update`` has the signature the hook's
    accumulator protocol uses) and writes ``pca{n}_{t}.npy`` at the end -- the device-resident form of
    ``extract_hspace_feat_unet_by_pca``."""

    def __init__(self):
        self._feats = {}

    def update(self, timestep_digit, feats, attrs=None):
        _hip.require_device(feats, "feats")
        self._feats.setdefault(timestep_digit, []).append(feats.detach().to(torch.float32).clone())

    def components(self, timestep_digit, n_components):
        return pca_components(torch.cat(self._feats[timestep_digit], dim=0), n_components)

    def finalize(self, write_root, n_components):
        os.makedirs(write_root, exist_ok=True)
        for t in sorted(self._feats):
            np.save(os.path.join(write_root, f"pca{n_components}_{t}"), self.components(t, n_components).cpu().numpy())
        return sorted(self._feats)


def extract_hspace_feat_unet_by_pca(read_path_root="mid_feat/unet_latent0_euler100", n_components=50, batch_num=10,
                                    is_debug=False, device="cuda"):
    """Same files in, same files out as the reference (tools/utils_pca.py:13-50): for every timestep found in
    ``read_path_root`` concatenate ``{batch_id}_{t}.npy`` for batch_id < batch_num and write ``pca{n}_{t}.npy``."""
    names = [n for n in os.listdir(read_path_root) if not should_ignore(n)]
    steps = sorted({n.split("_")[1].replace(".npy", "") for n in names})
    for t in steps:
        feats = [np.load(os.path.join(read_path_root, f"{b}_{t}.npy")) for b in range(batch_num)]
        if len(feats) == 0:
            raise ValueError("**** empty feat", t)
        x = torch.from_numpy(np.concatenate(feats, axis=0)).to(device)
        target = os.path.join(read_path_root, f"pca{n_components}_{t}" + ("_debug" if is_debug else ""))
        np.save(target, pca_components(x, n_components).cpu().numpy())
    return steps
