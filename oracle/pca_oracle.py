"""CPU restatement of the PCA direction extraction (tools/utils_vis.py:80-118 ``get_pca_components_sklearn`` =
sklearn ``PCA(svd_solver="full")``; tools/utils_pca.py:13-50 writes the result as ``pca{n}_{t}.npy``).
TEST INFRASTRUCTURE ONLY.  Pinned against ``tests/golden/pca_components.npz`` (the reference function itself)."""
import numpy as np


def pca_components(feats, n_components):
    """feats [N, ...] -> [n_components, ...]: right singular vectors of the centred data, largest variance first,
    with sklearn's deterministic sign (``svd_flip``, v-based: the entry of largest magnitude in each component is
    positive)."""
    x = np.asarray(feats, np.float64)
    shape = x.shape[1:]
    x = x.reshape(len(x), -1)
    x = x - x.mean(axis=0, keepdims=True)
    _u, _s, vt = np.linalg.svd(x, full_matrices=False)
    vt = vt[:n_components]
    idx = np.argmax(np.abs(vt), axis=1)
    vt = vt * np.sign(vt[np.arange(len(vt)), idx])[:, None]
    return vt.reshape((len(vt),) + shape).astype(np.float32)
