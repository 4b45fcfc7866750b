"""The u-space visualisation loop of the reference (tools/utils_vis.py:138-255): per round every process draws (or loads)
``mini_batch_size`` latents, runs one solve per ``write_scale`` with the edit hook active, the rows are gathered in rank order and
the main process writes one grid image per round.  This is the single entry point of BASELINE config 5 (U-ViT-L, mid-block
u-space hook, batch 256 over 8 GPUs).

Extension: ``sweep_fn(input_z=..., write_scales=[...], batch_id=..., **kwargs) -> [n_scales, B, C, H, W]`` runs the whole
sweep as ONE solve over n_scales * B rows (flow_matching.CNF.decode_write_scales, exact for fixed-step solvers) instead of
``len(write_scales)`` sequential solves; the written image is the same."""
import datetime
import os

import numpy as np
import torch

from .utils_attr import get_attr_name_from_attr_id
from .utils_uvit import amortize

_PADDING = 8          # tools/utils_vis.py:18


def pretty_datetime():
    return datetime.datetime.now().strftime("%Y%m%d_%H%M%S")


def make_grid(images, nrow, padding=_PADDING, pad_value=0.0):
    """torchvision.utils.make_grid for a [N,C,H,W] batch (no normalisation): ``nrow`` images per row, ``padding`` pixels of
    ``pad_value`` between and around them."""
    n, c, h, w = images.shape
    if c == 1:
        images = images.expand(n, 3, h, w)
        c = 3
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    hh, ww = h + padding, w + padding
    grid = images.new_full((c, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = images[k]
            k += 1
    return grid


def load_z_from_dir(root_path, has_attr=False, device=None):
    """tools/utils_vis.py:25-35: with ``has_attr`` the latents live in ``root_path + ".npz"`` under the key "latent" (the file
    the extraction step writes next to the attribute table, dissect_lfm.py:224-236); without it ``root_path`` is a plain
    ``.npy``.  Leniency beyond the reference: with ``has_attr`` an explicit ``*.npz`` path is accepted as well."""
    if has_attr:
        path = root_path + ".npz"
        if not os.path.exists(path) and str(root_path).endswith(".npz") and os.path.exists(root_path):
            path = root_path
        z = np.load(path, allow_pickle=False)["latent"]
    else:
        z = np.load(root_path, allow_pickle=False)
    return torch.from_numpy(np.asarray(z)).to(device)


def save_grid_topil(grid, path):
    """What the reference does with the grid (tools/utils_vis.py:249-251): ``ToPILImage()`` = ``mul(255).byte()`` --
    TRUNCATION, not the rounding of ``save_image`` -- then ``img.save``.  Values are clamped to [0, 255] first (the
    reference's ``byte()`` wraps outside that range; its inputs are already clamped by ``unpreprocess``)."""
    from PIL import Image
    arr = grid.detach().float().mul(255).clamp_(0, 255).to(torch.uint8).permute(1, 2, 0).to("cpu").numpy()
    Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(path)


def sample_for_hspace_vis(accelerator, path, sample_fn, unpreprocess_fn=None, padding=_PADDING, pad_value=1.0,
                          z_shape=None, device=None, n_samples=None, mini_batch_size=None, write_scales=None,
                          fixed_z_path=None, sweep_fn=None, attr_name_fn=None, save_grid_fn=None, generator=None, **kwargs):
    """Same signature and loop as the reference (tools/utils_vis.py:138-255) for ``dissect_name`` in {"read", "write_pca",
    "write_attr", "write_x0"}; returns the list of written files (main process) -- the reference returns None."""
    os.makedirs(path, exist_ok=True)
    unpreprocess_fn = unpreprocess_fn or (lambda v: v)
    idx = 0
    written = []
    batch_size = mini_batch_size * accelerator.num_processes
    _seed = kwargs.get("seed", None)
    name = kwargs["dissect_name"]
    _latent_z = None
    if name in ("write_pca", "write_attr", "write_x0") and fixed_z_path is not None:
        _latent_z = load_z_from_dir(fixed_z_path, has_attr=kwargs["has_attr"], device=device)     # KeyError like the reference
        n_samples = len(_latent_z)

    def draw():
        return torch.randn(mini_batch_size, *z_shape, device=device, generator=generator)

    for _batch_id, _batch_size in enumerate(amortize(n_samples, batch_size)):
        if name == "read":
            samples = unpreprocess_fn(sample_fn(input_z=draw(), batch_id=_batch_id, **kwargs))
        elif name in ("write_pca", "write_attr", "write_x0"):
            input_z = draw() if _latent_z is None else _latent_z[_batch_id * mini_batch_size:(_batch_id + 1) * mini_batch_size]
            if name == "write_x0":
                direction = np.load(os.path.join(kwargs["write_path_root"], "delta_latentz.npy"))[kwargs.get("ith_attr", None)]
                direction = torch.from_numpy(np.asarray(direction, np.float32)).to(device)[None]
            if sweep_fn is not None and name != "write_x0":
                sw = sweep_fn(input_z=input_z, write_scales=list(write_scales), batch_id=_batch_id, **kwargs)   # [S,B,...]
                samples = torch.stack([unpreprocess_fn(s) for s in sw], dim=1)                                   # [B,S,...]
            else:
                cols = []
                for write_scale in write_scales:
                    zz = input_z + write_scale * direction if name == "write_x0" else input_z
                    cols.append(unpreprocess_fn(sample_fn(input_z=zz, write_scale=write_scale, batch_id=_batch_id, **kwargs)))
                samples = torch.stack(cols, dim=1)
            samples = samples.reshape(-1, *samples.shape[2:])                                                    # (b s) c h w
        else:
            raise NotImplementedError(f"dissect_name should be read or write, but got: {name}")

        ith_attr = kwargs.get("ith_attr", None)
        _attr_name = (attr_name_fn or get_attr_name_from_attr_id)(ith_attr, kwargs["dataset_name"])
        samples = accelerator.gather(samples.contiguous())
        if accelerator.is_main_process:
            scales = "_".join(f"{s:.2f}" for s in (write_scales or []))
            img_path = os.path.join(path, f"{pretty_datetime()}_seed{_seed}_{idx}_{_attr_name}{scales}.png")
            grid = make_grid(samples, nrow=len(write_scales) if write_scales else 8, padding=padding, pad_value=pad_value)
            (save_grid_fn or save_grid_topil)(grid, img_path)
            written.append(img_path)
        idx += 1
    return written
