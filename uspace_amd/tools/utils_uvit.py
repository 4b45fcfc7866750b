"""The plug-in seam: string-keyed network factory (reference: tools/utils_uvit.py:27-41)."""


def get_nnet(name, **kwargs):
    if name == "uvit":
        from ..libs.uvit import UViT

        return UViT(**kwargs)
    if name == "uvit_t2i":
        from ..libs.uvit_t2i import UViT

        return UViT(**kwargs)
    if name == "unet_t2i":
        raise NotImplementedError("unet_t2i (SD UNet) is outside the U-ViT hot path this package implements")
    raise NotImplementedError(name)


def amortize(n_samples, batch_size):
    """Split n_samples into full batches plus a remainder (reference: tools/utils_uvit.py:258-261)."""
    k, r = divmod(n_samples, batch_size)
    return [batch_size] * k + ([r] if r else [])


class DistAccelerator:
    """The three members of ``accelerate.Accelerator`` the sampling loops use (tools/utils_uvit.py:264-277,
    tools/utils_vis.py:138-255) over plain ``torch.distributed`` (one process per GPU, RCCL): ``num_processes``,
    ``is_main_process`` and ``gather`` (all-gather along dim 0, rank order).  A real Accelerator can be passed to the
    loops instead; this class only removes the dependency for torchrun-style launches."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist, self._group = dist, group
        on = dist.is_available() and dist.is_initialized()
        self.num_processes = dist.get_world_size(group) if on else 1
        self.process_index = dist.get_rank(group) if on else 0

    @property
    def is_main_process(self):
        return self.process_index == 0

    def gather(self, t):
        if self.num_processes == 1:
            return t
        from ..sampling import gather_batch
        return gather_batch(t, t.shape[0] * self.num_processes, self._group)


def save_image(img, path):
    """``torchvision.utils.save_image`` for one [C,H,W] float image in [0,1] (C = 1 or 3): x*255 + 0.5, clamp, uint8, PNG."""
    import numpy as np
    from PIL import Image
    arr = img.detach().float().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu").numpy().astype(np.uint8)
    Image.fromarray(arr[:, :, 0] if arr.shape[2] == 1 else arr).save(path)


def sample2dir(accelerator, path, n_samples, mini_batch_size, sample_fn, unpreprocess_fn=None, save_fn=None):
    """The multi-GPU sampling loop of the reference (tools/utils_uvit.py:264-277), same signature: every process draws
    ``mini_batch_size`` samples per round with ``sample_fn(mini_batch_size)``, the rounds are gathered in rank order (the one
    collective of the path) and the main process writes ``{idx}.png``.  ``save_fn(sample, file)`` defaults to the PNG writer
    above (the reference uses torchvision's)."""
    import os
    os.makedirs(path, exist_ok=True)
    save_fn = save_fn or save_image
    unpreprocess_fn = unpreprocess_fn or (lambda v: v)
    idx = 0
    batch_size = mini_batch_size * accelerator.num_processes
    for _batch_size in amortize(n_samples, batch_size):
        samples = unpreprocess_fn(sample_fn(mini_batch_size))
        samples = accelerator.gather(samples.contiguous())[:_batch_size]
        if accelerator.is_main_process:
            for sample in samples:
                save_fn(sample, os.path.join(path, f"{idx}.png"))
                idx += 1
    return idx
