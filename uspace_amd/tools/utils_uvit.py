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
