"""fp32 CPU restatement of the reference's VAE decode (libs/autoencoder.py:303-409 Decoder, :446-450 decode).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against tests/golden/vae_decoder_tiny.npz
(image + intermediates produced by importing the reference's ``Decoder``).
"""
import numpy as np

from . import _cops as C


def swish(x):
    return (x / (1.0 + np.exp(-x))).astype(np.float32)      # x * sigmoid(x), libs/autoencoder.py:26-28


def resnet_block(x, sd, pre):
    """libs/autoencoder.py:114-134 with temb=None, dropout 0."""
    h = C.groupnorm(x, sd[pre + ".norm1.weight"], sd[pre + ".norm1.bias"])
    h = C.conv2d(swish(h), sd[pre + ".conv1.weight"], sd[pre + ".conv1.bias"])
    h = C.groupnorm(h, sd[pre + ".norm2.weight"], sd[pre + ".norm2.bias"])
    h = C.conv2d(swish(h), sd[pre + ".conv2.weight"], sd[pre + ".conv2.bias"])
    if pre + ".nin_shortcut.weight" in sd:
        x = C.conv2d(x, sd[pre + ".nin_shortcut.weight"], sd[pre + ".nin_shortcut.bias"])
    return (x + h).astype(np.float32)


def attn_block(x, sd, pre):
    """libs/autoencoder.py:171-195: single head over h*w tokens, scale c^-0.5, no SiLU after the norm."""
    B, Cc, H, W = x.shape
    h = C.groupnorm(x, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"])
    q = C.conv2d(h, sd[pre + ".q.weight"], sd[pre + ".q.bias"]).reshape(B, Cc, H * W)
    k = C.conv2d(h, sd[pre + ".k.weight"], sd[pre + ".k.bias"]).reshape(B, Cc, H * W)
    v = C.conv2d(h, sd[pre + ".v.weight"], sd[pre + ".v.bias"]).reshape(B, Cc, H * W)
    w = np.einsum("bci,bcj->bij", q, k).astype(np.float32) * np.float32(int(Cc) ** -0.5)
    w = w - w.max(axis=2, keepdims=True)
    w = np.exp(w)
    w = (w / w.sum(axis=2, keepdims=True)).astype(np.float32)
    o = np.einsum("bcj,bij->bci", v, w).astype(np.float32).reshape(B, Cc, H, W)
    o = C.conv2d(o, sd[pre + ".proj_out.weight"], sd[pre + ".proj_out.bias"])
    return (x + o).astype(np.float32)


def decode(sd, z, ch_mult, num_res_blocks, scale_factor=0.18215, taps=None):
    """FrozenAutoencoderKL.decode: z/scale -> post_quant_conv -> Decoder.forward."""
    z = (np.asarray(z, np.float32) * np.float32(1.0 / scale_factor)).astype(np.float32)
    h = C.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    h = C.conv2d(h, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"])
    tap = (lambda n, v: taps.__setitem__(n, v.copy())) if taps is not None else (lambda n, v: None)
    tap("conv_in", h)
    h = resnet_block(h, sd, "decoder.mid.block_1")
    h = attn_block(h, sd, "decoder.mid.attn_1")
    tap("attn", h)
    h = resnet_block(h, sd, "decoder.mid.block_2")
    for lvl in reversed(range(len(ch_mult))):
        for i in range(num_res_blocks + 1):
            h = resnet_block(h, sd, f"decoder.up.{lvl}.block.{i}")
            if lvl == 0 and i == 0:
                tap("up0_b0", h)
        if lvl != 0:
            h = np.repeat(np.repeat(h, 2, axis=2), 2, axis=3)      # F.interpolate(scale 2, nearest)
            h = C.conv2d(h, sd[f"decoder.up.{lvl}.upsample.conv.weight"], sd[f"decoder.up.{lvl}.upsample.conv.bias"])
            if lvl == 1:
                tap("up1_us", h)
    h = C.groupnorm(h, sd["decoder.norm_out.weight"], sd["decoder.norm_out.bias"])
    tap("norm_out", h)
    return C.conv2d(swish(h), sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"])
