"""VAE decode on the GPU: FrozenAutoencoderKL.decode of the reference
(libs/autoencoder.py:303-409, 446-450) through the C-ABI, against the reference's golden image and the
CPU oracle at the tiny configuration, and through size-independent properties at the real 256^2 shape."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu

SD_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                   ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def test_tiny_decoder_matches_reference_golden(golden_dir):
    from uspace_amd.libs.autoencoder import FrozenAutoencoderKL
    z = np.load(os.path.join(golden_dir, "vae_decoder_tiny.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    torch.manual_seed(meta["weight_seed"])
    vae = FrozenAutoencoderKL(meta["ddconfig"], 4).cuda()
    zz = torch.from_numpy(z["z"]).cuda()
    img = vae.decode(zz)
    assert img.shape == (3, 3, 16, 16) and img.dtype == torch.float32
    r = rel_l2(img.cpu().numpy(), z["img"])
    m = float(np.abs(img.cpu().numpy() - z["img"]).max() / np.abs(z["img"]).max())
    assert r < 1.5e-2 and m < 4e-2, (r, m)          # bf16 conv operands, fp32 accumulation / norms / residuals
    # chunked decode (1 image at a time) and repeated calls agree; input untouched
    z0 = zz.clone()
    a = vae.decode(zz, chunk=1)
    assert rel_l2(a.cpu().numpy(), img.cpu().numpy()) < 1e-3 and torch.equal(zz, z0)
    assert torch.equal(vae.decode(zz), img)
    assert torch.equal(vae(zz, "decode"), img)
    with pytest.raises(NotImplementedError):
        vae(zz, "encode")
    # a full checkpoint (with encoder.* / quant_conv.* entries) loads; those halves are ignored
    sd = dict(vae.state_dict())
    sd["encoder.conv_in.weight"] = torch.zeros(1)
    sd["quant_conv.weight"] = torch.zeros(1)
    vae.load_state_dict(sd)


def test_sd_vae_shape_runs_and_is_batch_consistent():
    """The real decoder (ch=128, mult 1-2-4-4, 4x32x32 -> 3x256x256), seeded default init."""
    from uspace_amd.libs.autoencoder import FrozenAutoencoderKL
    torch.manual_seed(1234)
    vae = FrozenAutoencoderKL(SD_DDCONFIG, 4).cuda()
    assert sum(p.numel() for p in vae.parameters()) == 49490199      # decoder 49,490,179 + post_quant_conv 20
    g = torch.Generator().manual_seed(7)
    z = (torch.randn(5, 4, 32, 32, generator=g) * 0.18215).cuda()
    img = vae.decode(z, chunk=4)                      # 4 + 1: ragged last chunk
    assert img.shape == (5, 3, 256, 256) and bool(torch.isfinite(img).all()) and float(img.std()) > 1e-4
    one = vae.decode(z[3:4].contiguous())
    assert rel_l2(one.cpu().numpy(), img[3:4].cpu().numpy()) < 2e-3
    # shifting the latent changes the image; zero latent gives a constant-free but finite image
    assert not torch.equal(vae.decode(z * 0.5, chunk=4), img)
