"""Throughput of the SD KL-VAE decode (4x32x32 -> 3x256x256) on the HIP path, with the algorithmic FLOPs of the
reference's Decoder (libs/autoencoder.py:303-409) counted from the configuration.
    python tools/vae_bench.py [--batch 8] [--iters 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SD = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
          num_res_blocks=2, attn_resolutions=[], dropout=0.0)


def decoder_flops(dd):
    """2*MACs of every conv / attention matmul of one image."""
    ch, mult, nrb, res = dd["ch"], dd["ch_mult"], dd["num_res_blocks"], dd["resolution"]
    n = len(mult)
    h = res // 2 ** (n - 1)
    c = ch * mult[-1]
    f = 0

    def conv(ci, co, k, hh):
        return 2 * ci * co * k * k * hh * hh

    def resb(ci, co, hh):
        return conv(ci, co, 3, hh) + conv(co, co, 3, hh) + (conv(ci, co, 1, hh) if ci != co else 0)
    f += conv(4, 4, 1, h) + conv(4, c, 3, h)
    f += 2 * resb(c, c, h) + 4 * conv(c, c, 1, h) + 2 * 2 * (h * h) ** 2 * c
    for lvl in reversed(range(n)):
        co = ch * mult[lvl]
        for _ in range(nrb + 1):
            f += resb(c, co, h)
            c = co
        if lvl:
            h *= 2
            f += conv(c, c, 3, h)
    f += conv(c, 3, 3, h)
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from uspace_amd.libs.autoencoder import FrozenAutoencoderKL
    torch.manual_seed(0)
    vae = FrozenAutoencoderKL(SD, 4).cuda()
    z = (torch.randn(a.batch, 4, 32, 32) * 0.18215).cuda()
    vae.decode(z, chunk=a.batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        vae.decode(z, chunk=a.batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    fl = decoder_flops(SD)
    print(json.dumps({"workload": "SD KL-VAE decode 4x32x32 -> 3x256x256", "batch": a.batch, "ms_per_batch": dt * 1e3,
                      "img_per_s": a.batch / dt, "gflop_per_img": fl / 1e9, "tflops": fl * a.batch / dt / 1e12}))


if __name__ == "__main__":
    main()
