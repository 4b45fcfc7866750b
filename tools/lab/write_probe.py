"""Write-only bandwidth of the chip by burst size (question behind the 7.3 us per round that a round of GEMM tiles takes to drain
its 33.5 MB of bf16 outputs): `fill_` of n MB, same region every time / walking through 2 GiB.
    python tools/lab/write_probe.py"""
import torch


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); b.synchronize()
    return a.elapsed_time(b) * 1e3


def main():
    dev = torch.device("cuda:0")
    big = torch.empty(1 << 29, device=dev, dtype=torch.float32)          # 2 GiB
    for mb in (8, 16, 33.5, 67, 135, 270, 1024):
        n = int(mb * (1 << 20) / 4)
        same, walk = [], []
        for r in range(12):
            same.append(timed(lambda: big[:n].fill_(1.0)))
        off = 0
        for r in range(12):
            if off + n > big.numel():
                off = 0
            walk.append(timed(lambda: big[off:off + n].fill_(2.0)))
            off += n + (1 << 20)
        same.sort(); walk.sort()
        s, w = same[len(same) // 2], walk[len(walk) // 2]
        print(f"{mb:7.1f} MB | same region {s:8.1f} us = {n * 4 / s * 1e-6:5.2f} TB/s | fresh regions {w:8.1f} us = {n * 4 / w * 1e-6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
