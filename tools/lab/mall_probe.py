"""Is a 67 MB fp32 tensor (the residual stream of U-ViT-L at 64 x 257 rows) served from the memory-side cache when it is read
right after being written, and after how much other traffic is it gone?  (Question behind prefetching the epilogue's residual
rows during the K loop.)  Prints the read rate of `x.sum()` / `y.copy_(x)` after k MB of intervening writes.

    python tools/lab/mall_probe.py
"""
import torch


def timed(fn, reps=1):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    n = 16448 * 1024
    x = torch.empty(n, device=dev, dtype=torch.float32)
    y = torch.empty_like(x)
    src = torch.randn(n, device=dev)
    filler = torch.empty(1 << 28, device=dev, dtype=torch.float32)      # 1 GiB
    for _ in range(3):
        x.copy_(src); x.sum(); y.copy_(x)
    torch.cuda.synchronize()
    for between_mb in (0, 64, 128, 192, 256, 384, 512, 1024):
        rs, rc = [], []
        for rep in range(7):
            for which in (0, 1):
                x.copy_(src)                                              # writes x (reads src)
                if between_mb:
                    filler[: between_mb * (1 << 18)].fill_(1.0)
                torch.cuda.synchronize()
                if which == 0:
                    rs.append(timed(lambda: x.sum()))
                else:
                    rc.append(timed(lambda: y.copy_(x)))
        rs.sort(); rc.sort()
        ms, mc = rs[len(rs) // 2], rc[len(rc) // 2]
        print(f"{between_mb:5d} MB written in between | x.sum() {ms:7.1f} us = {n * 4 / ms * 1e-6:6.2f} TB/s read | "
              f"y.copy_(x) {mc:7.1f} us = {n * 8 / mc * 1e-6:6.2f} TB/s read+write", flush=True)


if __name__ == "__main__":
    main()
