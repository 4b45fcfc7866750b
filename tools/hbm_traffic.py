#!/usr/bin/env python3
"""profiles/<tag>_hbm_traffic.md from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (U-ViT-L, B = 64):
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 tallies wide coalesced reads at half their size,
MI355X_MICROARCH.md, HBM section) against the algorithmic bytes of the launch (operands once + outputs once).
    python tools/hbm_traffic.py gpurun_out/<tag>_pmc_fetch.txt gpurun_out/<tag>_pmc_write.txt profiles/<tag>_hbm_traffic.md [tag]
(the tag in the title and in the command the file quotes defaults to the leading rNN of the output file's name)"""
import os
import re
import sys

M, D = 64 * 257, 1024
KERNELS = [   # (regex on the kernel name, label, algorithmic bytes)
    # round 4: the in-blocks' fc2 no longer writes a raw bf16 skip (the centred copy IS the skip), so it runs the same instantiation as
    # proj (flags 45): per forward 21 proj + 10 fc2 launches share one counter row -- their launch-weighted mean is what is compared
    (r"gemm_kernel<256, 256, 2, 4, 45,", "proj GEMM (K = D) and in-block fc2 GEMM (K = 4D): +bias +residual fp32 in place + centred bf16 copy + row partial sums; mean of 21 + 10 launches",
     (21 * (2 * M * D + 2 * D * D + 8 * M * D + 2 * M * D) + 10 * (2 * M * 4 * D + 2 * D * 4 * D + 8 * M * D + 2 * M * D)) // 31),
    (r"gemm_kernel<256, 256, 2, 4, 29,", "fc2 GEMM mid/out-blocks (+bias +residual, fp32 + raw bf16)", 2 * M * 4 * D + 2 * D * 4 * D + 8 * M * D + 2 * M * D),
    (r"gemm_kernel<256, 256, 2, 4, 83,", "fc1 GEMM (norm2 folded in, +bias +GELU -> bf16)", 2 * M * D + 2 * 4 * D * D + 2 * M * 4 * D),
    (r"gemm_kernel<256, 256, 2, 4, 81,", "qkv GEMM (norm1 folded in, -> bf16)", 2 * M * D + 2 * 3 * D * D + 2 * M * 3 * D),
    (r"gemm_kernel<256, 256, 2, 4, 169,", "skip GEMM (two K slabs, second one the centred skip + rank-1 term; fp32 out + centred bf16 copy)", 2 * M * 2 * D + 2 * D * 2 * D + 4 * M * D + 2 * M * D),
    (r"attention_kernel<17, 257", "attention (q, k, v read; out written)", 2 * M * 3 * D + 2 * M * D),
]


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line
        elif counter in line and name:
            out[name] = (float(line.split()[-1]), int(re.search(r"dispatches (\d+)", name).group(1)))
    return out


def main():
    fetch, write, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    m = re.match(r"(r\d+)_", os.path.basename(dst))
    if len(sys.argv) <= 4 and not m:
        raise SystemExit("hbm_traffic.py: name the round (4th argument) or write to profiles/rNN_hbm_traffic.md")
    tag = sys.argv[4] if len(sys.argv) > 4 else m.group(1)
    f, w = parse(fetch, "FETCH_SIZE"), parse(write, "WRITE_SIZE")
    rows = []
    for rx, label, alg in KERNELS:
        fk = next((k for k in f if re.search(rx, k)), None)
        wk = next((k for k in w if re.search(rx, k)), None)
        if not fk or not wk:
            continue
        fm, wm = f[fk][0] * 1024 / 1e6, w[wk][0] * 1024 / 1e6
        hbm = 2 * fm + wm
        rows.append(f"| {label} | {f[fk][1]} | {fm:.1f} | {wm:.1f} | {hbm:.1f} | {alg / 1e6:.1f} | {hbm / (alg / 1e6):.2f} |")
    txt = (f"# HBM traffic per launch, {tag} (U-ViT-L, B=64, LayerNorm folded into the GEMMs)\n\n"
           f"`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes over `tools/one_forward.py --reps 3`\n"
           f"(`tools/profile_round.sh {tag}`; raw per-kernel averages in `{tag}_pmc_fetch.txt` / `{tag}_pmc_write.txt`).  `FETCH_SIZE` on gfx950 tallies wide\n"
           "coalesced reads at half their size (MI355X_MICROARCH.md, HBM section; calibrated in round 1 on LayerNorm), so HBM bytes =\n"
           "2 x FETCH_SIZE + WRITE_SIZE (KB -> x 1024).  The read side counts fabric (L2-miss) requests including Infinity-Cache hits.\n\n"
           "| kernel | launches | FETCH_SIZE MB (raw) | WRITE_SIZE MB | HBM MB (corrected) | algorithmic MB | ratio |\n|---|---:|---:|---:|---:|---:|---:|\n"
           + "\n".join(rows) + "\n")
    open(dst, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
