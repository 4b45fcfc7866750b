#!/usr/bin/env python3
"""profiles/fc1_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, stamped with the sha256 of
the GEMM sources (gemm.hip + gemm_args.h + common.h) they were measured on, the workload and the tile form (bench.py drops the figure when any of them differs).
    python tools/make_fc1_traffic.py gpurun_out/<tag>_pmc_fetch.txt gpurun_out/<tag>_pmc_write.txt <tag>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEMM_SOURCES = ("gemm.hip", "gemm_args.h", "common.h")       # as bench.py: everything the GEMM kernels are compiled from
FC1 = re.compile(r"gemm_kernel<(\d+), (\d+), \d, \d, 83, (true|false)")      # LN_IN|BIAS|GELU|OUT_BF16 = 64+1+2+16


def per_launch(path, counter):
    best = None
    name = None
    for line in open(path):
        if not line.startswith(" "):
            name = line
        elif counter in line and name and FC1.search(name):
            n = int(re.search(r"dispatches (\d+)", name).group(1))
            v = float(line.split()[-1])
            if best is None or n > best[1]:
                best = (v, n, FC1.search(name))
    if best is None:
        raise SystemExit(f"no fc1 kernel with {counter} in {path}")
    return best


def main():
    fetch, write = sys.argv[1], sys.argv[2]
    if len(sys.argv) <= 3:
        raise SystemExit("make_fc1_traffic.py: name the round (third argument, e.g. r05)")
    tag = sys.argv[3]
    f_kb, n, m = per_launch(fetch, "FETCH_SIZE")
    w_kb, _, _ = per_launch(write, "WRITE_SIZE")
    M, N, K = 64 * 257, 4096, 1024
    h = hashlib.sha256()
    for f in GEMM_SOURCES:
        h.update(open(os.path.join(ROOT, "uspace_amd", "csrc", f), "rb").read())
    sha = h.hexdigest()
    out = {
        "kernel": f"gemm_kernel<{m.group(1)},{m.group(2)},LN_IN|BIAS|GELU|OUT_BF16> (fc1, M={M} N={N} K={K})",
        "model": "L_u", "batch": 64, "tile": [int(m.group(1)), int(m.group(2))], "gemm_source_sha256": sha, "gemm_sources": list(GEMM_SOURCES),
        "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh {tag}) over "
                  f"tools/one_forward.py ({n} launches)",
        "fetch_size_kb_per_launch": f_kb, "write_size_kb_per_launch": w_kb,
        "correction": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads (MI355X_MICROARCH.md, HBM section): "
                      "doubled; WRITE_SIZE taken as is (calibrated in round 2 on layernorm_kernel)",
        "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
        "algorithmic_bytes_per_launch": 2.0 * M * K + 2.0 * N * K + 2.0 * M * N,
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "fc1_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
