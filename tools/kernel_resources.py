#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use of the built library, read from the code objects embedded in libuspace_hip.so
(no GPU needed):  python tools/kernel_resources.py [--all]

Used by tests/test_host_logic.py to pin two invariants of the hot kernels: no scratch (a spill in a GEMM or attention
kernel is a silent 10-30 % loss -- it happened twice while the 256x128 tile form was being added) and at most 256 registers
for the kernels that share a SIMD between two waves."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=os.path.join(ROOT, "uspace_amd", "libuspace_hip.so")):
    """[{name, vgpr, agpr, sgpr, scratch, lds, wg}] for every device kernel of the library."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))       # llvm-objdump writes the bundles next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True,
                                   capture_output=True, text=True).stdout
            # the kernel records of the metadata are list items of amdhsa.kernels whose first key is .agpr_count
            for rec in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                rec = ".agpr_count:" + rec
                k = {}
                for key in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                            "max_flat_workgroup_size"):
                    m = re.search(r"\." + key + r":\s+(\d+)", rec)
                    if m:
                        k[key] = int(m.group(1))
                m = re.search(r"\n\s+\.name:\s+(\S+)", rec)
                if m:
                    k["name"] = m.group(1)
                out.append(k)
    res = []
    for k in out:
        if "name" in k and "vgpr_count" in k:
            res.append(dict(name=k["name"], vgpr=k.get("vgpr_count", 0), agpr=k.get("agpr_count", 0), sgpr=k.get("sgpr_count", 0),
                            scratch=k.get("private_segment_fixed_size", 0), lds=k.get("group_segment_fixed_size", 0),
                            wg=k.get("max_flat_workgroup_size", 0)))
    return res


def demangle(names):
    filt = shutil.which("c++filt")
    if not filt:
        return list(names)
    p = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True)
    return p.stdout.splitlines()


if __name__ == "__main__":
    ks = kernels()
    names = demangle([k["name"] for k in ks])
    show_all = "--all" in sys.argv
    print(f"{len(ks)} kernels; with scratch: {sum(1 for k in ks if k['scratch'])}")
    for k, n in sorted(zip(ks, names), key=lambda x: x[1]):
        if show_all or k["scratch"] or k["vgpr"] + k["agpr"] > 256:
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            print(f"{k['vgpr']:4d} v {k['agpr']:4d} a {k['scratch']:5d} B scratch {k['lds']:7d} B LDS  {n[:110]}")
