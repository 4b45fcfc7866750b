#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 (rocpd sqlite) counter-collection run."""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, out=None):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    # columns differ slightly across versions; discover
    q = "select * from pmc_events limit 1"
    row = c.execute(q).fetchone()
    names = [d[0] for d in c.execute(q).description]
    kcol = next(n for n in names if n in ("name", "kernel_name"))
    ccol = next(n for n in names if n in ("counter_name", "pmc_name", "symbol"))
    vcol = next(n for n in names if n in ("value", "counter_value"))
    dcol = next(n for n in names if n in ("dispatch_id", "event_id", "id"))
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for k, cn, v, d in c.execute(f"select {kcol}, {ccol}, {vcol}, {dcol} from pmc_events"):
        agg[k][cn] += v
        cnt[k].add(d)
    lines = []
    for k in agg:
        n = max(1, len(cnt[k]))
        short = re.sub(r"\(anonymous namespace\)::", "", k)[:90]
        vals = {cn: v / n for cn, v in agg[k].items()}
        lines.append((short, n, vals))
    lines.sort(key=lambda t: -t[2].get("SQ_BUSY_CYCLES", 0))
    txt = []
    for short, n, vals in lines:
        txt.append(f"{short}  (dispatches {n})")
        for cn in sorted(vals):
            txt.append(f"    {cn:32s} {vals[cn]:18.0f}")
        wc = vals.get("SQ_WAVE_CYCLES")
        if wc:
            for cn in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY"):
                if cn in vals:
                    txt.append(f"    {cn + ' / WAVE_CYCLES':32s} {vals[cn] / wc:18.3f}")
        if "SQ_LDS_IDX_ACTIVE" in vals and vals["SQ_LDS_IDX_ACTIVE"]:
            txt.append(f"    {'BANK_CONFLICT / LDS_IDX_ACTIVE':32s} {vals.get('SQ_LDS_BANK_CONFLICT', 0) / vals['SQ_LDS_IDX_ACTIVE']:18.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and vals.get("SQ_BUSY_CYCLES"):
            txt.append(f"    {'MFMA_BUSY / BUSY_CYCLES':32s} {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / vals['SQ_BUSY_CYCLES']:18.3f}")
    s = "\n".join(txt)
    if out:
        open(out, "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
