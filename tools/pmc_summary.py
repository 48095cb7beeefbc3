#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc passes (tools/pmc.sh).  usage: pmc_summary.py <name-regex> <pass dirs...>

Derived (MI355X_MICROARCH.md): effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA pipe busy =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); HBM bytes = 2 * FETCH_SIZE (gfx950 reports half of
wide coalesced reads) + WRITE_SIZE, both reported in KiB."""
import csv
import glob
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
pat = re.compile(sys.argv[1])
vals = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [per dispatch]
durs = defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not pat.search(n):
                continue
            short = re.sub(r"\(.*", "", re.sub(r"repo::", "", n))[:110]
            key = (short, r["Dispatch_Id"])
            per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        for (short, _), c in per.items():
            for k, v in c.items():
                (durs[short] if k == "_dur" else vals[short][k]).append(v)


def med(x):
    x = sorted(x)
    return x[len(x) // 2] if x else float("nan")


for k in sorted(vals):
    c = {n: med(v) for n, v in vals[k].items()}
    dur = med(durs[k])
    print(f"{k}\n    dispatches {len(durs[k]) // max(len(sys.argv) - 2, 1)}   duration under the counters (median) {dur:9.1f} us")
    for n in sorted(c):
        print(f"    {n:28s} {c[n]:16.0f}")
    g = c.get("GRBM_GUI_ACTIVE")
    if g:
        print(f"    -> effective clock {g / 8 / dur / 1e3:.2f} GHz")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            print(f"    -> MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g / 8):.3f}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        print(f"    -> HBM traffic per launch {rd / 1e6:.0f} MB read (x2 corrected) + {wr / 1e6:.0f} MB written = {(rd + wr) / 1e6:.0f} MB"
              f" ({(rd + wr) / dur / 1e6:.2f} TB/s)")
    if "SQ_WAVE_CYCLES" in c:
        w = c["SQ_WAVE_CYCLES"]
        print("    -> wave time: " + ", ".join(f"{n[3:].lower()} {c[n] / w:.2f}" for n in
                                               ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n in c))
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"    -> LDS bank-conflict cycles / LDS active cycles {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.2f}")
    if "TCC_HIT" in c and "TCC_MISS" in c:
        print(f"    -> L2 hit rate {c['TCC_HIT'] / (c['TCC_HIT'] + c['TCC_MISS']):.3f}")
