"""Launches per update from a rocprofv3 kernel-stats CSV of bench.py (tools/prof_bench.sh): usage launch_count.py <csv> [updates]
(updates default: the trace's dual_step_kernel calls, one per RePo update)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {re.sub(r"\(.*", "", re.sub(r"repo::|void ", "", r["Name"])): int(r["Calls"]) for r in rows}
nupd = float(sys.argv[2]) if len(sys.argv) > 2 else float(by.get("dual_step_kernel", 0) or by.get("clip_adam_kernel", 0) / 3)
tot = 0.0
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    n = re.sub(r"\(.*", "", re.sub(r"repo::|void ", "", r["Name"]))[:70]
    c = int(r["Calls"]) / nupd
    tot += c
    if c >= 0.9:
        print("%6.1f %8.1f us  %s" % (c, float(r["AverageNs"]) / 1e3, n))
print("launches per update: %.1f" % tot)
