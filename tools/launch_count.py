"""Launches per update from a rocprofv3 kernel-stats CSV of bench.py (tools/prof_bench.sh): usage launch_count.py <csv> <updates>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nupd = float(sys.argv[2])
tot = 0.0
for r in sorted(rows, key=lambda r: -int(r["Calls"])):
    n = re.sub(r"\(.*", "", re.sub(r"repo::|void ", "", r["Name"]))[:70]
    c = int(r["Calls"]) / nupd
    tot += c
    if c >= 0.9:
        print("%6.1f %8.1f us  %s" % (c, float(r["AverageNs"]) / 1e3, n))
print("launches per update: %.1f" % tot)
