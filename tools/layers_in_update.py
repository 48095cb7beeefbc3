#!/usr/bin/env python3
"""Per-kernel table of the update AS IT RUNS (both lanes and the side streams sharing the GPU), from the
rocprofv3 --kernel-trace --stats CSV of `bench.py --steps K` (tools/prof_bench.sh): calls per update, average
duration, and -- for the conv kernels, whose FLOPs follow from the geometry in the kernel's name -- TFLOP/s and
the fraction of the fp32-MFMA peak at that in-update duration.  Next to profiles/rNN_layers_isolated.txt this
shows what concurrency costs each kernel.

    python tools/layers_in_update.py <kernel_stats.csv> [updates in the trace] [--json profiles/dominant_kernel_rocprof.json]
"""
import csv
import json
import re
import sys

PEAK = 157.3
NIMG = 2450   # --nimg N: frames per update of the traced config (c4 / c4x128: 1568)


def conv_flop(name):
    m = re.search(r"Geo<(\d+), (\d+), (\d+), (\d+)>", name)
    if not m and name.startswith("tconv_up_kernel"):   # encoder conv2's data gradient in gather form: no Geo<> in the name
        m = re.search(r"(\d+), (\d+), (\d+), (\d+)", "32, 64, 31, 4")
    if not m and name.startswith("tconv_down_kernel"):   # TcdGeoT<KS, WB, WS, ..>: decoder conv3's data gradient / encoder conv2's forward
        t = re.search(r"TcdGeoT<(\d+), (\d+)", name)
        m = re.search(r"(\d+), (\d+), (\d+), (\d+)", "32, 64, %s, %s" % ((t.group(2), t.group(1)) if t else ("30", "6")))
    if not m:
        return None
    cb, cs, hb, ks = map(int, m.groups())
    hs = (hb - ks) // 2 + 1
    return 2.0 * NIMG * cs * hs * hs * cb * ks * ks


def main():
    global NIMG
    if "--nimg" in sys.argv:
        NIMG = int(sys.argv[sys.argv.index("--nimg") + 1])
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    # updates in the trace: one dual step (RePo) / three clip+Adam steps per update
    by = {re.sub(r"\(.*", "", re.sub(r"repo::|void ", "", r["Name"])): int(r["Calls"]) for r in rows}
    # one lambda-return launch per update in every algorithm
    nupd = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].replace(".", "").isdigit() else float(
        by.get("lambda_return_kernel", 0) or by.get("dual_step_kernel", 0) or by.get("clip_adam_kernel", 0) / 3)
    tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# kernels of the update as it runs (rocprofv3 --kernel-trace --stats of bench.py, {nupd:.0f} updates in the trace)")
    print(f"# sum of kernel time per update: {tot_ns / nupd / 1e6:.2f} ms")
    print(f"# {'calls/upd':>9} {'avg us':>9} {'us/upd':>9} {'TFLOP/s':>8} {'frac':>6}  kernel")
    dom, top, first = None, None, True
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        name = re.sub(r"\(.*", "", re.sub(r"repo::|void ", "", r["Name"]))
        calls = int(r["Calls"]) / nupd
        avg = float(r["AverageNs"]) / 1e3
        if calls < 0.5 or float(r["TotalDurationNs"]) / nupd < 2e3:
            continue
        fl = conv_flop(name) if ("conv" in name and "pack" not in name and "reduce" not in name) else None
        tf = fl / (avg * 1e-6) / 1e12 if fl else None
        print(f"  {calls:9.1f} {avg:9.1f} {float(r['TotalDurationNs']) / nupd / 1e3:9.1f} "
              f"{(f'{tf:8.1f}' if tf else '       -')} {(f'{tf / PEAK:6.3f}' if tf else '     -')}  {name[:100]}")
        row = {"name": name, "nimg": NIMG, "avg_ms_in_update": round(avg / 1e3, 4), "calls_per_update": round(calls, 2),
               "total_ms_per_update": round(float(r["TotalDurationNs"]) / nupd / 1e6, 4),
               "share_of_kernel_time": round(float(r["TotalDurationNs"]) / tot_ns, 4),
               "frac_of_fp32_peak_at_rocprof_duration": round(tf / PEAK, 4) if tf else None}
        if first:   # the first row by total time: what bench.py's `roofline` names and times live
            top, first = row, False
        if name.startswith("buconv_scatter_kernel<Geo<32, 64, 30, 6>") or name.startswith("uconv_scatter_kernel<Geo<32, 64, 30, 6>"):
            dom = row
    if "--json" in sys.argv and top:
        out = sys.argv[sys.argv.index("--json") + 1]
        csvname = sys.argv[sys.argv.index("--csv-name") + 1] if "--csv-name" in sys.argv else path
        summary = {"top_by_time": top, "largest_launch": dom, "nimg": NIMG,
                   "kernel_time_sum_ms_per_update": round(tot_ns / nupd / 1e6, 3), "updates_in_trace": nupd, "csv": csvname,
                   "source": ("rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 "
                              f"(tools/prof_bench.sh; {nupd:.0f} updates in the trace: warm-up, timed, resident-batch and "
                              "pinned-path loops; the isolated re-runs of the timed kernels are skipped under the profiler)")}
        json.dump(summary, open(out, "w"), indent=1)
        print("# wrote", out, summary["top_by_time"])


if __name__ == "__main__":
    main()
