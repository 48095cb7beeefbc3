"""Copy the summaries tools/refresh_profiles.sh left in gpurun_out/ to their tracked names under profiles/ (RND=r06 by default)."""
import json, os, re, shutil, sys
RND = os.environ.get("RND", "r06")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pairs = [
    (f"{RND}_layers_isolated.txt", f"{RND}_layers_isolated.txt"),
    (f"{RND}_layers_in_update.txt", f"{RND}_layers_in_update.txt"),
    (f"{RND}_bench_pipelined_kernel_stats.csv", f"{RND}_bench_kernel_stats_pipelined.csv"),
    (f"{RND}_bench_join_kernel_stats.csv", f"{RND}_bench_kernel_stats_join.csv"),
    (f"{RND}_launch_count.txt", f"{RND}_launch_count.txt"),
    ("pmc_dec3/summary.txt", f"{RND}_pmc_dec3_kernels.txt"),
    ("pmc_convs/summary.txt", f"{RND}_pmc_conv_layers.txt"),
    ("pmc_scan_rollout/summary.txt", f"{RND}_pmc_scan_rollout.txt"),
    ("pmc_c3/summary.txt", f"{RND}_pmc_3channel_layers.txt"),
    ("pmc_mlp/summary.txt", f"{RND}_pmc_mlp_heads.txt"),
    ("pmc_wtr/summary.txt", f"{RND}_pmc_head_wgrad.txt"),
    (f"{RND}_heads_isolated.txt", f"{RND}_heads_isolated.txt"),
    (f"{RND}_bench_final.json", f"{RND}_bench_final.json"),
    (f"{RND}_bench_c4_c5.json", f"{RND}_bench_c4_c5.json"),
    (f"{RND}_bench_shards.json", f"{RND}_bench_shards.json"),
    (f"{RND}_lane_time.txt", f"{RND}_lane_time.txt"),
    (f"{RND}_phase_time.txt", f"{RND}_phase_time.txt"),
    ("dominant_kernel_rocprof.json", "dominant_kernel_rocprof.json"),
    (f"{RND}_scan_cs.txt", f"{RND}_scan_cs.txt"),
    (f"{RND}_layers_isolated_128.txt", f"{RND}_layers_isolated_128.txt"),
    (f"{RND}_c4x128_layers_in_update.txt", f"{RND}_c4x128_layers_in_update.txt"),
    (f"{RND}_c4x128_kernel_stats.csv", f"{RND}_bench_kernel_stats_c4x128.csv"),
    (f"{RND}_bench_c4x128.json", f"{RND}_bench_c4x128.json"),
    (f"{RND}_bench_tia.json", f"{RND}_bench_tia.json"),
    (f"{RND}_bench_mt.json", f"{RND}_bench_mt.json"),
    (f"{RND}_bgemm_probe.txt", f"{RND}_bgemm_probe.txt"),
    (f"{RND}_gemm_isolated.txt", f"{RND}_gemm_isolated.txt"),
    (f"{RND}_rollout_engines.txt", f"{RND}_rollout_engines.txt"),
    (f"{RND}_ab_vs_round5.txt", f"{RND}_ab_vs_round5.txt"),
    (f"{RND}_gpu_tests.txt", f"{RND}_gpu_tests.txt"),
    (f"{RND}_smoke.txt", f"{RND}_smoke.txt"),
]
for src, dst in pairs:
    s = os.path.join(G, src)
    if not os.path.exists(s):
        print("missing", src)
        continue
    text = open(s).read()
    text = "\n".join(l for l in text.splitlines() if "amdgpu.ids" not in l) + "\n"
    open(os.path.join(P, dst), "w").write(text)
    print("wrote", dst, len(text))
# the counters bench.py quotes for the kernels its `roofline` can name: every conv / rollout / scan kernel block of the
# committed --pmc summaries, keyed by kernel name
# the problem each summary file's kernels ran on (bench.py quotes the counters only for a run of the same size: its
# `nimg` = frames per update = T * B = rows of the rollout, 2450 at B=50 L=50)
_CONV = "conv layers on 2450 frames (B=50, L=50)"
PROBLEM = {f"{RND}_pmc_dec3_kernels.txt": _CONV, f"{RND}_pmc_conv_layers.txt": _CONV, f"{RND}_pmc_3channel_layers.txt": _CONV,
           f"{RND}_pmc_scan_rollout.txt": "observe scans: T=49 steps x B=50 rows; rollout: N=2450 start rows x 14 steps, A=6"}
kern = {}
for fn in (f"{RND}_pmc_dec3_kernels.txt", f"{RND}_pmc_conv_layers.txt", f"{RND}_pmc_3channel_layers.txt", f"{RND}_pmc_scan_rollout.txt"):
    path = os.path.join(P, fn)
    if not os.path.exists(path):
        continue
    txt = open(path).read()
    for blk in re.split(r"\n(?=void |repo::|[a-z_0-9]+_kernel)", "\n" + txt):
        lines = blk.strip().splitlines()
        if not lines:
            continue
        name = re.sub(r"^void ", "", lines[0]).strip()
        traffic = re.search(r"= (\d+) MB \(", blk)
        rdwr = re.search(r"per launch (\d+) MB read \(x2 corrected\) \+ (\d+) MB written", blk)
        busy = re.search(r"MFMA pipe busy ([0-9.]+)", blk)
        clock = re.search(r"effective clock ([0-9.]+)", blk)
        dur = re.search(r"\(median\)\s+([0-9.]+) us", blk)
        if not (traffic and dur):
            continue
        kern[name] = {"nimg": 2450, "problem": PROBLEM[fn], "traffic_bytes_per_launch": int(traffic.group(1)) * 1_000_000,
                      "fetch_bytes_corrected_x2": int(rdwr.group(1)) * 1e6 if rdwr else None,
                      "write_bytes": int(rdwr.group(2)) * 1e6 if rdwr else None,
                      "mfma_pipe_busy": float(busy.group(1)) if busy else None,
                      "effective_clock_ghz": float(clock.group(1)) if clock else None,
                      "duration_us_under_counters": float(dur.group(1)),
                      "source": f"profiles/{fn} (tools/pmc.sh: rocprofv3 --pmc passes, FETCH_SIZE x2 per MI355X_MICROARCH.md, "
                                f"median of 5 dispatches of the kernel alone; {PROBLEM[fn]})"}
print("pmc kernels:", len(kern))
if "--write-json" in sys.argv and kern:
    json.dump({"kernels": kern}, open(os.path.join(P, "dominant_kernel_pmc.json"), "w"), indent=1)
