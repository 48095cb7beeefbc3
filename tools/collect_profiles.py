"""Copy the summaries tools/refresh_profiles.sh left in gpurun_out/ to their tracked names under profiles/ (round 5)."""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pairs = [
    ("r05_layers_isolated.txt", "r05_layers_isolated.txt"),
    ("r05_layers_in_update.txt", "r05_layers_in_update.txt"),
    ("r05_bench_pipelined_kernel_stats.csv", "r05_bench_kernel_stats_pipelined.csv"),
    ("r05_bench_join_kernel_stats.csv", "r05_bench_kernel_stats_join.csv"),
    ("r05_launch_count.txt", "r05_launch_count.txt"),
    ("pmc_dec3/summary.txt", "r05_pmc_dec3_kernels.txt"),
    ("pmc_convs/summary.txt", "r05_pmc_conv_layers.txt"),
    ("pmc_scan_rollout/summary.txt", "r05_pmc_scan_rollout.txt"),
    ("pmc_c3/summary.txt", "r05_pmc_3channel_layers.txt"),
    ("pmc_mlp/summary.txt", "r05_pmc_mlp_heads.txt"),
    ("pmc_wtr/summary.txt", "r05_pmc_head_wgrad.txt"),
    ("r05_heads_isolated.txt", "r05_heads_isolated.txt"),
    ("r05_bench_final.json", "r05_bench_final.json"),
    ("r05_bench_c4_c5.json", "r05_bench_c4_c5.json"),
    ("r05_bench_shards.json", "r05_bench_shards.json"),
    ("r05_lane_time.txt", "r05_lane_time.txt"),
    ("r05_phase_time.txt", "r05_phase_time.txt"),
    ("dominant_kernel_rocprof.json", "dominant_kernel_rocprof.json"),
    ("r05_scan_cs.txt", "r05_scan_cs.txt"),
    ("r05_layers_isolated_128.txt", "r05_layers_isolated_128.txt"),
    ("r05_c4x128_layers_in_update.txt", "r05_c4x128_layers_in_update.txt"),
    ("r05_c4x128_kernel_stats.csv", "r05_bench_kernel_stats_c4x128.csv"),
    ("r05_bench_c4x128.json", "r05_bench_c4x128.json"),
    ("r05_bench_tia.json", "r05_bench_tia.json"),
    ("r05_bench_mt.json", "r05_bench_mt.json"),
    ("r05_bgemm_probe.txt", "r05_bgemm_probe.txt"),
    ("r05_gemm_isolated.txt", "r05_gemm_isolated.txt"),
    ("r05_rollout_engines.txt", "r05_rollout_engines.txt"),
    ("r05_ab_vs_round4.txt", "r05_ab_vs_round4.txt"),
    ("r05_gpu_tests.txt", "r05_gpu_tests.txt"),
    ("r05_smoke.txt", "r05_smoke.txt"),
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
kern = {}
for fn in ("r05_pmc_dec3_kernels.txt", "r05_pmc_conv_layers.txt", "r05_pmc_3channel_layers.txt", "r05_pmc_scan_rollout.txt"):
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
        kern[name] = {"nimg": 2450, "traffic_bytes_per_launch": int(traffic.group(1)) * 1_000_000,
                      "fetch_bytes_corrected_x2": int(rdwr.group(1)) * 1e6 if rdwr else None,
                      "write_bytes": int(rdwr.group(2)) * 1e6 if rdwr else None,
                      "mfma_pipe_busy": float(busy.group(1)) if busy else None,
                      "effective_clock_ghz": float(clock.group(1)) if clock else None,
                      "duration_us_under_counters": float(dur.group(1)),
                      "source": f"profiles/{fn} (tools/pmc.sh: rocprofv3 --pmc passes, FETCH_SIZE x2 per MI355X_MICROARCH.md, "
                                "median of 5 dispatches of the kernel alone at 2450 frames)"}
print("pmc kernels:", len(kern))
if "--write-json" in sys.argv and kern:
    json.dump({"kernels": kern}, open(os.path.join(P, "dominant_kernel_pmc.json"), "w"), indent=1)
