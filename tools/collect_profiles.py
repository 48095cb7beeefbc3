"""Copy the summaries tools/refresh_profiles.sh left in gpurun_out/ to their tracked names under profiles/ (round 4)."""
import json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
pairs = [
    ("r04_layers_isolated.txt", "r04_layers_isolated.txt"),
    ("r04_layers_in_update.txt", "r04_layers_in_update.txt"),
    ("r04_bench_pipelined_kernel_stats.csv", "r04_bench_kernel_stats_pipelined.csv"),
    ("r04_bench_join_kernel_stats.csv", "r04_bench_kernel_stats_join.csv"),
    ("r04_launch_count.txt", "r04_launch_count.txt"),
    ("pmc_dec3/summary.txt", "r04_pmc_dec3_kernels.txt"),
    ("pmc_convs/summary.txt", "r04_pmc_conv_layers.txt"),
    ("pmc_scan_rollout/summary.txt", "r04_pmc_scan_rollout.txt"),
    ("pmc_c3/summary.txt", "r04_pmc_3channel_layers.txt"),
    ("pmc_mlp/summary.txt", "r04_pmc_mlp_heads.txt"),
    ("r04_bench_final.json", "r04_bench_final.json"),
    ("r04_bench_c4_c5.json", "r04_bench_c4_c5.json"),
    ("r04_bench_shards.json", "r04_bench_shards.json"),
    ("r04_lane_time.txt", "r04_lane_time.txt"),
    ("r04_phase_time.txt", "r04_phase_time.txt"),
    ("dominant_kernel_rocprof.json", "dominant_kernel_rocprof.json"),
    ("r04_scan_cs.txt", "r04_scan_cs.txt"),
    ("r04_layers_isolated_128.txt", "r04_layers_isolated_128.txt"),
    ("r04_c4x128_layers_in_update.txt", "r04_c4x128_layers_in_update.txt"),
    ("r04_c4x128_kernel_stats.csv", "r04_bench_kernel_stats_c4x128.csv"),
    ("r04_bench_c4x128.json", "r04_bench_c4x128.json"),
    ("r04_bench_tia.json", "r04_bench_tia.json"),
    ("r04_bench_mt.json", "r04_bench_mt.json"),
    ("r04_bgemm_probe.txt", "r04_bgemm_probe.txt"),
    ("r04_gemm_isolated.txt", "r04_gemm_isolated.txt"),
    ("r04_rollout_engines.txt", "r04_rollout_engines.txt"),
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
# the dominant kernel's counters, as bench.py quotes them
txt = open(os.path.join(P, "r04_pmc_dec3_kernels.txt")).read()
blk = re.search(r"buconv_scatter_kernel.*?(?=\nvoid |\Z)", txt, re.S)
if blk:
    b = blk.group(0)
    traffic = re.search(r"= (\d+) MB \(", b)
    busy = re.search(r"MFMA pipe busy ([0-9.]+)", b)
    clock = re.search(r"effective clock ([0-9.]+)", b)
    dur = re.search(r"\(median\)\s+([0-9.]+) us", b)
    j = {"kernel": "buconv_scatter_kernel<GDec3>", "traffic_bytes_per_launch": int(traffic.group(1)) * 1_000_000,
         "mfma_pipe_busy": float(busy.group(1)), "effective_clock_ghz": float(clock.group(1)),
         "duration_us_under_counters": float(dur.group(1)),
         "source": "profiles/r04_pmc_dec3_kernels.txt (tools/pmc.sh dec3 ... tools/run_micro_case.py 'conv dec3': rocprofv3 --pmc passes, FETCH_SIZE x2 per MI355X_MICROARCH.md, median of 5 dispatches)"}
    old = json.load(open(os.path.join(P, "dominant_kernel_pmc.json")))
    print("dominant kernel pmc: old", {k: old.get(k) for k in j if k != "source"})
    print("dominant kernel pmc: new", {k: j[k] for k in j if k != "source"})
    if "--write-json" in sys.argv:
        old.update(j)
        json.dump(old, open(os.path.join(P, "dominant_kernel_pmc.json"), "w"), indent=1)
