#!/usr/bin/env python3
"""Copy the outputs of tools/final_profiles.sh from gpurun_out/ (scratch) into profiles/<tag>_* (tracked).  usage: collect_profiles.py r03"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
pairs = {
    "bench_final.json": "final_bench.json", "bench_t9.json": "bench_t9.json", "bench_b16.json": "bench_b16.json", "bench_b1.json": "bench_b1.json",
    "stats_serial.csv": "final_serial_kernel_stats.csv", "stats_fj.csv": "final_kernel_stats.csv", "stats_s16.csv": "bf16_storage_serial_kernel_stats.csv",
    "summary_serial.md": "final_serial_summary.md", "summary_fj.md": "final_summary.md", "summary_s16.md": "bf16_storage_serial_summary.md",
    "timeline_gaps.txt": "timeline_gaps.txt", "phase_timeline.txt": "phase_timeline.txt", "gemm_shapes.txt": "gemm_shapes.txt",
    "ln_fold_shapes.txt": "ln_fold_shapes_final.txt", "summary_train.md": "train_b2_final.md", "train_aten_ops.txt": "train_aten_ops.txt",
}
for src, dst in pairs.items():
    s = os.path.join(G, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copyfile(s, os.path.join(P, f"{tag}_{dst}"))
        print("copied", src, "->", f"{tag}_{dst}")
    else:
        print("MISSING", src)
lines = {}
for name in ("train_b2_eager", "train_b2_graph", "train_b2_graph_bf16", "train_b2_graph_bf16_trainmode", "train_b8_eager"):
    f = os.path.join(G, name + ".json")
    if os.path.exists(f) and os.path.getsize(f):
        lines[name] = json.loads(open(f).read().strip().splitlines()[-1])
if lines:
    json.dump(lines, open(os.path.join(P, f"{tag}_train_lines.json"), "w"), indent=1)
    print("train lines:", {k: v["ms_per_step"] for k, v in lines.items()})
if os.path.isdir(os.path.join(G, "pmc_bench_FETCH_SIZE")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc_bench.py"), tag], check=False)
