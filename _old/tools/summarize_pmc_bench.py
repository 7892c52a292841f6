#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh into profiles/<tag>_pmc_bench_traffic.{md,json}.
FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md §HBM);
WRITE_SIZE is taken as is.  Units of both counters: KiB."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def agg(counter):
    f = max(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_bench_{counter}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"]) * 1024.0
    return d


F, W = agg("FETCH_SIZE"), agg("WRITE_SIZE")
short = lambda k: re.sub(r"\(anonymous namespace\)::|^void ", "", k).split("(")[0]
forwards = next(n for k, (n, _) in F.items() if "final_conv_kernel" in k)
rows = sorted(((2 * f + W.get(k, [0, 0])[1], short(k), n, 2 * f, W.get(k, [0, 0])[1]) for k, (n, f) in F.items()), reverse=True)
is_gemm = lambda n: (n.startswith("linear_kernel") and ", true," not in n.split("<")[1][:24]) or "gemm_ws_kernel" in n or "gemm_ws64_kernel" in n
gemm = [r for r in rows if is_gemm(r[1]) or r[1].startswith("splitk_reduce")]
# ABI launches of mumpy_linear_wsz_fwd + mumpy_linear_rows_fwd per forward = non-conv linear_kernel + gemm_ws_kernel dispatches
abi_launches = sum(r[2] for r in gemm if is_gemm(r[1]))
gemm_bytes = sum(r[3] + r[4] for r in gemm)
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline --no-alt --no-graph, MUMPY_SERIAL=1; FETCH_SIZE x2 (gfx950 correction)",
       "forwards_in_profile": forwards,
       "mumpy_linear_wsz_fwd": {"launches": abi_launches, "hbm_bytes_per_launch": round(gemm_bytes / abi_launches),
                               "read_bytes_per_launch": round(sum(r[3] for r in gemm) / abi_launches),
                               "write_bytes_per_launch": round(sum(r[4] for r in gemm) / abi_launches)},
       "per_forward_hbm_bytes_all_kernels": round(sum(r[0] for r in rows) / forwards)}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_bench_traffic.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_bench_traffic.md"), "w") as f:
    f.write(f"# HBM-side traffic of the bench workload by kernel (PMC)\n\n{out['source']}.\n{forwards} forwards of B=8, T=5 in the run; "
            f"all kernels together move {out['per_forward_hbm_bytes_all_kernels'] / 1e9:.2f} GB per forward.\n\n"
            "| kernel | dispatches | read MB / dispatch (FETCH_SIZE x2) | written MB / dispatch | share of all traffic |\n|---|---:|---:|---:|---:|\n")
    tot = sum(r[0] for r in rows)
    for r in rows[:24]:
        f.write(f"| `{r[1][:80]}` | {r[2]} | {r[3] / r[2] / 1e6:.2f} | {r[4] / r[2] / 1e6:.2f} | {100 * r[0] / tot:.1f} % |\n")
    g = out["mumpy_linear_wsz_fwd"]
    f.write(f"\nDominant ABI entry `mumpy_linear_wsz_fwd` (+`_rows_fwd`; its `gemm_ws_kernel`, `linear_kernel` and `splitk_reduce_kernel` dispatches): "
            f"{g['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch ({g['read_bytes_per_launch'] / 1e6:.1f} read + "
            f"{g['write_bytes_per_launch'] / 1e6:.1f} written) -- this is `roofline.traffic` of `bench.py`.\n")
print(json.dumps(out, indent=1))
