#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace csv: dispatches that ran on fewer workgroups than the chip has CUs and still took long -- the
latency-bound suspects (a long dependent chain in too few threads).  usage: small_grid_kernels.py <kernel_trace.csv> [min_us] [max_blocks]"""
import collections, csv, re, sys
f, min_us, max_blocks = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 10.0, int(sys.argv[3]) if len(sys.argv) > 3 else 256
rows = collections.defaultdict(lambda: [0, 0.0])
total = collections.Counter()
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    blocks = grid // max(wg, 1)
    total[name] += 1
    if blocks < max_blocks and dur >= min_us:
        k = (name, blocks, wg)
        rows[k][0] += 1
        rows[k][1] += dur
print(f"dispatches with < {max_blocks} workgroups and >= {min_us} us:")
print("| kernel | workgroups x threads | dispatches | avg us | total ms |\n|---|---:|---:|---:|---:|")
for (name, blocks, wg), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| `{name}` | {blocks} x {wg} | {n} | {t / n:.1f} | {t / 1e3:.3f} |")
