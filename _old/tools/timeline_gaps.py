#!/usr/bin/env python3
"""GPU busy/idle analysis of a rocprofv3 --kernel-trace of bench.py: for the last replayed steps, the union of kernel
intervals vs the step span, the concurrency histogram, and the largest idle gaps with the kernels around them.
usage: python tools/timeline_gaps.py <dir with *_kernel_trace.csv> [n_steps]"""
import csv, glob, sys, re
d = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
short = lambda k: re.sub(r"\(anonymous namespace\)::|^void ", "", k).split("(")[0][:48]
# steps are delimited by final_conv_kernel (last kernel of a forward)
ends = [i for i, r in enumerate(rows) if "final_conv_kernel" in r[2]]
spans = [(rows[b][1] - rows[a + 1][0], a, b) for a, b in zip(ends[:-1], ends[1:])]
print("step spans (ms):", [round(x[0] / 1e6, 1) for x in spans])
best = sorted(spans)[:nsteps]                        # the graph replays are the shortest steps of the run
tot_span = tot_busy = 0
gaps = []
conc = {}
for _, a, b in best:
    ks = rows[a + 1:b + 1]
    t0, t1 = ks[0][0], max(k[1] for k in ks)
    ev = sorted([(k[0], 1) for k in ks] + [(k[1], -1) for k in ks])
    n = 0; last = t0; busy = 0
    for t, dlt in ev:
        conc[min(n, 4)] = conc.get(min(n, 4), 0) + (t - last)
        if n > 0: busy += t - last
        n += dlt; last = t
    tot_span += t1 - t0; tot_busy += busy
    # idle gaps: sweep
    cur_end = ks[0][1]; prev = ks[0]
    for k in sorted(ks):
        if k[0] > cur_end:
            gaps.append((k[0] - cur_end, short(prev[2]), short(k[2])))
        if k[1] > cur_end: cur_end = k[1]; prev = k
print(f"{len(ends)-1} steps: span {tot_span/1e6/len(best):.2f} ms/step, GPU busy (>=1 kernel) {100*tot_busy/tot_span:.1f} %, idle {(tot_span-tot_busy)/1e6/len(best):.2f} ms/step")
tot = sum(conc.values())
print("time share by number of concurrently running kernels:", {k: f"{100*v/tot:.1f}%" for k, v in sorted(conc.items())})
gaps.sort(reverse=True)
print(f"{len(gaps)} idle gaps, {len(gaps)/len(best):.0f} per step; mean {sum(g[0] for g in gaps)/max(len(gaps),1)/1e3:.2f} us; largest:")
for g in gaps[:12]:
    print(f"  {g[0]/1e3:7.1f} us  after {g[1]}  before {g[2]}")
