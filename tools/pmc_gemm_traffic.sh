#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the persistent GEMM per SHAPE (one counter per pass), to see which shapes over-fetch.
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
shapes=("fc1 7840 2048 512 1" "fc2 7840 512 2048 0" "qkv 7840 1536 512 0" "proj 7840 512 512 0" "gfc1 1960 3072 768 1" "gfc2 1960 768 3072 0" "s0fc1 125440 512 128 1" "s3fc2 1960 1024 4096 0")
for s in "${shapes[@]}"; do
  set -- $s
  for c in FETCH_SIZE WRITE_SIZE; do
    bash tools/pmc.sh pmc3_$1_$c $c -- linear $2 $3 $4 $5 || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
out = open(f"{R}/gpurun_out/pmc3_gemm_traffic.md", "w")
out.write("| shape (M N K) | kernel | read MB (FETCH_SIZE x2) | written MB | algorithmic read / write MB | us |\n|---|---|---:|---:|---:|---:|\n")
for tag, m, n, k in (("fc1", 7840, 2048, 512), ("fc2", 7840, 512, 2048), ("qkv", 7840, 1536, 512), ("proj", 7840, 512, 512), ("gfc1", 1960, 3072, 768),
                     ("gfc2", 1960, 768, 3072), ("s0fc1", 125440, 512, 128), ("s3fc2", 1960, 1024, 4096)):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{R}/gpurun_out/pmc3_{tag}_{c}/*/*counter_collection.csv")[0]
        rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] or "linear_kernel" in r["Kernel_Name"] or "splitk" in r["Kernel_Name"]]
        per = {}
        for r in rows:
            per.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(float(r["Counter_Value"]))
        vals[c] = {kname: sum(v) / len(v) for kname, v in per.items()}
        tf = glob.glob(f"{R}/gpurun_out/pmc3_{tag}_{c}/*/*kernel_trace.csv")[0]
        d = {}
        for r in csv.DictReader(open(tf)):
            d.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for kname in vals["FETCH_SIZE"]:
        rd = vals["FETCH_SIZE"][kname] * 2 * 1024 / 1e6            # KiB, doubled (gfx950: 128-B requests tallied at 64 B)
        wr = vals["WRITE_SIZE"].get(kname, 0) * 1024 / 1e6
        out.write(f"| {tag} {m} {n} {k} | `{kname}` | {rd:.1f} | {wr:.1f} | {4e-6 * (m * k + n * k):.1f} / {4e-6 * m * n:.1f} | {sum(d[kname]) / len(d[kname]) / 1e3:.1f} |\n")
out.close()
print(open(f"{R}/gpurun_out/pmc3_gemm_traffic.md").read())
PY
