#!/bin/bash
# MFMA-busy counter passes (one counter per pass) for the persistent GEMM and window attention on the round's final code
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  bash tools/pmc.sh pmc2_gemm_fc1_$c $c -- linear 7840 2048 512 1 || exit 1
  bash tools/pmc.sh pmc2_gemm_fc2_$c $c -- linear 7840 512 2048 0 || exit 1
  bash tools/pmc.sh pmc2_wa_$c $c -- winattn 8 280 56 128 3 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def mean(tag, ctr, kern):
    f = glob.glob(f"{R}/gpurun_out/pmc2_{tag}_{ctr}/*/*counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
def dur(tag, ctr, kern):
    f = glob.glob(f"{R}/gpurun_out/pmc2_{tag}_{ctr}/*/*kernel_trace.csv")[0]
    v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    return sum(v) / len(v) / 1e3
out = open(f"{R}/gpurun_out/pmc2_mfma.md", "w")
out.write("| kernel, shape | duration under the counters | GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES | MFMA-busy | launches |\n|---|---:|---:|---:|---:|---:|\n")
for tag, kern, what in (("gemm_fc1", "gemm_ws_kernel", "gemm_ws_kernel<2>, M=7840 N=2048 K=512 +GELU"),
                        ("gemm_fc2", "gemm_ws_kernel", "gemm_ws_kernel<1>, M=7840 N=512 K=2048"),
                        ("wa", "win_attn_self_kernel", "win_attn_self_kernel, 10,240 window-heads, shift 3")):
    busy, n = mean(tag, "SQ_VALU_MFMA_BUSY_CYCLES", kern)
    act, _ = mean(tag, "GRBM_GUI_ACTIVE", kern)
    d = dur(tag, "GRBM_GUI_ACTIVE", kern)
    out.write(f"| `{what}` | {d:.1f} us | {act:,.0f} | {busy:,.0f} | **{100 * busy / (1024 * act / 8):.1f} %** | {n} |\n")
out.close()
print(open(f"{R}/gpurun_out/pmc2_mfma.md").read())
PY
