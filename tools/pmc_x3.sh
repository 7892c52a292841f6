#!/bin/bash
# PMC passes over the split-precision GEMM (one small counter group per pass): bash tools/pmc_x3.sh M N K
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
export TMPDIR=/tmp MUMPY_MATH=${MUMPY_MATH:-bf16x3} REPS=5
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_x3_$i -- python3 $R/tools/kernel_micro.py linear "$@" > $R/gpurun_out/pmc_x3_$i.log 2>&1 || echo "group $i failed"
done
cd $R && python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/pmc_x3_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear" in r["Kernel_Name"] and "splitk" not in r["Kernel_Name"]:
            k = r["Counter_Name"]; acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:28s} {v / n:16.1f}  (mean of {n})")
PY
