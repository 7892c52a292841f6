#!/bin/bash
# The training part of tools/final_profiles.sh alone (config-5 lines + rocprofv3 summary of the fp32 B=2 step + launches per step).
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/train_ddp_bench.py --batch 2 --math fp32 > gpurun_out/train_b2_eager.json 2> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math fp32 --graph --steps 10 > gpurun_out/train_b2_graph.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math bf16 --graph --steps 10 > gpurun_out/train_b2_graph_bf16.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math bf16 --graph --steps 10 --train-mode > gpurun_out/train_b2_graph_bf16_trainmode.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 8 --math fp32 > gpurun_out/train_b8_eager.json 2>> gpurun_out/train_final.err
TOP=60 python tools/train_aten_ops.py > gpurun_out/train_aten_ops.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tr -- python3 $R/tools/train_ddp_bench.py --batch 2 --math fp32 --steps 3 --warmup 1 > $R/gpurun_out/prof_tr.log 2>&1
cd $R
f=$(find gpurun_out/prof_tr -name "*kernel_stats.csv" | head -1); python tools/summarize_rocprof.py $f 45 > gpurun_out/summary_train.md; rm -rf gpurun_out/prof_tr
cat gpurun_out/train_b2_graph.json gpurun_out/train_b2_graph_bf16.json
