#!/bin/bash
# Round-end measurement pass (on the GPU box): smoke, default bench line, T=9 line, rocprofv3 kernel stats (serial and fork/join
# schedules), timeline gaps of the replayed graph, PMC traffic passes, config-5 training lines.  Outputs under gpurun_out/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || echo "SMOKE FAILED"
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || echo "BENCH FAILED"
python bench.py --frames 9 --no-cpu-baseline --no-alt > gpurun_out/bench_t9.json 2> gpurun_out/bench_t9.err || echo "BENCH T9 FAILED"
cd /tmp
MUMPY_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fj -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_fj.log 2>&1
cd $R
python tools/timeline_gaps.py gpurun_out/prof_fj 5 > gpurun_out/timeline_gaps.txt 2>&1
for t in serial fj; do
  f=$(find gpurun_out/prof_$t -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/stats_$t.csv
  python tools/summarize_rocprof.py $f 45 > gpurun_out/summary_$t.md
  tail -1 gpurun_out/prof_$t.log > gpurun_out/benchline_$t.json
  rm -rf gpurun_out/prof_$t
done
rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
python tools/train_ddp_bench.py --batch 2 --math fp32 > gpurun_out/train_b2_eager.json 2> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math fp32 --graph > gpurun_out/train_b2_graph.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math bf16 --graph > gpurun_out/train_b2_graph_bf16.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 8 --math fp32 > gpurun_out/train_b8_eager.json 2>> gpurun_out/train_final.err
tail -2 gpurun_out/smoke.log
