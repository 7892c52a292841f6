#!/bin/bash
# Round-end measurement pass (on the GPU box): default bench line, rocprofv3 kernel stats (serial, fork/join, bf16x3), smoke.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || echo "SMOKE FAILED"
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || echo "BENCH FAILED"
cd /tmp
MUMPY_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fj -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_fj.log 2>&1
MUMPY_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x3 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --math bf16x3 > $R/gpurun_out/prof_x3.log 2>&1
cd $R
for t in serial fj x3; do
  f=$(find gpurun_out/prof_$t -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/stats_$t.csv
  python tools/summarize_rocprof.py $f 45 > gpurun_out/summary_$t.md
  tail -1 gpurun_out/prof_$t.log > gpurun_out/benchline_$t.json
  rm -rf gpurun_out/prof_$t
done
tail -2 gpurun_out/smoke.log
