#!/bin/bash
# usage (on the GPU box): bash tools/pmc_bench.sh   -> gpurun_out/pmc_bench_{FETCH_SIZE,WRITE_SIZE}/ (one counter per pass)
# HBM-side traffic of every kernel of the bench workload (eager, serial schedule, 2 timed steps).
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  MUMPY_SERIAL=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_bench_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-graph > $R/gpurun_out/pmc_bench_$c.log 2>&1 || exit 1
done
find $R/gpurun_out/pmc_bench_* -name "*.csv" | head
