#!/bin/bash
# Round-end measurement pass (on the GPU box): smoke, default bench line, T=9 / B=16 lines, rocprofv3 kernel stats (serial and fork/join
# schedules, fp32 and bf16 storage), timeline gaps of the replayed graph, PMC traffic passes, phase timeline, GEMM shape table,
# config-5 training lines.  Outputs under gpurun_out/ (tag them into profiles/ with tools/collect_profiles.py).
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-16}   # a box reports every host core; torch would start one thread per core
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 || echo "SMOKE FAILED"
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || echo "BENCH FAILED"
echo "bench done"
python bench.py --frames 9 --no-cpu-baseline --no-alt > gpurun_out/bench_t9.json 2> gpurun_out/bench_t9.err || echo "BENCH T9 FAILED"
python bench.py --batch 16 --no-cpu-baseline --no-alt > gpurun_out/bench_b16.json 2> gpurun_out/bench_b16.err || echo "BENCH B16 FAILED"
python bench.py --batch 1 --no-cpu-baseline --no-alt > gpurun_out/bench_b1.json 2> gpurun_out/bench_b1.err || echo "BENCH B1 FAILED"
echo "bench variants done"
cd /tmp
MUMPY_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fj -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_fj.log 2>&1
MUMPY_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s16 -- python3 $R/bench.py --storage bf16 --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $R/gpurun_out/prof_s16.log 2>&1
echo "rocprof done"
cd $R
python tools/timeline_gaps.py gpurun_out/prof_fj 5 > gpurun_out/timeline_gaps.txt 2>&1
for t in serial fj s16; do
  f=$(find gpurun_out/prof_$t -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/stats_$t.csv
  python tools/summarize_rocprof.py $f 60 > gpurun_out/summary_$t.md
  tail -1 gpurun_out/prof_$t.log > gpurun_out/benchline_$t.json
  rm -rf gpurun_out/prof_$t
done
rm -rf gpurun_out/pmc_bench_FETCH_SIZE gpurun_out/pmc_bench_WRITE_SIZE
bash tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1
echo "pmc done"
python tools/phase_timeline.py --graph > gpurun_out/phase_timeline.txt 2>&1
python tools/gemm_shapes.py > gpurun_out/gemm_shapes.txt 2>&1
python tools/ln_fold_shapes.py > gpurun_out/ln_fold_shapes.txt 2>&1
echo "tables done"
python tools/train_ddp_bench.py --batch 2 --math fp32 > gpurun_out/train_b2_eager.json 2> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math fp32 --graph --steps 10 > gpurun_out/train_b2_graph.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math bf16 --graph --steps 10 > gpurun_out/train_b2_graph_bf16.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 2 --math bf16 --graph --steps 10 --train-mode > gpurun_out/train_b2_graph_bf16_trainmode.json 2>> gpurun_out/train_final.err
python tools/train_ddp_bench.py --batch 8 --math fp32 > gpurun_out/train_b8_eager.json 2>> gpurun_out/train_final.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tr -- python3 $R/tools/train_ddp_bench.py --batch 2 --math fp32 --steps 3 --warmup 1 > $R/gpurun_out/prof_tr.log 2>&1
cd $R
f=$(find gpurun_out/prof_tr -name "*kernel_stats.csv" | head -1); python tools/summarize_rocprof.py $f 45 > gpurun_out/summary_train.md; rm -rf gpurun_out/prof_tr
tail -2 gpurun_out/smoke.log
