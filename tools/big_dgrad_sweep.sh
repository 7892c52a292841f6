#!/bin/bash
# Threshold of the large-M dX route of LinearFn.backward (forward GEMM on a transposed copy of W) against the one-call route.
export OMP_NUM_THREADS=16
run() { python tools/train_ddp_bench.py --batch 2 --math $2 --graph --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'])"; }
for m in fp32 bf16; do
  for t in 4096 2048 8192 16384 1000000000; do MUMPY_BIG_DGRAD_ROWS=$t run rows=$t $m; done
done
