#!/bin/bash
# The tiled kernels' split-K rule (slices >= 384 deep, at most 16 splits, K >= 768) against more aggressive ones, on the whole forward.
export OMP_NUM_THREADS=16 MUMPY_TUNING=1
run() { python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={e['kernel']:e for e in d['kernels']}
print('$1', d['value'], d['ms_per_step'], 'conv ms', k['mumpy_conv2d_nhwc_fwd']['ms'], 'linear ms', k['mumpy_linear_wsz_fwd']['ms'])"; }
for i in 1 2; do
run base
MUMPY_GEMM_MINSLICE=128 MUMPY_GEMM_MAXKS=64 run s128k64
MUMPY_GEMM_MINSLICE=192 MUMPY_GEMM_MAXKS=32 run s192k32
MUMPY_GEMM_MINSLICE=256 MUMPY_GEMM_MAXKS=64 run s256k64
MUMPY_GEMM_MINSLICE=128 MUMPY_GEMM_MAXKS=32 MUMPY_GEMM_SPLIT_MINK=512 run s128k32m512
done
