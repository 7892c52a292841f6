#!/usr/bin/env python3
"""Ablation timing of one GEMM shape in a given matrix-math mode: python tools/x3abl.py M N K [gelu]  (MUMPY_MATH, MUMPY_GEMM_DBG)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
ops.set_matrix_math(os.environ.get("MUMPY_MATH", "fp32"))
m, n, k = map(int, sys.argv[1:4]); act = ops.ACT_GELU if len(sys.argv) > 4 else ops.ACT_NONE
x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
for _ in range(3): ops.linear(x, w, b, act=act)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.linear(x, w, b, act=act)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
print(f"{os.environ.get('MUMPY_MATH','fp32'):7s} dbg={os.environ.get('MUMPY_GEMM_DBG','0'):3s} M={m} N={n} K={k} act={act}: {us:7.1f} us  {2.0*m*n*k/us/1e6:6.1f} TF")
