#!/usr/bin/env python3
"""Per-launch cost of the LayerNorm folding on the view-3 shapes: producer GEMM with / without the statistics epilogue, consumer GEMM
with / without the folded normalisation, and the LayerNorm launch it replaces."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
dev = torch.device("cuda:0")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for m, c in ((7840, 512), (125440, 128), (31360, 256), (1960, 1024)):
    h4 = torch.randn(m, 4 * c, device=dev); a = torch.randn(m, c, device=dev); r = torch.randn(m, c, device=dev)
    wproj, bproj = torch.randn(c, c, device=dev) / c ** 0.5, torch.randn(c, device=dev)
    wfc2, bfc2 = torch.randn(c, 4 * c, device=dev) / (4 * c) ** 0.5, torch.randn(c, device=dev)
    wqkv, bqkv = torch.randn(3 * c, c, device=dev) / c ** 0.5, torch.randn(3 * c, device=dev)
    wfc1, bfc1 = torch.randn(4 * c, c, device=dev) / c ** 0.5, torch.randn(4 * c, device=dev)
    gam, bet = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    x = ops.linear(a, wproj, bproj, residual=r, emit_stats=True)
    st = ops.ln_stats_of(x)
    print(f"M={m} C={c}  (tiles: proj {ops.linear_ln_tiles(m, c, c)}, fc2 {ops.linear_ln_tiles(m, c, 4 * c)}, qkv {ops.linear_ln_tiles(m, 3 * c, c)}, fc1 {ops.linear_ln_tiles(m, 4 * c, c)})")
    print(f"  layernorm                      {timed(lambda: ops.layernorm(x, gam, bet)):8.1f} us")
    for name, fn0, fn1 in (
        ("proj + residual   plain | emit", lambda: ops.linear(a, wproj, bproj, residual=r), lambda: ops.linear(a, wproj, bproj, residual=r, emit_stats=True)),
        ("fc2 + residual    plain | emit", lambda: ops.linear(h4, wfc2, bfc2, residual=r), lambda: ops.linear(h4, wfc2, bfc2, residual=r, emit_stats=True)),
    ):
        print(f"  {name}  {timed(fn0):8.1f} | {timed(fn1):8.1f} us")
    if st is not None:
        fq = ops.fold_ln_weights(wqkv, bqkv, gam, bet); f1 = ops.fold_ln_weights(wfc1, bfc1, gam, bet)
        xn = ops.layernorm(x, gam, bet)
        print(f"  qkv               plain | fold  {timed(lambda: ops.linear(xn, wqkv, bqkv)):8.1f} | {timed(lambda: ops.linear_ln(x, st, *fq, 1e-5)):8.1f} us")
        print(f"  fc1 + GELU        plain | fold  {timed(lambda: ops.linear(xn, wfc1, bfc1, act=ops.ACT_GELU)):8.1f} | {timed(lambda: ops.linear_ln(x, st, *f1, 1e-5, act=ops.ACT_GELU)):8.1f} us")
