#!/usr/bin/env python3
"""Time mumpy_linear_bwd (dX + dW + db in one call) on every Linear shape of the training step at micro-batch B (default 2,
config 5's per-GPU batch) and print microseconds and TFLOP/s next to the first version's route (transposed copies + the
forward GEMM).  Run on the GPU box: python tools/linear_bwd_shapes.py [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
from gemm_shapes import model_shapes


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    tot_new = tot_old = tot_f = 0.0
    print(f"{'shape':18s} {'M':>7s} {'N':>5s} {'K':>6s} {'cnt':>3s} {'one call us':>11s} {'TF':>6s} {'legacy us':>10s}")
    for (m, n, k), (cnt, tag) in sorted(model_shapes(B=B).items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
        if tag.startswith("dec "):
            continue
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; dy = torch.randn(m, n, device=dev)
        gw, gb = torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)

        def new():
            big = m >= 4096
            ops.linear_bwd(x, w, dy, need_dx=not big, need_dw=True, need_db=True, dw_out=gw, db_out=gb)
            if big:
                ops.linear(dy, ops.transpose(w))

        def old():
            ops.linear(dy, ops.transpose(w))
            gw.add_(ops.linear(ops.transpose(dy, 32), ops.transpose(x, 32)))
            gb.add_(ops.col_sum(dy))
        un, uo = timed(new), timed(old)
        fl = 4.0 * m * n * k
        print(f"{tag:18s} {m:7d} {n:5d} {k:6d} {cnt:3d} {un:11.1f} {fl / un / 1e6:6.1f} {uo:10.1f}")
        tot_new += cnt * un / 1e3; tot_old += cnt * uo / 1e3; tot_f += cnt * fl
    print(f"TOTAL one call {tot_new:.2f} ms ({tot_f / tot_new / 1e9:.1f} TF), legacy {tot_old:.2f} ms")


if __name__ == "__main__":
    main()
