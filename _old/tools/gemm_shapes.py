#!/usr/bin/env python3
"""Time mumpy_linear_fwd on every GEMM shape of the B=8,T=5 forward (with its launch count) and print TFLOP/s."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops


def model_shapes(B=8, T=5):
    shapes = {}
    def add(m, n, k, cnt=1, tag=""):
        key = (m, n, k)
        c, t = shapes.get(key, (0, ""))
        shapes[key] = (c + cnt, t or tag)
    depth = [[2, 2, 6, 2], [2, 2, 18, 2], [2, 2, 18, 2]]
    for v in range(3):
        for s in range(4):
            C = (128 if v == 2 else 96) * 2 ** s
            M = B * (T if v == 2 else 1) * 3136 // 4 ** s
            d = depth[v][s]
            add(M, 3 * C, C, d, f"v{v+1}s{s} qkv"); add(M, C, C, d, f"v{v+1}s{s} proj")
            add(M, 4 * C, C, d, f"v{v+1}s{s} fc1"); add(M, C, 4 * C, d, f"v{v+1}s{s} fc2")
            if s < 3:
                add(M // 4, 2 * C, 4 * C, 1, f"v{v+1}s{s} merge")
            if v < 2:
                C2 = (128 if v == 1 else 96) * 2 ** s
                M2 = B * (T if v == 1 else 1) * 3136 // 4 ** s
                add(M2, C, C2, 1, f"v{v+1}s{s} pre"); add(M, C, C, 2, f"v{v+1}s{s} proj_q/out"); add(M2, 2 * C, C, 1, f"v{v+1}s{s} kv")
    Mg = B * 49 * T
    add(Mg, 768, 2560, 1, "global embed")
    add(Mg, 2304, 768, 12, "g qkv"); add(Mg, 768, 768, 12, "g proj"); add(Mg, 3072, 768, 12, "g fc1"); add(Mg, 768, 3072, 12, "g fc2")
    for s, cin in enumerate([320, 640, 1280, 2560]):
        add(B * 3136 // 4 ** s, 256, cin * T, 1, f"dec rgb{s+1}")
    return shapes


def main():
    dev = torch.device("cuda:0")
    tile = int(os.environ.get("MUMPY_GEMM_TILE", "0"))
    ops.set_matrix_math(os.environ.get("MUMPY_MATH", "fp32"))       # fp32 | bf16 | bf16x3
    check = os.environ.get("MUMPY_GEMM_CHECK", "0") != "0"           # max / rms error against an fp64 product
    tot_t = tot_f = 0.0
    rows = []
    for (m, n, k), (cnt, tag) in sorted(model_shapes().items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * kv[1][0]):
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
        for _ in range(3):
            ops.linear(x, w, b)
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 2.0 * m * n * k
        err = ""
        if os.environ.get("MUMPY_COMPARE_TORCH", "0") != "0":            # reference point only: torch's library GEMM (rocBLAS/hipBLASLt)
            wt = w.t().contiguous()
            for _ in range(3):
                torch.addmm(b, x, wt)
            e0.record()
            for _ in range(reps):
                torch.addmm(b, x, wt)
            e1.record(); torch.cuda.synchronize()
            ust = e0.elapsed_time(e1) * 1e3 / reps
            err = f"  | torch.addmm {ust:8.1f} us {fl / ust / 1e6:6.1f} TF  ratio {us / ust:5.2f}"
        if check:
            mm = min(m, 2048)
            ref = x[:mm].double() @ w.double().t() + b.double()
            d = (ops.linear(x, w, b)[:mm].double() - ref)
            err = f"  max {d.abs().max().item():.2e} rms {d.pow(2).mean().sqrt().item():.2e} (ref rms {ref.pow(2).mean().sqrt().item():.2f})"
        rows.append((tag, m, n, k, cnt, us, fl / us / 1e6, cnt * us / 1e3, err))
        tot_t += cnt * us / 1e3; tot_f += cnt * fl
    print(f"{'shape':18s} {'M':>7s} {'N':>5s} {'K':>6s} {'cnt':>3s} {'us':>8s} {'TF':>6s} {'ms tot':>7s}")
    for r in rows:
        print(f"{r[0]:18s} {r[1]:7d} {r[2]:5d} {r[3]:6d} {r[4]:3d} {r[5]:8.1f} {r[6]:6.1f} {r[7]:7.2f}{r[8]}")
    print(f"TOTAL {tot_t:.2f} ms, {tot_f/1e9:.1f} GFLOP, {tot_f/tot_t/1e9:.1f} TF average")


if __name__ == "__main__":
    main()
