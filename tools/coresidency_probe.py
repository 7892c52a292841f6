#!/usr/bin/env python3
"""Can another kernel run on a CU while the persistent GEMM (one 768-thread workgroup per CU holding the whole 160 KB of LDS) is
resident?  Two streams: the big GEMM back to back on one, a chain of small kernels on the other; wall time alone vs together.
  - layernorm (no LDS), window attention (16 KB LDS), the tiled 64x64 GEMM (36.8 KB LDS), the 64x64 persistent GEMM (64 KB)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
from models.modules.swinTransformer import relative_position_index
dev = torch.device("cuda:0")
xb = torch.randn(7840, 512, device=dev); wb = torch.randn(2048, 512, device=dev) / 22.0; bb = torch.randn(2048, device=dev)
big = lambda: ops.linear(xb, wb, bb, act=ops.ACT_GELU)
xs = torch.randn(1568, 384, device=dev); g = torch.ones(384, device=dev); b0 = torch.zeros(384, device=dev)
w1 = torch.randn(1536, 384, device=dev) / 20.0; b1 = torch.randn(1536, device=dev)
wq = torch.randn(1152, 384, device=dev) / 20.0; bq = torch.randn(1152, device=dev)
qkv = torch.randn(8, 196, 1152, device=dev)
bias = ops.expand_relpos_bias(torch.randn(169, 12, device=dev) * 0.2, relative_position_index(7, 7).to(dev))
small = {
    "layernorm (no LDS)": lambda: ops.layernorm(xs, g, b0),
    "window attention (16 KB LDS)": lambda: ops.window_attention(qkv, bias, 8, 14, 14, 384, 0, 32 ** -0.5),
    "tiled 64x64 GEMM +GELU (36.8 KB LDS)": lambda: ops.linear(xs, w1, b1, act=ops.ACT_GELU),
    "persistent 64x64 GEMM (64 KB LDS)": lambda: ops.linear(xs, wq, bq),
    "register-direct GEMM +GELU (no LDS)": lambda: bgd(lambda: ops.linear(xs, w1, b1, act=ops.ACT_GELU)),
    "register-direct GEMM qkv (no LDS)": lambda: bgd(lambda: ops.linear(xs, wq, bq)),
    "window attention, background form (no LDS)": lambda: bgd(lambda: ops.window_attention(qkv, bias, 8, 14, 14, 384, 0, 32 ** -0.5)),
}


def bgd(fn):
    with ops.background():
        return fn()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def wall(fa, na, fb, nb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(na):
                fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(nb):
                fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(200):          # steady clock before anything is measured
    big()
torch.cuda.synchronize()
for name, fs in list(small.items()) * 2:
    for _ in range(3):
        big(); fs()
    nb_ = 40
    tb = wall(big, nb_, None, 0)
    ns = 400
    ts = wall(None, 0, fs, ns)
    ns = max(50, int(ns * tb / ts))               # same wall time alone
    ts = wall(None, 0, fs, ns)
    tt = wall(big, nb_, fs, ns)
    print(f"{name:40s} big alone {tb:6.2f} ms | {ns:4d} small alone {ts:6.2f} ms | together {tt:6.2f} ms  (sum {tb + ts:6.2f}, max {max(tb, ts):6.2f})")
