#!/usr/bin/env python3
"""mumpy_linear_bf16s_fwd (bf16 operands in memory; config 3) on the large Linear shapes of the B=8,T=5 forward: microseconds,
TFLOP/s and the error against an fp64 product of the same bf16 operands.  MUMPY_GEMM_WS16=0 python tools/gemm16_shapes.py times the
tiled bf16 kernels instead of the persistent one; MUMPY_GEMM_WS16=2 forces the persistent kernel on every eligible shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
dev = torch.device("cuda:0")
shapes = [(7840, 2048, 512, 1, True, "v3s2 fc1"), (7840, 512, 2048, 0, False, "v3s2 fc2"), (7840, 1536, 512, 0, True, "v3s2 qkv"),
          (7840, 512, 512, 0, False, "v3s2 proj"), (1960, 3072, 768, 1, True, "g fc1"), (1960, 768, 3072, 0, False, "g fc2"),
          (125440, 512, 128, 1, True, "v3s0 fc1"), (125440, 128, 512, 0, False, "v3s0 fc2"), (31360, 1024, 256, 1, True, "v3s1 fc1"),
          (31360, 256, 1024, 0, False, "v3s1 fc2"), (1000, 224, 192, 1, True, "ragged"), (1000, 224, 256, 0, False, "ragged"), (37, 32, 192, 0, True, "tiny")]
tot = 0.0
for m, n, k, act, out16, tag in shapes:
    x = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    b = torch.randn(n, device=dev); res = None if out16 else torch.randn(m, n, device=dev)
    fn = lambda: ops.linear_bf16s(x, w, b, act=act, residual=res, out_bf16=out16)
    y = fn()
    mm = min(m, 1024)
    ref = x[:mm].double() @ w.double().t() + b.double()
    if act:
        ref = torch.nn.functional.gelu(ref)
    if res is not None:
        ref = ref + res[:mm].double()
    err = (y[:mm].double() - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    tot += us
    print(f"{tag:10s} M={m:6d} N={n:5d} K={k:5d} act={act} out16={int(out16)}  {us:8.1f} us {2.0 * m * n * k / us / 1e6:7.1f} TF   rel err {err:.2e}")
print(f"sum {tot:.1f} us")
