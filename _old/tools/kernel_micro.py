#!/usr/bin/env python3
"""Launch one kernel shape repeatedly (for rocprofv3 --pmc / timing).
  kernel_micro.py linear M N K [act]      | kernel_micro.py winattn B Hs W C shift | kernel_micro.py sample B Hs2 W C
  | kernel_micro.py winattn_bwd B Hs W C shift | kernel_micro.py ln_bwd rows C
  | kernel_micro.py adamw N | kernel_micro.py maskloss B P | kernel_micro.py ln rows C"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
dev = torch.device("cuda:0")
ops.set_matrix_math(os.environ.get("MUMPY_MATH", "fp32"))      # fp32 | bf16 | bf16x3 for the linear micro
op, a = sys.argv[1], [int(v) for v in sys.argv[2:]]
reps = int(os.environ.get("REPS", "20"))
if op == "linear":
    m, n, k = a[:3]; act = a[3] if len(a) > 3 else 0
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    fn = lambda: ops.linear(x, w, b, act=act)
    work, unit = 2.0 * m * n * k, "TFLOP/s"
elif op == "winattn":
    b, hs, w, c, shift = a
    qkv = torch.randn(b, hs * w, 3 * c, device=dev)
    bias = ops.expand_relpos_bias(torch.randn(169, c // 32, device=dev) * 0.2,
                                  __import__("models.modules.swinTransformer", fromlist=["x"]).relative_position_index(7, 7).to(dev))
    tab = ids = None
    if shift:
        from models.modules.swinTransformer import build_shift_mask
        tab, ids = ops.compact_attn_mask(build_shift_mask(hs, w, 7, shift).to(dev))
    fn = lambda: ops.window_attention(qkv, bias, b, hs, w, c, shift, 32 ** -0.5, tab, ids)
    work, unit = 307328.0 * b * (hs // 7) * (w // 7) * (c // 32), "TFLOP/s"
elif op == "sample":
    b, hs2, w, c = a
    nw = b * (hs2 // 7) * (w // 7)
    x2 = torch.randn(b, hs2 * w, c, device=dev); pos = torch.rand(nw, 3, 49, 2, device=dev) * 2 - 1
    fn = lambda: ops.deform_sample(x2, pos, b, hs2, w, c, nw)
    work, unit = 4.0 * (2 * nw * 49 * c + nw * 3 * 49 * 2), "GB/s"
if op == "winattn_bwd":
    b, hs, w, c, shift = a
    from models.modules.swinTransformer import build_shift_mask, relative_position_index
    qkv = torch.randn(b, hs * w, 3 * c, device=dev); dout = torch.randn(b, hs * w, c, device=dev)
    idx = relative_position_index(7, 7).to(dev)
    bias = ops.expand_relpos_bias(torch.randn(169, c // 32, device=dev) * 0.2, idx)
    idx32 = idx.to(torch.int32).reshape(-1).contiguous()
    tab = ids = None
    if shift:
        tab, ids = ops.compact_attn_mask(build_shift_mask(hs, w, 7, shift).to(dev))
    fn = lambda: ops.window_attention_bwd(qkv, dout, bias, idx32, b, hs, w, c, shift, 32 ** -0.5, tab, ids)
    work, unit = 5 * 153664.0 * b * (hs // 7) * (w // 7) * (c // 32), "TFLOP/s"      # 5 products of 2*49*49*32 FLOP per unit
if op == "ln_bwd":
    rows, c = a
    x = torch.randn(rows, c, device=dev); g = torch.ones(c, device=dev); dy = torch.randn(rows, c, device=dev)
    fn = lambda: ops.layernorm_bwd(x, g, dy)
    work, unit = 12.0 * rows * c, "GB/s"
if op == "adamw":
    n, = a
    bufs = [torch.randn(n, device=dev) for _ in range(2)] + [torch.zeros(n, device=dev) for _ in range(2)]
    fn = lambda: ops.adamw_step(bufs[0], bufs[1], bufs[2], bufs[3], 3, lr=1e-3)
    work, unit = 28.0 * n, "GB/s"
if op == "maskloss":
    b, p_ = a
    z = torch.randn(b, p_, device=dev); t = (torch.rand(b, p_, device=dev) < 0.1).float()
    fn = lambda: ops.mask_loss(z, t)
    work, unit = 20.0 * b * p_, "GB/s"
if op == "ln":
    rows, c = a
    x = torch.randn(rows, c, device=dev); g = torch.ones(c, device=dev); b = torch.zeros(c, device=dev)
    fn = lambda: ops.layernorm(x, g, b)
    work, unit = 8.0 * rows * c, "GB/s"
for _ in range(3):
    fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print(f"{op} {a}: {us:.1f} us/launch, {work / us / (1e6 if unit == 'TFLOP/s' else 1e3):.1f} {unit}")
