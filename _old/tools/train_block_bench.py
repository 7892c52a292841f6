#!/usr/bin/env python3
"""Forward + backward + AdamW of ONE Swin block on the HIP kernels at a stage shape of the B=8,T=5 workload.
usage: train_block_bench.py [stage]   (stage 0: 40 frames 56x56 C=128; stage 2: 40 frames 14x14 C=512)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_
from models.modules.swinTransformer import SwinTransformerBlock
from mumpy_hip.autograd import swin_block_train
from mumpy_hip.train import FlatAdamW
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
side, c = {0: (56, 128), 1: (28, 256), 2: (14, 512), 3: (7, 1024)}[stage]
dev = torch.device("cuda:0")
blk = fill_module_(SwinTransformerBlock(dim=c, input_resolution=(side, side), num_heads=c // 32, window_size=7,
                                        shift_size=3 if side > 7 else 0, temporal_dim=5)).to(dev)
opt = FlatAdamW(blk.parameters(), lr=1e-4)
x = torch.randn(8, 5 * side * side, c, device=dev, requires_grad=True)
g = torch.randn_like(x)
def step():
    y = swin_block_train(blk, x)
    y.backward(g)
    opt.step()
    opt.zero_grad()
for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
tok = 8 * 5 * side * side
flops_fwd = 2.0 * tok * (12 * c * c) + 2 * 307328.0 * 40 * (side // 7) ** 2 * (c // 32) / 2
print(f"stage {stage}: {tok} tokens x C={c}: fwd+bwd+AdamW {ms:.3f} ms/step; forward GEMM+attention FLOPs {flops_fwd / 1e9:.1f} G -> "
      f"~{3 * flops_fwd / ms / 1e9:.1f} TFLOP/s at 3x forward FLOPs")
with torch.no_grad():
    e0.record()
    for _ in range(n):
        blk(x)
    e1.record(); torch.cuda.synchronize()
print(f"          inference forward of the same block: {e0.elapsed_time(e1) / n:.3f} ms")
