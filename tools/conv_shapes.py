#!/usr/bin/env python3
"""Time mumpy_conv2d_nhwc_fwd on every decoder convolution of the B=8 forward."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops
dev = torch.device("cuda:0")
B = 8
shapes = [  # tag, Cin, Cout, kh, kw, H
    ("freq0", 32, 128, 3, 3, 112), ("freq1", 128, 128, 3, 3, 56), ("freq2", 128, 128, 3, 3, 28), ("freq3", 128, 32, 3, 3, 14),
    ("freq4", 32, 128, 3, 3, 7), ("gcm1 l1", 2816, 128, 7, 1, 7), ("gcm1 l2", 128, 128, 1, 7, 7), ("gcm2 l1", 256, 32, 7, 1, 14),
    ("gcm2 l2", 32, 32, 1, 7, 14), ("gcm3 l1", 256, 128, 7, 1, 28), ("gcm3 l2", 128, 128, 1, 7, 28), ("gcm4 l1", 256, 128, 7, 1, 56),
    ("gcm4 l2", 128, 128, 1, 7, 56), ("seb1", 256, 256, 3, 3, 7), ("seb2", 512, 256, 3, 3, 14), ("seb3", 768, 256, 3, 3, 28),
    ("dec2", 32, 128, 3, 3, 14), ("dec3", 128, 128, 3, 3, 28), ("dec4", 128, 128, 3, 3, 56), ("dec5", 128, 128, 3, 3, 112)]
tot = 0.0
for tag, cin, cout, kh, kw, h in shapes:
    x = torch.randn(B, h, h, cin, device=dev).permute(0, 3, 1, 2)
    w = torch.randn(cout, kh, kw, cin, device=dev) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout, device=dev)
    for _ in range(3):
        ops.conv2d_nhwc(x, w, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_nhwc(x, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    mult = 2 if tag.startswith("gcm") else 1      # l and r branches have mirrored shapes
    fl = 2.0 * B * h * h * cout * kh * kw * cin
    tot += mult * us
    print(f"{tag:9s} M={B*h*h:7d} N={cout:4d} K={kh*kw*cin:6d} x{mult} {us:8.1f} us {fl/us/1e6:6.1f} TF")
print(f"TOTAL {tot/1e3:.2f} ms")
