#!/usr/bin/env python3
"""Which host lines launch ATen kernels during one eager forward (B=8, T=5)?  torch.profiler with stacks, grouped by the first
frame inside the package."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"), PKG]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
from mumpy_hip.pipeline import fused_forward
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
enc = fill_module_(Encoder(num_frames=5).eval()).to(dev)
dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, 5]).eval()).to(dev)
x = seeded_randn(1, 8, 5, 3, 224, 224).to(dev)
with torch.no_grad():
    for _ in range(2):
        fused_forward(enc, dec, x, with_mask=True)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        fused_forward(enc, dec, x, with_mask=True)
        torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    site = next((s for s in ev.stack if "_amd/" in s and "mumpy_hip/ops.py" not in s), None) or next((s for s in ev.stack if "_amd/" in s), "?")
    site = site.split("_amd/")[-1]
    rows[(ev.name, ev.kernels[0].name[:60], site)] += 1
for (name, kern, site), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {kern:60s} {site}")
print("total ATen-launched kernels per forward:", sum(rows.values()))
