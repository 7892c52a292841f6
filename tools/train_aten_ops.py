#!/usr/bin/env python3
"""Which ATen ops (and with what shapes) one eager training step of the three-view model launches (B=2, T=5): torch.profiler,
grouped by (op, input shapes).  python tools/train_aten_ops.py [--math bf16]"""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
from mumpy_hip import ops
from mumpy_hip.autograd import decoder_train, encoder_train
from mumpy_hip.train import build_optimizers
from torch.profiler import profile, ProfilerActivity
if "--math" in sys.argv:
    ops.set_matrix_math(sys.argv[sys.argv.index("--math") + 1])
dev = torch.device("cuda:0")
enc = fill_module_(Encoder(num_frames=5)).eval().to(dev)
dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, 5])).eval().to(dev)
opts = build_optimizers(enc, dec, lr_cnn=1e-6, lr=1e-5, lr_cva=1e-6, weight_decay=1e-4, weight_decay_cnn=1e-4)
x = seeded_randn(100, 2, 5, 3, 224, 224).to(dev)
target = (torch.rand(2, 1, 224, 224, generator=torch.Generator().manual_seed(7)) < 0.1).float().to(dev)


def step():
    fx, vx, dx = encoder_train(enc, x)
    logits, _ = decoder_train(dec, fx, vx, dx)
    loss3, dl = ops.mask_loss(logits.detach(), target)
    logits.backward(dl)
    for o in opts.values():
        o.step(); o.zero_grad()


for _ in range(2):
    step()
torch.cuda.synchronize()
ops.PROFILE = {}
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
ours, ops.PROFILE = ops.PROFILE, None
rows = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.kernels:
        rows[(ev.name, str(ev.input_shapes)[:110])] += len(ev.kernels)
tot = sum(rows.values())
print(f"ATen-launched kernels per step: {tot};  C-ABI calls per step: {sum(len(v) for v in ours.values())}")
by_op = collections.Counter()
for (n, sh), c in rows.items():
    by_op[n] += c
print("by op:", dict(by_op.most_common()))
for (n, sh), c in rows.most_common(int(os.environ.get("TOP", "45"))):
    print(f"{c:5d} {n:22s} {sh}")
print("C-ABI calls:", {k: len(v) for k, v in sorted(ours.items(), key=lambda kv: -len(kv[1]))})
