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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ours, ops.PROFILE = ops.PROFILE, None
rows = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.kernels:
        rows[(ev.name, str(ev.input_shapes)[:110])] += len(ev.kernels)
# device-side memcpy / memset activities (hipMemcpyAsync D2D is what a contiguous same-dtype copy_ becomes: no kernel name): every
# device event of the step by name, and the CPU ops that own a "Memcpy" one with the innermost package frame of their stack
dev_names = collections.Counter()
mem_sites = collections.Counter()
for ev in prof.events():
    if ev.kernels and not any(c.kernels for c in ev.cpu_children):          # innermost CPU op owning device work
        for k in ev.kernels:
            dev_names[k.name.split("(")[0][:70]] += 1
            if "emcpy" in k.name or "emset" in k.name or not k.name.strip():
                fr = next((f for f in (ev.stack or []) if "_amd/" in f and "tools/" not in f), (ev.stack or ["?"])[0] if ev.stack else "?")
                mem_sites[(ev.name, str(ev.input_shapes)[:70], fr.split("_amd/")[-1][:90])] += 1
from torch.autograd import DeviceType
n_dev = sum(1 for ev in prof.events() if ev.device_type == DeviceType.CUDA)
print(f"device activities (kernels + memcpy / memset) in ONE step: {n_dev}")
print("device activities of the step by name (top 25):")
for n, c in dev_names.most_common(25):
    print(f"{c:6d} {n}")
print("memcpy / memset / unnamed-activity owners:")
for (n, sh, fr), c in mem_sites.most_common(40):
    print(f"{c:5d} {n:20s} {sh:70s} {fr}")
tot = sum(rows.values())
print(f"ATen-launched kernels per step: {tot};  C-ABI calls per step: {sum(len(v) for v in ours.values())}")
by_op = collections.Counter()
for (n, sh), c in rows.items():
    by_op[n] += c
print("by op:", dict(by_op.most_common()))
for (n, sh), c in rows.most_common(int(os.environ.get("TOP", "45"))):
    print(f"{c:5d} {n:22s} {sh}")
print("C-ABI calls:", {k: len(v) for k, v in sorted(ours.items(), key=lambda kv: -len(kv[1]))})
