#!/usr/bin/env python3
"""The slowest ops.linear calls of one eager training step (B=2, T=5) with their shapes and the package line that issued them."""
import os, sys, traceback, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
from mumpy_hip import ops
from mumpy_hip.autograd import decoder_train, encoder_train
from mumpy_hip.train import build_optimizers
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


step(); step()
torch.cuda.synchronize()
log = []
real = ops.linear


def spy(xx, weight, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fr = next((f for f in reversed(traceback.extract_stack()[:-1]) if "_amd/" in f.filename and "ops.py" not in f.filename), None)
    e0.record()
    y = real(xx, weight, *a, **k)
    e1.record()
    log.append((e0, e1, tuple(xx.shape), tuple(weight.shape), f"{fr.filename.split('_amd/')[-1]}:{fr.lineno}" if fr else "?"))
    return y


ops.linear = spy
step()
torch.cuda.synchronize()
rows = sorted(((e0.elapsed_time(e1) * 1e3, xs, ws, where) for e0, e1, xs, ws, where in log), reverse=True)
print(f"{len(rows)} ops.linear calls in the step; slowest 25 (us incl. host launch time of the call):")
for t, xs, ws, where in rows[:25]:
    print(f"{t:8.1f}  x {xs}  W {ws}  {where}")
