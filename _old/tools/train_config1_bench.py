#!/usr/bin/env python3
"""One training step of config 1 (BaselineEncoder + BaselineDecoder, T=3, 224x224) on the HIP kernels: forward with the
tape, mask loss, backward, fused AdamW.  usage: train_config1_bench.py [batch]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.decoder.decoder import BaselineDecoder
from models.encoder.encoder import BaselineEncoder
from mumpy_hip import ops
from mumpy_hip.autograd import baseline_decoder_train, baseline_encoder_train
from mumpy_hip.train import build_optimizers
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
enc = fill_module_(BaselineEncoder()).eval().to(dev)
dec = fill_module_(BaselineDecoder(in_channels=1024)).eval().to(dev)
opts = build_optimizers(enc, dec, lr_cnn=1e-6, lr=1e-5, weight_decay=1e-4, weight_decay_cnn=1e-4)
x = seeded_randn(1, B, 3, 3, 224, 224).to(dev)
target = (torch.rand(B, 1, 224, 224, device=dev) < 0.1).float()
def step():
    logits = baseline_decoder_train(dec, baseline_encoder_train(enc, x))
    loss3, dlogits = ops.mask_loss(logits.detach(), target)
    logits.backward(dlogits)
    for o in opts.values():
        o.step(); o.zero_grad()
    return loss3
for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
with torch.no_grad():
    for _ in range(2):
        dec(enc(x))
    e0.record()
    for _ in range(n):
        dec(enc(x))
    e1.record(); torch.cuda.synchronize()
inf = e0.elapsed_time(e1) / n
nparam = sum(p.numel() for p in list(enc.parameters()) + list(dec.parameters()))
print(f"config 1 train step, B={B}: {ms:.2f} ms/step = {B / ms * 1e3:.1f} clips/s  ({nparam / 1e6:.1f} M parameters); "
      f"eager inference forward of the same model {inf:.2f} ms -> step / forward = {ms / inf:.2f}")
