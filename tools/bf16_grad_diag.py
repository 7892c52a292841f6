#!/usr/bin/env python3
"""Gradient quality of the bf16 operand mode at config 5's micro-batch (B=2, T=5): per optimizer group and per parameter, the bf16
gradient (one-call backward, and the first version's transposed-copy route for comparison) against the fp32 gradient of the same
HIP model.  python tools/bf16_grad_diag.py [--legacy]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
from mumpy_hip import ops, autograd as AG
from mumpy_hip.autograd import decoder_train, encoder_train
from mumpy_hip.train import split_param_groups

B, T = 2, 5
x = seeded_randn(990, B, T, 3, 224, 224).cuda()
g = seeded_randn(991, B, 1, 224, 224).cuda()


def run(mode, legacy=False):
    ops.set_matrix_math(mode)
    AG.LEGACY_LINEAR_BWD = legacy
    enc = fill_module_(Encoder(num_frames=T)).eval().cuda()
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, T])).eval().cuda()
    fx, vx, dx = encoder_train(enc, x)
    lg, _ = decoder_train(dec, fx, vx, dx)
    if "--loss" in sys.argv:          # config 5's training loss (softIoU + focal, train.py:107-113) on synthetic Bernoulli(0.1) masks
        target = (torch.rand(B, 1, 224, 224, generator=torch.Generator().manual_seed(7)) < 0.1).float().cuda()
        loss3, dl = ops.mask_loss(lg.detach(), target)
        lg.backward(dl)
    else:
        (lg * g).sum().backward()
    ops.set_matrix_math("fp32")
    AG.LEGACY_LINEAR_BWD = False
    grads = {("enc", n): p.grad.detach().double().cpu() for n, p in enc.named_parameters()}
    grads.update({("dec", n): p.grad.detach().double().cpu() for n, p in dec.named_parameters()})
    groups = {k: [n for n, p in (list(enc.named_parameters()) if k != "dec" else list(dec.named_parameters()))
                  if (k == "dec") or (("cva" in n) == (k == "cva"))] for k in ("enc", "cva", "dec")}
    return lg.detach().double().cpu(), grads, groups


ref_l, ref, groups = run("fp32")
for label, legacy in (("bf16 one-call", False),) + ((("bf16 legacy", True),) if "--legacy" in sys.argv else ()):
    lg, gr, _ = run("bf16", legacy)
    print(f"== {label}: logits rel {float((lg - ref_l).abs().max() / ref_l.abs().max()):.3e}")
    for gname, names in groups.items():
        which = "dec" if gname == "dec" else "enc"
        a = torch.cat([gr[(which, n)].reshape(-1) for n in names]); b = torch.cat([ref[(which, n)].reshape(-1) for n in names])
        print(f"  group {gname}: rel L2 {float((a - b).norm() / b.norm()):.4f}  cos {float(torch.dot(a, b) / (a.norm() * b.norm())):.5f}  |ref| {float(b.norm()):.3e}")
        per = sorted(((float((gr[(which, n)] - ref[(which, n)]).norm()), float(ref[(which, n)].norm()), n) for n in names), reverse=True)[:8]
        for e, r, n in per:
            print(f"      {n:75s} err {e:.3e}  |ref| {r:.3e}  rel {e / max(r, 1e-30):.4f}")
