#!/usr/bin/env python3
"""One training step of the full three-view model (Encoder + Decoder) on the HIP kernels: taped forward, mask loss, backward,
fused AdamW over the cva / encoder / decoder groups (train.py:94-138).  usage: train_full_bench.py [batch] [frames] [--bf16 | --x3 | --x2] [--graph]
--x3: split-precision GEMMs / convolutions (fp32 products from three bf16 pieces per operand, fp32-level accuracy).
--bf16: bf16-operand GEMMs / convolutions (fp32 accumulate, fp32 master weights, fp32 everything else) in forward and backward,
i.e. config 5's matrix arithmetic; prints the gradient deviation from the fp32 step as well."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.decoder.decoder import Decoder
from models.encoder.encoder import Encoder
from mumpy_hip import ops
from mumpy_hip.autograd import decoder_train, encoder_train
from mumpy_hip.train import build_optimizers
args = [a for a in sys.argv[1:] if not a.startswith("--")]
BF16 = "--bf16" in sys.argv
MODE = "bf16x3" if "--x3" in sys.argv else "bf16x2" if "--x2" in sys.argv else "bf16"
B = int(args[0]) if args else 2
T = int(args[1]) if len(args) > 1 else 5
dev = torch.device("cuda:0")
enc = fill_module_(Encoder(num_frames=T)).eval().to(dev)
dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, T])).eval().to(dev)
opts = build_optimizers(enc, dec, lr_cnn=1e-6, lr=1e-5, lr_cva=1e-6, weight_decay=1e-4, weight_decay_cnn=1e-4)
x = seeded_randn(1, B, T, 3, 224, 224).to(dev)
target = (torch.rand(B, 1, 224, 224, device=dev) < 0.1).float()
def step():
    fx, vx, dx = encoder_train(enc, x)
    logits, _ = decoder_train(dec, fx, vx, dx)
    loss3, dlogits = ops.mask_loss(logits.detach(), target)
    logits.backward(dlogits)
    for o in opts.values():
        o.step(); o.zero_grad()
    return loss3
if BF16 or MODE != "bf16":                # one step's flat encoder gradient in fp32 vs bf16 matrix math, same weights
    def grads(mode):
        ops.set_matrix_math(mode)
        for o in opts.values():
            o.zero_grad()
        fx, vx, dx = encoder_train(enc, x)
        logits, _ = decoder_train(dec, fx, vx, dx)
        logits.backward(ops.mask_loss(logits.detach(), target)[1])
        return torch.cat([o.grad.clone() for o in opts.values()])
    g32, g16 = grads("fp32"), grads(MODE)
    print(f"{MODE}-math gradient vs fp32 gradient: relative L2 distance {float((g16 - g32).norm() / g32.norm()):.3e}, "
          f"cosine {float(torch.dot(g16, g32) / (g16.norm() * g32.norm())):.6f}")
    for o in opts.values():
        o.zero_grad()
losses = [float(step()[0]) for _ in range(2)]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 3
e0.record()
for _ in range(n):
    losses.append(float(step()[0]))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
if "--graph" in sys.argv:
    from mumpy_hip.train import GraphedTrainStep
    def fwd(xx):
        fx, vx, dx = encoder_train(enc, xx)
        return decoder_train(dec, fx, vx, dx)[0]
    gs = GraphedTrainStep(fwd, opts, x, target)
    for _ in range(2):
        gs.step()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        lg = gs.step()
    e1.record(); torch.cuda.synchronize()
    print(f"   hipGraph replay of the same step: {e0.elapsed_time(e1) / n:.1f} ms/step = {B / (e0.elapsed_time(e1) / n) * 1e3:.1f} clips/s; loss {float(lg[0]):.4f}")
with torch.no_grad():
    for _ in range(2):
        dec(*enc(x))
    e0.record()
    for _ in range(n):
        dec(*enc(x))
    e1.record(); torch.cuda.synchronize()
inf = e0.elapsed_time(e1) / n
nparam = sum(p.numel() for p in list(enc.parameters()) + list(dec.parameters()))
print(f"full model train step ({ops.matrix_math()} matrix math), B={B}, T={T}: {ms:.1f} ms/step = {B / ms * 1e3:.1f} clips/s ({nparam / 1e6:.1f} M parameters, groups {sorted(opts)}); "
      f"eager inference forward {inf:.1f} ms -> step / forward = {ms / inf:.2f}; loss over 5 steps {[round(l, 4) for l in losses]}")
