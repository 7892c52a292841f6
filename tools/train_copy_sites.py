#!/usr/bin/env python3
"""Which lines of the package make torch copy a tensor during one eager training step (B=2, T=5)?  Tensor.contiguous / clone / copy_ /
reshape / to are wrapped and every call that really moves data is attributed to the innermost package frame."""
import collections, os, sys, traceback, torch
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
count = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "_amd/" in fr.filename and "tools/" not in fr.filename:
            return f"{fr.filename.split('_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def wrap(name, moves):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if ON[0] and self.is_cuda and moves(self, out, a, k):
            count[(name, site())] += 1
        return out
    setattr(torch.Tensor, name, f)


wrap("contiguous", lambda s, o, a, k: o.data_ptr() != s.data_ptr())
wrap("clone", lambda s, o, a, k: True)
wrap("copy_", lambda s, o, a, k: True)
wrap("reshape", lambda s, o, a, k: o.numel() > 0 and o.data_ptr() != s.data_ptr() and o.untyped_storage().data_ptr() != s.untyped_storage().data_ptr())
ON[0] = True
step()
ON[0] = False
torch.cuda.synchronize()
print("copying calls per step:", sum(count.values()))
for (n, s), c in count.most_common(60):
    print(f"{c:4d} {n:11s} {s}")
