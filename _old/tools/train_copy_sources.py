#!/usr/bin/env python3
"""Where do the ATen copy / add / fill launches of one eager training step come from?  torch.profiler with stacks, grouped by
operator and input shapes.  Run on the GPU box: python tools/train_copy_sources.py [B]"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from torch.profiler import ProfilerActivity, profile


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    from mumpy_hip import ops
    from models.decoder.decoder import Decoder
    from models.encoder.encoder import Encoder
    from mumpy_hip.autograd import decoder_train, encoder_train
    from mumpy_hip.train import build_optimizers
    from weight_fill import fill_module_, seeded_randn
    enc = fill_module_(Encoder(num_frames=5)).eval().to(dev)
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, 5])).eval().to(dev)
    opts = build_optimizers(enc, dec, lr_cnn=1e-6, lr=1e-5, lr_cva=1e-6, weight_decay=1e-4, weight_decay_cnn=1e-4)
    x = seeded_randn(100, B, 5, 3, 224, 224).to(dev)
    target = (torch.rand(B, 1, 224, 224) < 0.1).float().to(dev)

    def step():
        fx, vx, dx = encoder_train(enc, x)
        logits, _ = decoder_train(dec, fx, vx, dx)
        loss3, dlogits = ops.mask_loss(logits.detach(), target)
        logits.backward(dlogits)
        for o in opts.values():
            o.step(); o.zero_grad()
    step(); step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        step()
    torch.cuda.synchronize()
    want = ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::contiguous", "aten::clone", "aten::cat",
            "aten::_to_copy", "aten::index", "aten::flip", "aten::repeat", "aten::sum", "aten::mul")
    by = collections.Counter()
    for ev in prof.events():
        if ev.name not in want:
            continue
        where = str([tuple(sh) if isinstance(sh, (list, tuple)) else sh for sh in (ev.input_shapes or [])][:2])
        by[(ev.name, where)] += 1
    for (name, where), n in by.most_common(60):
        print(f"{n:5d}  {name:18s} {where}")


if __name__ == "__main__":
    main()
