#!/usr/bin/env python3
"""Golden vectors for the training tail, produced by IMPORTING the reference's own loss and scheduler
classes in the build container (never on the GPU box).  Writes tests/golden/train_tail.npz.

Stand-ins, as in gen_goldens.py: `timm.scheduler.scheduler.Scheduler` (an unused import of scheduler.py:3) and
`torch.Tensor.cuda` = identity (loss.py:12 calls `.cuda()` in a constructor).  Inputs are stored as (seed, shape) only.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weight_fill import seeded_randn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    d = tempfile.mkdtemp(prefix="mumpy_stubs_")
    os.makedirs(os.path.join(d, "timm", "scheduler"))
    open(os.path.join(d, "timm", "__init__.py"), "w").close()
    open(os.path.join(d, "timm", "scheduler", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "scheduler", "scheduler.py"), "w") as f:
        f.write("class Scheduler:\n    pass\n")
    from gen_goldens import STUB_MLC, STUB_TIMM                  # the model-side stand-ins, same text as gen_goldens.py
    os.makedirs(os.path.join(d, "timm", "models"))
    open(os.path.join(d, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "models", "layers.py"), "w") as f:
        f.write(STUB_TIMM)
    with open(os.path.join(d, "ml_collections.py"), "w") as f:
        f.write(STUB_MLC)
    sys.path.insert(0, d)
    sys.path.insert(0, args.ref)
    torch.Tensor.cuda = lambda self, *a, **k: self
    from utils.loss import WeightedFocalLoss, softIoULoss
    from utils.optimizer.scheduler import PolynomialLR

    store = {}
    siou, focal = softIoULoss(), WeightedFocalLoss()
    for tag, seed, b, hw in (("toy", 501, 3, (1, 10, 100)), ("odd", 502, 5, (1, 7, 331)), ("full", 503, 2, (1, 224, 224))):
        # train.py:94-113: y_mask (B,1,P) 0/1, out_mask (B,1,H,W) logits
        z = (seeded_randn(seed, b, *hw) * 2.0).requires_grad_(True)
        t = (seeded_randn(seed + 50, b, 1, hw[1] * hw[2]) > 0.8).float()
        l_iou = torch.mean(siou(t.reshape(-1, t.size()[-1]), z.reshape(z.size()[0], -1)))
        l_foc = torch.mean(focal(t.reshape(-1, t.size()[-1]), z.reshape(z.size()[0], -1)))
        loss = (l_iou + l_foc) / 2.0                     # accumulation_steps = 2
        loss.backward()
        store[f"{tag}/seed_shape"] = np.array([seed, b, *hw], dtype=np.int64)
        store[f"{tag}/loss3"] = np.array([loss.item(), l_iou.item(), l_foc.item()], dtype=np.float64)
        store[f"{tag}/dlogits"] = z.grad.numpy().astype(np.float32)
        print(tag, store[f"{tag}/loss3"])

    # PolynomialLR as train.py:226-262 builds it (power 0.9, min_lr 1e-5, step_size 1, no warm-up), stepped past iter_max
    for tag, base, iter_max in (("sched_a", 1e-3, 40), ("sched_b", 0.9, 7)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=base)
        sch = PolynomialLR(opt, step_size=1, iter_warmup=0.0, iter_max=iter_max, power=0.9, min_lr=1e-5)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(iter_max + 5):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        store[f"{tag}/base_itermax"] = np.array([base, iter_max], dtype=np.float64)
        store[f"{tag}/lrs"] = np.array(lrs, dtype=np.float64)
        print(tag, lrs[:3], lrs[-3:])
    # SwinTransformerBlock forward + backward through the reference's own module and torch autograd (swin:185-307):
    # x (2, 14*14, 96), 3 heads, shift 0 and 3; loss = sum(y * g) with a seeded cotangent g
    from weight_fill import fill_module_
    from models.modules.swinTransformer import SwinTransformerBlock
    for tag, shift in (("blk_s0", 0), ("blk_s3", 3)):
        blk = SwinTransformerBlock(dim=96, input_resolution=(14, 14), num_heads=3, window_size=7, shift_size=shift)
        fill_module_(blk)
        blk.eval()                                           # DropPath is Identity at rate 0 either way
        x = seeded_randn(700 + shift, 2, 196, 96).requires_grad_(True)
        g = seeded_randn(710 + shift, 2, 196, 96)
        y = blk(x)
        (y * g).sum().backward()
        store[f"{tag}/y"] = y.detach().numpy().astype(np.float32)
        store[f"{tag}/dx"] = x.grad.numpy().astype(np.float32)
        for name, prm in blk.named_parameters():
            store[f"{tag}/grad/{name}"] = prm.grad.numpy().astype(np.float32)
        print(tag, float(y.abs().max()), float(x.grad.abs().max()), len(list(blk.named_parameters())), "param grads")
    np.savez_compressed(os.path.join(args.out, "train_tail.npz"), **store)
    print("train_tail.npz", os.path.getsize(os.path.join(args.out, "train_tail.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
