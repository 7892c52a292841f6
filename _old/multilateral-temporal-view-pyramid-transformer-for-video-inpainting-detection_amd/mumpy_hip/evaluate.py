"""Eval step of the reference (test.py:86-111 + measure.py:57-62,86-89) on device — SURVEY 8f-1.

    frames (uint8, optional) -> normalize_u8 -> Encoder -> Decoder.predict_mask (logits + thresholded uint8 mask from the
    same last kernel) -> per-clip F1 / IoU sums -> ONE all-reduce of the 3-float metric vector across ranks.
"""
import torch

from . import ops
from .distributed import all_reduce_metric, eval_metric_vector


@torch.no_grad()
def eval_step(encoder, decoder, x, gt_mask=None, thr=0.5):
    """x: (B,T,3,224,224) float32 clip, or (B,T,Hs,Ws,3) uint8 frames of any size (then the loader's resize to 224x224 --
    PIL NEAREST, universaldataset.py:75-79 -- and ToTensor+Normalize run on device in one kernel).
    Returns (mask uint8 (B,1,224,224), logits, metric_vector or None)."""
    if x.dtype == torch.uint8:
        x = ops.normalize_u8(x, size=(224, 224))
    fx, vx, dx = encoder(x)
    logits, mask, _ = decoder.predict_mask(fx, vx, dx, thr)
    metric = eval_metric_vector(mask, gt_mask) if gt_mask is not None else None
    return mask, logits, metric


def finalize_metrics(metric_sum: torch.Tensor):
    """All-reduce the accumulated [sum F1, sum IoU, n] vector (the path's only collective) -> (mean F1, mean IoU, n)."""
    v = all_reduce_metric(metric_sum.clone())
    n = float(v[2])
    return float(v[0]) / max(n, 1.0), float(v[1]) / max(n, 1.0), int(n)
