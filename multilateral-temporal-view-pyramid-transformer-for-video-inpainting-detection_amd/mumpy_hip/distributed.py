"""One process per GPU, clips sharded on the batch axis, ONE collective: a sum all-reduce of the metric vector
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  Replaces the reference's only multi-GPU
mechanism, torch.nn.DataParallel (test.py:56-58, train.py:290-292).

The forward couples the samples of a micro-batch (SwinDAttention pairs windows across the batch, deform:330,394), so
the partition unit is the MICRO-BATCH: rank r owns a contiguous slice of the global batch and its result equals the
reference forward run on that slice alone — not a slice of a whole-batch forward."""
import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_process_group(backend: str = "nccl", device: torch.device = None, timeout_s: float = None):
    """Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torch.distributed.run sets them).
    The rendezvous / collective timeout is explicit (MUMPY_DIST_TIMEOUT seconds, default 300): a rank that never arrives
    makes the others raise within minutes instead of sitting in the store for c10d's 10-30 minute default."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MUMPY_FORCE_DIST=1: build the process group even for ONE rank -- the rehearsal of the N-rank code path (RCCL communicator,
    # barriers, the metric all-reduce, hipGraph capture beside RCCL's watchdog thread) on a box with a single GPU
    force = os.environ.get("MUMPY_FORCE_DIST", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        if timeout_s is None:
            timeout_s = float(os.environ.get("MUMPY_DIST_TIMEOUT", "300"))
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return int(os.environ.get("RANK", "0")), world


def micro_batch_slice(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, equal slices (the remainder goes to the lowest ranks): [start, stop) of rank's micro-batch."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def eval_metric_vector(pred_mask: torch.Tensor, gt_mask: torch.Tensor) -> torch.Tensor:
    """float64[3] = [sum_i F1_i, sum_i IoU_i, n] over the rank's clips, per-image formulas of measure.py:57-62, 86-89
    (recall's denominator is sum(gt + 1e-6) over all pixels, as written there).  Runs on the tensors' device."""
    p = pred_mask.reshape(pred_mask.shape[0], -1).bool()
    g = gt_mask.reshape(gt_mask.shape[0], -1).bool()
    inter = (p & g).sum(1).double()
    union = (p | g).sum(1).double()
    recall = inter / (g.sum(1).double() + 1e-6 * p.shape[1])
    precision = inter / (p.sum(1).double() + 1e-6)
    f1 = 2 * precision * recall / (precision + recall + 1e-6)
    iou = (inter + 1e-5) / (union + 1e-5)
    n = torch.tensor(float(p.shape[0]), dtype=torch.float64, device=p.device)
    return torch.stack([f1.sum(), iou.sum(), n])


def all_reduce_metric(vec: torch.Tensor) -> torch.Tensor:
    """The single collective of the inference path (24 bytes: latency-bound, topology-irrelevant)."""
    if dist.is_available() and dist.is_initialized():
        if vec.is_cuda and dist.get_backend() == "gloo":     # CPU rehearsal backend: reduce on the host
            host = vec.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            return vec.copy_(host)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def max_over_ranks(seconds: float, device=None) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
