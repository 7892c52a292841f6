"""CPU, world_size 2, gloo: the N>1 host path — micro-batch partition, the single metric all-reduce, max-over-ranks
timing — gives the same answer as one process over the whole global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mumpy_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    from conftest import PKG  # noqa: F401  (puts the package on sys.path)
    from mumpy_hip import distributed as D
    r, w = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(global_batch, 1, 32, 32, generator=g)
    gt = torch.rand(global_batch, 1, 32, 32, generator=g) < 0.3
    a, b = D.micro_batch_slice(global_batch, world, rank)
    vec = D.eval_metric_vector(O.mask_from_logits(logits[a:b]), gt[a:b])
    vec = D.all_reduce_metric(vec)
    tmax = D.max_over_ranks(1.0 + rank)
    q.put((rank, vec.tolist(), tmax, (a, b)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [8, 7])
def test_two_rank_metric_allreduce(global_batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(global_batch, 1, 32, 32, generator=g)
    gt = torch.rand(global_batch, 1, 32, 32, generator=g) < 0.3
    ref = O.metric_vector(O.mask_from_logits(logits), gt)
    slices = [r[3] for r in res]
    assert slices[0][0] == 0 and slices[0][1] == slices[1][0] and slices[1][1] == global_batch   # contiguous cover
    for _, vec, tmax, _ in res:
        assert torch.allclose(torch.tensor(vec, dtype=torch.float64), ref, rtol=1e-12, atol=1e-12)
        assert tmax == 2.0


def test_micro_batch_slices_cover_and_balance():
    from mumpy_hip.distributed import micro_batch_slice
    for gb in (1, 7, 8, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [micro_batch_slice(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        micro_batch_slice(8, 2, 2)
