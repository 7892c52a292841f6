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


def test_bench_plain_multi_gpu_invocation_spawns_ranks():
    """`python bench.py --gpus 2` invoked plainly (no torch.distributed.run) must fan out into two rank processes itself
    and print ONE JSON line with n_gpus 2.  --rehearse swaps the forward for a sleep, so the launch path (spawn,
    rendezvous on 127.0.0.1, barrier, metric all-reduce, max-over-ranks) runs here on CPU over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MUMPY_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]       # (gloo itself prints a connection note on stdout)
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["rehearsal"] is True
    assert out["eval_metric"]["clips"] == 16          # both ranks' micro-batches went through the one all-reduce


def test_bench_spawned_rank_failure_ends_the_job_fast():
    """Rank 1 exits with code 3 before the rendezvous: the parent must notice (it supervises every child, not only rank 0),
    terminate rank 0 -- which is sitting in the store waiting for its peer --, exit non-zero with rank 1's code and stderr
    tail, and print no JSON line; well inside 30 s, not after the process-group timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MUMPY_BENCH_BACKEND="gloo", MUMPY_REHEARSE_DIE="1:3")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       env=env, capture_output=True, text=True, timeout=120)
    dt = time.monotonic() - t0
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert dt < 30, f"took {dt:.1f} s"
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
    assert "rank 1 exited with code 3" in r.stderr and "before rendezvous" in r.stderr


def test_bench_rejects_mismatched_world_before_touching_a_gpu():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_rccl_backend_executes_on_this_gpu():
    """The `nccl` (= RCCL) branch of the launch path on real hardware: a ONE-rank RCCL process group on cuda:0 (RCCL refuses two
    ranks on one device, and a gpurun box has one GPU) -- init with device_id, the metric all-reduce and the max-over-ranks timing
    reduction on device tensors, a barrier, teardown.  It cannot show scaling; it shows that the code the driver's N-GPU run takes
    (distributed.init_process_group("nccl", dev), all_reduce_metric on a GPU tensor) runs through RCCL on this image."""
    import subprocess
    import sys
    from conftest import PKG, ROOT
    code = r'''
import os, sys, datetime, torch, torch.distributed as dist
sys.path[:0] = [%r, %r]
from mumpy_hip import distributed as D
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)      # what D.init_process_group does for world > 1
assert dist.get_backend() == "nccl"
pred = torch.rand(4, 1, 32, 32, device=dev) > 0.5
gt = torch.rand(4, 1, 32, 32, device=dev) > 0.5
v = D.eval_metric_vector(pred, gt)
ref = v.clone()
dist.all_reduce(v, op=dist.ReduceOp.SUM)                    # the one collective of the path, on a device tensor, through RCCL
assert torch.equal(v, ref) and v.is_cuda
assert D.max_over_ranks(1.25, dev) == 1.25
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl ok")
''' % (PKG, ROOT)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_bench_rank_path_with_a_live_rccl_group_on_one_gpu():
    """bench.py exactly as a rank of the driver's N-GPU launch runs it -- RANK / WORLD_SIZE / MASTER_* from the environment,
    backend nccl, RCCL communicator alive while the forward is captured into a hipGraph (RCCL's watchdog thread polls events
    meanwhile: the capture uses thread-local error mode), barriers around the timed loop, the metric all-reduce and the
    max-over-ranks reduction through RCCL -- with WORLD_SIZE = 1 forced through the process-group path (MUMPY_FORCE_DIST=1),
    because a gpurun box has one GPU.  Checks the JSON line's contract fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", MUMPY_FORCE_DIST="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-alt",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["unit"] == "clips/s" and out["value"] > 50
    assert out["scaling"] == "weak" and out["eval_metric"]["clips"] == 8 and out["roofline"]["frac"] > 0.3


@pytest.mark.gpu
def test_train_rank_path_with_a_live_rccl_group_on_one_gpu():
    """Config 5's launchable job (tools/train_ddp_bench.py --graph) as ONE forced rank: two hipGraphs (forward + backward | AdamW)
    with the bucketed gradient all-reduce issued through RCCL between their replays, the replica check through RCCL MIN / MAX."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29521", MUMPY_FORCE_DIST="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "train_ddp_bench.py"), "--batch", "1", "--frames", "3", "--steps", "2",
                        "--math", "bf16", "--graph"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["graph"] is True and out["replicas_identical_after_steps"] is True
    assert all(v == v for v in out["loss"])            # finite
