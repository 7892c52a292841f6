#!/usr/bin/env python3
"""Config 5 as a launchable job: data-parallel training step of the full three-view model, one process per GPU.
Each rank owns a micro-batch (default B=2: config 5's B=16 over 8 GPUs), runs forward + mask loss + backward on the HIP kernels,
sums the flat gradient across ranks in buckets (RCCL over xGMI with backend nccl) and applies the fused AdamW with the
1/world factor.  Prints ONE JSON line on rank 0 (whole-job clips/s, max over ranks).

  python tools/train_ddp_bench.py [--batch 2] [--frames 5] [--steps 5] [--math fp32|bf16|bf16x3]            # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_ddp_bench.py ...
  MUMPY_BENCH_BACKEND=gloo ...   rehearses the launch on a box with fewer GPUs than ranks (ranks share devices; buckets staged on host)
"""
import argparse, json, os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2); ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--math", choices=["fp32", "bf16", "bf16x3"], default="bf16")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--train-mode", action="store_true", help=".train(): stochastic depth on (masks from torch's graph-safe generator)")
    ap.add_argument("--graph", action="store_true", help="replay the step from hipGraphs (forward+backward | AdamW), all-reduce eager between them")
    args = ap.parse_args()
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    backend = os.environ.get("MUMPY_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // max(world, 1))))      # (a box reports all host cores; a rank owns a share)
    from mumpy_hip import distributed as D, ops
    D.init_process_group(backend, dev)
    from models.decoder.decoder import Decoder
    from models.encoder.encoder import Encoder
    from mumpy_hip.autograd import decoder_train, encoder_train
    from mumpy_hip.train import build_optimizers
    from weight_fill import fill_module_, seeded_randn
    ops.set_matrix_math(args.math)
    enc = fill_module_(Encoder(num_frames=args.frames)).eval().to(dev)          # same weights on every rank (deterministic fill)
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, args.frames])).eval().to(dev)
    if args.train_mode:
        enc.train(); dec.train()
    opts = build_optimizers(enc, dec, lr_cnn=1e-6, lr=1e-5, lr_cva=1e-6, weight_decay=1e-4, weight_decay_cnn=1e-4)
    x = seeded_randn(100 + rank, args.batch, args.frames, 3, 224, 224).to(dev)
    target = (torch.rand(args.batch, 1, 224, 224, generator=torch.Generator().manual_seed(7 + rank)) < 0.1).float().to(dev)

    def step():
        fx, vx, dx = encoder_train(enc, x)
        logits, _ = decoder_train(dec, fx, vx, dx)
        loss3, dlogits = ops.mask_loss(logits.detach(), target)
        logits.backward(dlogits)
        for o in opts.values():
            scale = o.all_reduce_grads(bucket_bytes=args.bucket_mb << 20)       # the step's collectives
            o.step(grad_scale=scale)
            o.zero_grad()
        return loss3

    if args.graph:
        from mumpy_hip.train import GraphedTrainStep
        gs = GraphedTrainStep(lambda xx: decoder_train(dec, *encoder_train(enc, xx))[0], opts, x, target, warmup=max(args.warmup, 2),
                              all_reduce=world > 1 or os.environ.get("MUMPY_FORCE_DIST", "0") == "1")
        step = gs.step
        args.warmup = 1
    for _ in range(args.warmup):
        step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss3 = step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev if backend == "nccl" else None)
    # every rank must hold the same parameters after the same steps (same init, summed gradients)
    chk = torch.stack([o.param.double().sum() for o in opts.values()])
    if backend != "nccl":
        chk = chk.cpu()          # gloo reduces host tensors; RCCL needs the tensor on this rank's GPU
    same = True
    if dist.is_initialized():
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(lo, hi))
    if rank == 0:
        print(json.dumps({"metric": "train clips/s (fwd + loss + bwd + grad all-reduce + AdamW)", "value": round(args.batch * world * args.steps / dt, 3),
                          "unit": "clips/s", "n_gpus": world, "ms_per_step": round(1e3 * dt / args.steps, 2), "graph": bool(args.graph), "micro_batch": args.batch,
                          "frames": args.frames, "math": args.math, "train_mode": bool(args.train_mode), "backend": backend, "loss": [round(float(v), 5) for v in loss3],
                          "replicas_identical_after_steps": same}))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
