#!/usr/bin/env python3
"""bench.py — forward clips/s of the Mumpy hot path (encoder + decoder) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): per rank a micro-batch of B=8 synthetic clips, T=5, 224x224, fp32, full Mumpy
(three temporal views with tubelets (5,4,1), pyramid decoder), synthetic deterministic weights.  A "step" is one forward
of one micro-batch per rank (weak scaling: clips are independent across micro-batches, so ranks shard the batch axis with
no data-path collective; the only collective is ONE all-reduce of the 3-float metric vector after the timed loop's last
step — SURVEY 8e).  Inputs are resident in HBM before the timed region.  The forward is replayed from one hipGraph.

Prints ONE JSON line (rank 0) with the driver's contract plus:
  "roofline":     the kernel with the largest share of device time (measured live with events on the launch stream in an
                  eager pass of the same forward), algorithmic FLOPs or bytes / its summed duration vs peak;
  "kernels":      the same figures for every C-ABI kernel, incl. the two the north_star names
                  (mumpy_window_attention_fwd: MFMA; mumpy_deform_sample_fwd: HBM);
  "cpu_baseline": the oracle (CPU restatement, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak
HBM_BOUND = {"mumpy_deform_sample_fwd", "mumpy_layernorm_fwd", "mumpy_gn_stats_nhwc_fwd", "mumpy_gn_apply_resample_nhwc_fwd", "mumpy_final_conv_fwd", "mumpy_patch_merge_ln_fwd",
             "mumpy_add_fwd"}
GFLOP_PER_CLIP_T5 = 253.9         # whole-forward algorithmic work at T=5


T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (micro-batch)")
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--math", choices=["fp32", "bf16", "bf16x3", "bf16x2"], default="fp32",
                    help="matrix arithmetic of the GEMMs/convolutions: fp32 on the fp32 MFMA (headline, default), bf16x3 = fp32 "
                         "products from three bf16 pieces per operand on the bf16 MFMA (fp32-level accuracy), bf16 = bf16 "
                         "operands + fp32 accumulate")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: BASELINE config 3 as written (bf16 activations / weights in HBM inside the Swin blocks, bf16 MFMA, "
                         "f32 accumulate / statistics / residual stream); never the headline configuration")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x3 measurement reported beside the fp32 headline")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="launch-path rehearsal without a GPU: ranks, rendezvous, barrier, the metric all-reduce and the JSON "
                         "line are real, the forward is replaced by a sleep (CPU tests; MUMPY_BENCH_BACKEND=gloo)")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    return ap.parse_args()


def profile_kernels(enc, dec, x):
    """One eager forward with every C-ABI launch bracketed by events on its launch stream."""
    from mumpy_hip import ops, streams
    was_serial, streams.SERIAL = streams.SERIAL, True        # per-kernel durations: no co-scheduled branches
    with torch.no_grad():
        dec(*enc(x))
        torch.cuda.synchronize()
        ops.PROFILE = {}
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        dec(*enc(x))
        t1.record()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    streams.SERIAL = was_serial
    total_ms = t0.elapsed_time(t1)
    # one GEMM entry point for the roofline: mumpy_linear_lnx_fwd is mumpy_linear_wsz_fwd with the LayerNorm statistics / finish in
    # its epilogue (same kernels, same shapes) -- reported together so that the figure covers EVERY nn.Linear launch, not the
    # large ones only
    if "mumpy_linear_lnx_fwd" in prof:
        prof.setdefault("mumpy_linear_wsz_fwd", []).extend(prof.pop("mumpy_linear_lnx_fwd"))
    rows = []
    for name, evs in prof.items():
        ms = sum(a.elapsed_time(b) for a, b, _ in evs)
        work = sum(w for _, _, w in evs)
        row = {"kernel": name, "launches": len(evs), "ms": round(ms, 4), "avg_us": round(1e3 * ms / len(evs), 2)}
        if work > 0 and ms > 0:
            if name in HBM_BOUND:
                ach = work / (ms * 1e-3) / 1e9
                row.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4))
            else:
                ach = work / (ms * 1e-3) / 1e12
                row.update(bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                           frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4))
        rows.append(row)
    rows.sort(key=lambda r: -r["ms"])
    return rows, total_ms


def north_star_kernels(batch, frames, dev):
    """The two kernels BASELINE.json names, timed on their LARGEST launch of this workload (stage 0: view 3's shifted
    window attention; view 2 <- view 3 deformable sampling) with events on the launch stream, 20 launches each."""
    from models.modules.swinTransformer import build_shift_mask, relative_position_index
    from mumpy_hip import ops
    out = {}
    hs, w, c = frames * 56, 56, 128
    qkv = torch.randn(batch, hs * w, 3 * c, device=dev)
    bias = ops.expand_relpos_bias(torch.randn(169, c // 32, device=dev) * 0.2, relative_position_index(7, 7).to(dev))
    tab, ids = ops.compact_attn_mask(build_shift_mask(hs, w, 7, 3).to(dev))

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    t = timed(lambda: ops.window_attention(qkv, bias, batch, hs, w, c, 3, 32 ** -0.5, tab, ids))
    units = batch * (hs // 7) * (w // 7) * (c // 32)
    ach = units * 307328.0 / t / 1e12
    out["window_attention"] = {"kernel": "win_attn_self_kernel", "launch": f"B={batch}, grid {hs}x{w}, C={c}, shift 3: {units} (window,head) units",
                               "bound": "mfma", "flop_per_unit": 307328, "avg_us": round(t * 1e6, 2), "achieved": round(ach, 2),
                               "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                               "hbm_bytes_per_unit": 25088, "hbm_gbs": round(units * 25088 / t / 1e9, 1)}
    c2 = 96
    nwin = batch * (hs // 7) * (w // 7)
    x2 = torch.randn(batch, hs * w, c2, device=dev)
    pos = torch.rand(batch * 64, 3, 49, 2, device=dev) * 2 - 1
    t = timed(lambda: ops.deform_sample(x2, pos, batch, hs, w, c2, batch * 64))
    nbytes = 4.0 * (2 * nwin * 49 * c2 + nwin * 3 * 49 * 2)
    ach = nbytes / t / 1e9
    out["deform_sample"] = {"kernel": "deform_sample_lds_kernel<96>", "launch": f"{nwin} kv windows x 49 points x {c2} ch",
                            "bound": "hbm", "bytes_per_launch": int(nbytes), "avg_us": round(t * 1e6, 2), "achieved": round(ach, 1),
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4)}
    return out


def cpu_baseline(frames, sample_b):
    """Oracle ('port' of the reference CPU path) on this box's host cores, bounded sample."""
    from oracle import mumpy_oracle as O
    from weight_fill import fill_state_dict_, seeded_randn
    man_e = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_encoder_t5.json" if frames == 5 else "state_dict_encoder.json")))
    man_d = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_decoder_t5.json" if frames == 5 else "state_dict_decoder.json")))
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = max(1, min(cores, int(os.environ.get("MUMPY_CPU_THREADS", "16"))))   # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    log(f"  cpu baseline on {cores} threads")

    def build(man):
        from models.modules.swinTransformer import build_shift_mask, relative_position_index
        sd = {}
        for k, (shape, dt) in man.items():
            if k.endswith("relative_position_index"):
                sd[k] = relative_position_index(7, 7)
            elif k.endswith("attn_mask"):
                side = [56, 28, 14, 7][int(k.split("layers.layers.")[1].split(".")[0])]
                sd[k] = build_shift_mask(49 * shape[0] // side, side, 7, 3)
            else:
                sd[k] = torch.zeros(shape, dtype=getattr(torch, dt))
        return fill_state_dict_(sd)

    sde, sdd = build(man_e), build(man_d)
    x = seeded_randn(1234, sample_b, frames, 3, 224, 224)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.full_forward(sde, sdd, x[:1])                      # warm-up (thread pools, allocator)
        log(f"  warm-up B=1 pass: {time.perf_counter() - t0:.2f} s")
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.full_forward(sde, sdd, x)
            ts.append(time.perf_counter() - t0)
            log(f"  B={sample_b} pass: {ts[-1]:.2f} s")
    t = min(ts)
    # how the oracle compares with the reference's own CPU forward (measured where both can run: the build container,
    # tools/cpu_ref_ratio.py -> tests/golden/cpu_ref_ratio.json; the round-1 review measured 1.50 vs 0.98 clips/s there)
    ratio = None
    try:
        r = json.load(open(os.path.join(ROOT, "tests", "golden", "cpu_ref_ratio.json")))
        ratio = {"oracle_over_reference": r["oracle_over_reference"], "oracle_clips_s": r["oracle_clips_s"],
                 "reference_clips_s": r["reference_clips_s"], "where": "build container, 8 vCPU (tools/cpu_ref_ratio.py)",
                 "reference_equivalent_value": round(sample_b / t / r["oracle_over_reference"], 3)}
    except (OSError, KeyError):
        pass
    return {"value": round(sample_b / t, 3), "unit": "clips/s", "cores": cores, "kind": "port", "ref_ratio": ratio,
            "sample": f"oracle full forward, B={sample_b}, T={frames}, 224x224 fp32, 1 warm-up (B=1) + 3 timed, best of 3 "
                      f"({t:.2f} s per pass)"}


def spawn_ranks(args):
    """`python bench.py --gpus N` invoked plainly (no torch.distributed.run): start N fresh child processes, one rank per
    GPU, supervise ALL of them and relay rank 0's JSON line.  Children, never a re-exec: this parent has not touched the
    GPU and never will (replaces the reference's in-process nn.DataParallel fan-out, test.py:56-58).
    Fail fast: the first rank that exits non-zero ends the job -- the others are terminated (SIGTERM, then SIGKILL), the
    parent exits non-zero with that rank's stderr tail and prints no JSON line; ranks that ran into a taken rendezvous
    port (picked by bind-then-close, so another process can grab it in between) are relaunched on a fresh port."""
    import socket
    import subprocess
    import tempfile
    deadline = time.monotonic() + float(os.environ.get("MUMPY_BENCH_JOB_TIMEOUT", "3000"))

    def launch(port, tmp):
        procs, logs = [], []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            out = open(os.path.join(tmp, f"rank{r}.out"), "wb")       # files, not pipes: no rank can block on a full pipe
            err = open(os.path.join(tmp, f"rank{r}.err"), "wb")
            logs.append((out, err))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
        return procs, logs

    def tail(tmp, r, n=2000):
        try:
            with open(os.path.join(tmp, f"rank{r}.err"), "rb") as f:
                return f.read()[-n:].decode(errors="replace")
        except OSError:
            return ""

    def stop(procs):
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    for attempt in range(3):
        with socket.socket() as sk:
            sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        with tempfile.TemporaryDirectory(prefix="mumpy_bench_") as tmp:
            procs, logs = launch(port, tmp)
            failed = None
            shown = 0
            try:
                while True:
                    rcs = [p.poll() for p in procs]
                    # relay rank 0's progress lines (stderr) as they come
                    try:
                        with open(os.path.join(tmp, "rank0.err"), "rb") as f:
                            f.seek(shown)
                            chunk = f.read()
                        if chunk:
                            shown += len(chunk)
                            sys.stderr.write(chunk.decode(errors="replace"))
                            sys.stderr.flush()
                    except OSError:
                        pass
                    bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
                    if bad:
                        failed = bad[0]
                        break
                    if all(rc == 0 for rc in rcs):
                        break
                    if time.monotonic() > deadline:
                        failed = -1
                        break
                    time.sleep(0.05)
            finally:
                stop(procs)
                for out, err in logs:
                    out.close()
                    err.close()
            if failed is None:
                with open(os.path.join(tmp, "rank0.out"), "rb") as f:
                    sys.stdout.write(f.read().decode())
                sys.stdout.flush()
                return
            if failed == -1:
                raise SystemExit("bench.py: job timeout (MUMPY_BENCH_JOB_TIMEOUT) -- ranks terminated, no result")
            rc, msg = procs[failed].returncode, tail(tmp, failed)
            port_taken = any(("EADDRINUSE" in tail(tmp, r, 8000) or "address already in use" in tail(tmp, r, 8000).lower())
                             for r in range(args.gpus))
        if port_taken and attempt < 2:
            print(f"[bench] rendezvous port {port} was taken; relaunching the ranks on a fresh port", file=sys.stderr, flush=True)
            continue
        sys.stderr.write(f"[bench] rank {failed} exited with code {rc}; the other ranks were terminated.  Its stderr tail:\n{msg}\n")
        raise SystemExit(rc if isinstance(rc, int) and 0 < rc < 256 else 1)


def rehearse(args, world, rank):
    """The N-rank launch path with the forward replaced by a sleep: same rendezvous, barriers, metric all-reduce,
    max-over-ranks timing and JSON line as the real run.  Needs no GPU (gloo)."""
    from mumpy_hip import distributed as D
    backend = os.environ.get("MUMPY_BENCH_BACKEND", "gloo")
    die = os.environ.get("MUMPY_REHEARSE_DIE", "")           # "<rank>:<code>": that rank exits before the rendezvous
    if die and int(die.split(":")[0]) == rank:
        print(f"rehearsal: rank {rank} exiting with code {die.split(':')[1]} before rendezvous", file=sys.stderr, flush=True)
        raise SystemExit(int(die.split(":")[1]))
    D.init_process_group(backend, None)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    metric = D.all_reduce_metric(torch.tensor([0.5 * args.batch, 0.25 * args.batch, float(args.batch)], dtype=torch.float64))
    if world > 1:
        dist.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, None)
    if rank == 0:
        print(json.dumps({"metric": "clips/sec fwd (B=8,T=5,224x224)", "value": None, "unit": "clips/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "rehearsal": True, "config": {"workload": "launch-path rehearsal (no forward)", "global_batch": args.batch * world,
                                                        "parallelism": f"batch-sharded x{world}, one metric all-reduce"},
                          "eval_metric": {"f1": float(metric[0] / metric[2]), "iou": float(metric[1] / metric[2]), "clips": int(metric[2])}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    # first statement, before anything can initialise a GPU: settle the launch.  Under torch.distributed.run the
    # environment carries the rank; a plain `python bench.py --gpus N` fans out into N children here.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return spawn_ranks(args)
        world, rank, local = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with `python -m torch.distributed.run "
                             f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...` or plainly as `python bench.py --gpus {args.gpus}`")
    if args.rehearse:
        return rehearse(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP kernels are the only implementation (no CPU path)")
    # MUMPY_BENCH_BACKEND=gloo rehearses the N>1 launch path on a box with fewer GPUs than ranks (ranks share devices);
    # the driver's runs use the default: RCCL, one GPU per rank.
    backend = os.environ.get("MUMPY_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host threads: N ranks build their weights concurrently on one node; keep each within its share of the cores
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // max(world, 1))))
    from mumpy_hip import distributed as D
    D.init_process_group(backend, dev)                       # "nccl" = RCCL over xGMI (no-op at world 1)

    from models.decoder.decoder import Decoder
    from models.encoder.encoder import Encoder
    from mumpy_hip import ops
    from mumpy_hip.graph import GraphedForward
    from mumpy_hip.pipeline import fused_forward
    from weight_fill import fill_module_, seeded_randn

    ops.set_matrix_math(args.math)
    if args.storage == "bf16":
        ops.set_storage("bf16")                   # (implies the bf16 matrix-math mode for the GEMMs outside the Swin blocks)
        args.math = "bf16s"
    log("building model + synthetic weights")
    enc = fill_module_(Encoder(num_frames=args.frames).eval()).to(dev)
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, args.frames]).eval()).to(dev)
    x = seeded_randn(1234 + rank, args.batch, args.frames, 3, 224, 224).to(dev)         # resident before timing
    gt = (torch.rand(args.batch, 1, 224, 224, generator=torch.Generator().manual_seed(99 + rank)) < 0.1).to(dev)

    log("warm-up (eager)")
    with torch.no_grad():
        for i in range(max(args.warmup, 1)):
            fx, vx, dx = enc(x)
            torch.cuda.synchronize()
            if i == 0:
                log("  first encoder forward done")
            logits = dec(fx, vx, dx)[0]
            torch.cuda.synchronize()
            if i == 0:
                log("  first decoder forward done")
    log("capturing hipGraph")
    fwd = None if args.no_graph else GraphedForward(enc, dec, x, with_mask=True)
    log("timing")

    def step():                     # one forward of the micro-batch, thresholded mask included (fused in the last kernel)
        if fwd is None:
            return fused_forward(enc, dec, x, with_mask=True)[1]
        return fwd(x)[1]

    # untimed warm-up of the timed path itself: the W steps the caller asked for, and at least ~0.3 s of replays -- a GPU that
    # idled through model construction needs that long to reach its steady clock (a cold first measurement of the short bf16
    # step read 18.8 ms against 9.9 ms warm on the same box)
    t_w = time.perf_counter()
    n_w = 0
    while n_w < max(args.warmup, 2) or time.perf_counter() - t_w < 0.3:
        step()
        n_w += 1
        if n_w % 4 == 0:
            torch.cuda.synchronize()

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mask = step()
    # per-image F1/IoU on device (measure.py:57-62,86-89) + the ONE collective of the path: all-reduce of the metric vector
    metric = D.all_reduce_metric(D.eval_metric_vector(mask, gt))
    barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev if backend == "nccl" else None)
    ops.check_workspaces()          # (outside the timed region) no split GEMM launch gave up on a partial tile: the masks are complete

    if rank == 0:
        log(f"timed: {1e3 * dt / args.steps:.2f} ms/step; profiling kernels")
        kernels, eager_ms = profile_kernels(enc, dec, x)
        dom = dict(kernels[0])
        # HBM-side bytes per launch of the dominant entry point from the PMC passes of the same workload (FETCH_SIZE x2 +
        # WRITE_SIZE; tools/pmc_bench.sh + tools/summarize_pmc_bench.py -> profiles/): bench.py cannot run rocprofv3 on itself
        dom["traffic"] = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_bench_traffic.json")))
            if args.math == "fp32" and args.batch == 8 and args.frames == 5 and dom["kernel"] in pmc:
                dom["traffic"] = pmc[dom["kernel"]]["hbm_bytes_per_launch"]
                dom["traffic_unit"] = "B per launch (PMC: profiles/r03_pmc_bench_traffic.md)"
        except OSError:
            pass
        dom.pop("launches", None)
        dom.pop("ms", None)
        clips = args.batch * world * args.steps
        out = {
            "metric": "clips/sec fwd (B=8,T=5,224x224)", "value": round(clips / dt, 3), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "f32 (matrix products from 3 bf16 pieces per operand on the bf16 MFMA, f32 accumulate)",
                      "bf16x2": "matrix operands as 2 bf16 pieces (16 mantissa bits, TF32-class), f32 accumulate and storage",
                      "bf16": "bf16 matrix operands, f32 accumulate and storage",
                      "bf16s": "bf16 activations / weights in HBM and bf16 MFMA operands inside the Swin blocks; f32 accumulate, statistics "
                               "and residual stream (config 3)"}[args.math], "data": "synthetic",
            "config": {"workload": f"full Mumpy forward (3 temporal views, pyramid decoder), B={args.batch} clips/GPU, "
                                   f"T={args.frames}, 224x224, {'bf16 storage' if args.math == 'bf16s' else 'fp32'}, "
                                   f"tubelets ({args.frames},{args.frames - 1},1)",
                       "global_batch": args.batch * world, "launch": ("eager" if fwd is None else "hipGraph replay") + ", fork/join over HIP streams (independent branches co-scheduled)",
                       "parallelism": f"batch-sharded x{world}, one metric all-reduce"},
            "roofline": dom,
            "north_star_kernels": north_star_kernels(args.batch, args.frames, dev),
            "kernels": kernels,
            "forward_gflop_per_clip": GFLOP_PER_CLIP_T5 if args.frames == 5 else None,
            "whole_forward_frac_of_f32_mfma_peak": round(GFLOP_PER_CLIP_T5 * 1e9 * clips / dt / world / (PEAK_F32_MFMA_TFLOPS * 1e12), 4)
            if args.frames == 5 else None,
            "serial_eager_forward_ms": round(eager_ms, 3),
            "eval_metric": {"f1": float(metric[0] / metric[2]), "iou": float(metric[1] / metric[2]), "clips": int(metric[2])},
        }
        if args.math == "fp32" and world == 1 and not args.no_alt:
            # the same workload with the GEMMs / convolutions in the split-precision modes: bf16x3 (fp32-level accuracy, tests:
            # test_linear_bf16x3_math, test_full_model_bf16x3_math_t5) and bf16x2 (16-bit-mantissa operands, TF32-class:
            # test_linear_bf16x2_math) -- reported beside the headline, never as `value`
            for mode, what in (("bf16x3", "fp32 products as 6 bf16 piece products (3 pieces per operand), f32 accumulate"),
                               ("bf16x2", "3 bf16 piece products (2 pieces = 16 mantissa bits per operand), f32 accumulate")):
                try:
                    log(f"alt: {mode} matrix math")
                    ops.set_matrix_math(mode)
                    with torch.no_grad():
                        fused_forward(enc, dec, x, with_mask=True)
                    fwd3 = None if args.no_graph else GraphedForward(enc, dec, x, with_mask=True)
                    run3 = (lambda: fused_forward(enc, dec, x, with_mask=True)[1]) if fwd3 is None else (lambda: fwd3(x)[1])
                    for _ in range(max(args.warmup, 2)):
                        run3()
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    for _ in range(args.steps):
                        mask3 = run3()
                    torch.cuda.synchronize()
                    dt3 = time.perf_counter() - t3
                    out["alt_" + mode] = {"math": what, "value": round(args.batch * args.steps / dt3, 3), "unit": "clips/s",
                                          "ms_per_step": round(1e3 * dt3 / args.steps, 3),
                                          "mask_pixels_differing_from_fp32_path": int((mask3 != mask).sum())}
                    del fwd3
                except Exception as e:          # the extra measurements must never cost the headline line
                    out["alt_" + mode] = {"error": repr(e)[:200]}
                finally:
                    ops.set_matrix_math("fp32")
        if args.math == "fp32" and world == 1 and not args.no_alt:
            # BASELINE config 3's per-GPU workload as written: bf16 STORAGE inside the Swin blocks (LayerNorm output, qkv,
            # attention output and the 4C MLP hidden tensor are bf16 in HBM, weights read as bf16) + bf16 matrix math for every
            # other GEMM / convolution; fp32 accumulate, statistics and residual stream (test_full_model_bf16_storage_b8_t5)
            try:
                log("alt: bf16 storage")
                ops.set_storage("bf16")
                with torch.no_grad():
                    fused_forward(enc, dec, x, with_mask=True)
                fwd16 = None if args.no_graph else GraphedForward(enc, dec, x, with_mask=True)
                run16 = (lambda: fused_forward(enc, dec, x, with_mask=True)[1]) if fwd16 is None else (lambda: fwd16(x)[1])
                for _ in range(max(args.warmup, 2) + 8):
                    run16()
                torch.cuda.synchronize()
                t3 = time.perf_counter()
                for _ in range(args.steps):
                    mask16 = run16()
                torch.cuda.synchronize()
                dt3 = time.perf_counter() - t3
                out["alt_bf16_storage"] = {"what": "config 3 arithmetic and storage: bf16 activations/weights in HBM inside the Swin blocks, "
                                                   "bf16 MFMA products everywhere, f32 accumulate / statistics / residual stream",
                                           "value": round(args.batch * args.steps / dt3, 3), "unit": "clips/s",
                                           "ms_per_step": round(1e3 * dt3 / args.steps, 3),
                                           "mask_pixels_differing_from_fp32_path": int((mask16 != mask).sum())}
                del fwd16
            except Exception as e:
                out["alt_bf16_storage"] = {"error": repr(e)[:200]}
            finally:
                ops.set_storage("fp32")
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle on host cores)")
            try:
                out["cpu_baseline"] = cpu_baseline(args.frames, args.cpu_sample_batch)
            except Exception as e:                   # reported baseline only: never lose the measured line over it
                out["cpu_baseline"] = {"error": repr(e)[:200]}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
