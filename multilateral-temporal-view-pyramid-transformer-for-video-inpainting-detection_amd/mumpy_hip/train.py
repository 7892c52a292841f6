"""Training tail of config 5: parameter groups, fused AdamW on flat buffers, the polynomial LR schedule and
the bucketed gradient all-reduce.  The model backward itself is not part of this round (forward kernels only), so these
pieces are exercised on their own: loss + gradient wrt the logits, optimizer update, schedule, collective.

Reference behaviour reproduced:
  * three optimizers: encoder parameters whose name contains "cva" / the other encoder parameters / the decoder
    (train.py:198-213), each `torch.optim.AdamW(lr, weight_decay)` with torch defaults otherwise (utils/utils.py:258);
  * `PolynomialLR` (utils/optimizer/scheduler.py:6-43) with power 0.9, min_lr 1e-5, step_size 1, no warm-up
    (train.py:226-262), stepped once per optimizer step;
  * gradient accumulation: loss / accumulation_steps (train.py:115), update every accumulation_steps iterations;
  * nn.DataParallel's gradient (grad of the mean loss over the global batch) == the mean over ranks of per-rank gradients
    for equal shards: one sum all-reduce of the flat gradient buffer in buckets + the 1/world factor folded into AdamW.
"""
from typing import Dict, Iterable, List, Optional

import os

import torch
import torch.distributed as dist

from . import ops
from .state import bump_weights_epoch


def split_param_groups(encoder: torch.nn.Module, decoder: torch.nn.Module) -> Dict[str, List[torch.nn.Parameter]]:
    """train.py:198-213: {"cva": encoder params with "cva" in the name, "enc": the other encoder params, "dec": decoder}."""
    groups = {"cva": [], "enc": [], "dec": [p for p in decoder.parameters() if p.requires_grad]}
    for name, p in encoder.named_parameters():
        if p.requires_grad:
            groups["cva" if "cva" in name else "enc"].append(p)
    return groups


def polynomial_lr(base_lr: float, current_lr: float, it: int, iter_max: int, power: float = 0.9, min_lr: float = 1e-5,
                  iter_warmup: int = 0, step_size: int = 1) -> float:
    """Learning rate after the `it`-th scheduler step (scheduler.py:24-41, `last_epoch` = it).  Faithful to its guards:
    the rate is left unchanged at it == 0, when it is not a multiple of step_size, and past iter_max."""
    iter_max, iter_warmup = int(iter_max), int(iter_warmup)
    if it == 0 or it % step_size != 0 or it > iter_max:
        return current_lr
    if it < iter_warmup:
        coef = it / iter_warmup * (1 - iter_warmup / iter_max) ** power
    else:
        coef = (1 - it / iter_max) ** power
    return (base_lr - min_lr) * coef + min_lr


class FlatAdamW:
    """One parameter group of the reference's AdamW, held as flat fp32 buffers: the parameters are re-pointed at views
    of `self.param`, their `.grad` at views of `self.grad`, so a step is ONE kernel over the group and the gradient
    all-reduce runs over one contiguous buffer.  State layout (exp_avg, exp_avg_sq, step) matches torch.optim.AdamW."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, weight_decay: float = 1e-2, betas=(0.9, 0.999),
                 eps: float = 1e-8):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: empty parameter group")
        dev = self.params[0].device      # buffers can be built anywhere; step() needs the GPU (the HIP kernel has no CPU twin)
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]          # 16-B aligned slots
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.param = torch.zeros(n, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            view = self._slot(self.param, p, o)
            view.copy_(p.data)
            p.data = view
            p.grad = self._slot(self.grad, p, o)
            p._mumpy_flat_grad = p.grad           # marker: the backward kernels may accumulate into this view (autograd._grad_slot)
        self.base_lr = self.lr = lr
        self.weight_decay, self.betas, self.eps = weight_decay, betas, eps
        self.steps = 0            # optimizer steps taken
        self.sched_it = 0         # scheduler steps taken (PolynomialLR.last_epoch)

    @staticmethod
    def _slot(buf, p, o):
        """The view of flat buffer `buf` that has p's logical shape.  Spatial nn.Conv2d weights (Cout,Cin,kh,kw) are stored in
        channels_last order: the (Cout,kh,kw,Cin) image the implicit-GEMM kernels read is then a VIEW of the parameter (no permuted
        copy per step) and the weight-gradient kernel accumulates straight into the matching view of the gradient.  Logical shapes,
        state_dict keys / shapes / dtypes are unchanged."""
        if p.dim() == 4 and p.shape[1] > 1 and p.shape[2] * p.shape[3] > 1:
            co, ci, kh, kw = p.shape
            return buf[o:o + p.numel()].view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return buf[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()

    # ---- checkpointing: torch.optim.AdamW's state_dict layout, so a file written here reads like the reference's
    # enc_opt_{e}.pt / dec_opt_{e}.pt (utils/utils.py:264-276) and vice versa; the counters torch keeps elsewhere
    # (scheduler position, base rate) ride in an extra "mumpy" entry that torch's loader ignores
    def state_dict(self) -> dict:
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self._slot(self.exp_avg, p, o).detach().contiguous().clone(),
                        "exp_avg_sq": self._slot(self.exp_avg_sq, p, o).detach().contiguous().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group], "mumpy": {"steps": self.steps, "sched_it": self.sched_it, "base_lr": self.base_lr}}

    def load_state_dict(self, sd: dict) -> None:
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError(f"FlatAdamW.load_state_dict: expected one group of {len(self.params)} parameters")
        g = groups[0]
        self.lr, self.betas, self.eps, self.weight_decay = float(g["lr"]), tuple(g["betas"]), float(g["eps"]), float(g["weight_decay"])
        steps = 0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            n = p.numel()
            if st is None:                                   # torch omits parameters that never received a gradient
                self.exp_avg[o:o + n].zero_(); self.exp_avg_sq[o:o + n].zero_()
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdamW.load_state_dict: parameter {i} has shape {tuple(p.shape)}, state {tuple(st['exp_avg'].shape)}")
            self._slot(self.exp_avg, p, o).copy_(st["exp_avg"])
            self._slot(self.exp_avg_sq, p, o).copy_(st["exp_avg_sq"])
            steps = max(steps, int(float(st["step"])))
        extra = sd.get("mumpy", {})
        self.steps = int(extra.get("steps", steps))          # (one step count per group: every parameter steps together here)
        self.sched_it = int(extra.get("sched_it", self.steps))
        self.base_lr = float(extra.get("base_lr", g.get("initial_lr", self.lr)))

    def all_reduce_grads(self, bucket_bytes: int = 64 << 20):
        """Sum all-reduce of the flat gradient in buckets (RCCL ring over xGMI: per-link bound, so a few tens of MB per
        call keeps the ring busy without delaying the first bucket); returns the factor AdamW must apply (1/world)."""
        if not (dist.is_available() and dist.is_initialized()) or \
                (dist.get_world_size() == 1 and os.environ.get("MUMPY_FORCE_DIST", "0") != "1"):    # (forced: one-rank RCCL rehearsal)
            return 1.0
        per = max(1, bucket_bytes // 4)
        host = self.grad.is_cuda and dist.get_backend() == "gloo"        # CPU rehearsal backend: stage each bucket on the host
        for o in range(0, self.grad.numel(), per):
            bucket = self.grad[o:o + per]
            if host:
                h = bucket.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                bucket.copy_(h)
            else:
                dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        return 1.0 / dist.get_world_size()

    def step(self, grad_scale: float = 1.0):
        self.steps += 1
        ops.adamw_step(self.param, self.grad, self.exp_avg, self.exp_avg_sq, self.steps, self.lr, self.betas, self.eps,
                       self.weight_decay, grad_scale)
        bump_weights_epoch()              # weight-derived caches (models.modules.layers.Derived) must be rebuilt

    # ---- hipGraph replay: the launch is frozen at capture, so the step-dependent constants live in device memory ----
    def enable_device_hyper(self):
        self.hyper_dev = torch.zeros(8, device=self.param.device, dtype=torch.float32)

    def stage_hyper(self, grad_scale: float = 1.0, advance: bool = True):
        """Host side of a (captured) step: advance the step count and copy this step's constants to the device buffer.
        advance=False stages the constants of the NEXT step without counting it (used while capturing: nothing executes)."""
        if advance:
            self.steps += 1
        self._hyper_host = ops.adamw_hyper(self.steps if advance else self.steps + 1, self.lr, self.betas, self.eps,
                                           self.weight_decay, grad_scale)
        self.hyper_dev.copy_(self._hyper_host, non_blocking=True)

    def step_dev(self):
        """Device side: the update with constants from `hyper_dev` (what gets captured)."""
        ops.adamw_step_dev(self.param, self.grad, self.exp_avg, self.exp_avg_sq, self.hyper_dev)

    def scheduler_step(self, iter_max: int, power: float = 0.9, min_lr: float = 1e-5):
        self.sched_it += 1
        self.lr = polynomial_lr(self.base_lr, self.lr, self.sched_it, iter_max, power, min_lr)
        return self.lr


def build_optimizers(encoder, decoder, lr_cnn: float, lr: float, lr_cva: Optional[float] = None, weight_decay: float = 1e-2,
                     weight_decay_cnn: float = 1e-2) -> Dict[str, FlatAdamW]:
    """train.py:211-213: cva / encoder / decoder optimizers (cva omitted when the encoder has no such parameters)."""
    g = split_param_groups(encoder, decoder)
    opts = {"enc": FlatAdamW(g["enc"], lr_cnn, weight_decay_cnn), "dec": FlatAdamW(g["dec"], lr, weight_decay)}
    if g["cva"]:
        opts["cva"] = FlatAdamW(g["cva"], lr_cva if lr_cva is not None else lr_cnn, weight_decay)
    return opts


class GraphedTrainStep:
    """One training step (taped forward, mask loss, backward, AdamW on every group, gradient reset) captured into a hipGraph
    and replayed: at config 5's micro-batch the eager step is bound by ~10^4 host-side launches, not by the GPU.
    `forward_fn(x) -> logits` must be built from mumpy_hip.autograd functions (capture-safe: no host synchronisation).
    Train mode works: the stochastic-depth masks are drawn by torch's graph-safe Philox generator inside the capture, so every
    replay draws new ones (test_hip_graphed_train_step_draws_fresh_drop_path_masks).  Learning-rate schedules keep working: the AdamW constants
    are staged into device memory before each replay.  Call `step(x, target)` -> loss3 (device tensor [total, iou, focal])."""

    def __init__(self, forward_fn, optimizers, x, target, warmup: int = 3, loss_scale: float = 1.0, all_reduce: bool = False):
        """all_reduce=True (data-parallel ranks): TWO graphs -- forward + loss + backward, and AdamW + gradient reset -- with
        the bucketed gradient all-reduce (RCCL) issued eagerly between their replays."""
        self.opts = list(optimizers.values()) if isinstance(optimizers, dict) else list(optimizers)
        self.x, self.target = x.clone(), target.clone()
        self.all_reduce = all_reduce
        for o in self.opts:
            o.enable_device_hyper()

        def fwd_bwd():
            logits = forward_fn(self.x)
            loss3, dlogits = ops.mask_loss(logits.detach(), self.target, loss_scale=loss_scale)
            logits.backward(dlogits)
            return loss3

        def update():
            for o in self.opts:
                o.step_dev()
                o.zero_grad()

        from .streams import new_distinct_stream
        side = new_distinct_stream(self.x.device, (torch.cuda.current_stream().cuda_stream,))
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        for _ in range(warmup):                          # warm-up steps are real steps (caches, allocator, lazy inits)
            with torch.cuda.stream(side):
                fwd_bwd()
            scales = [1.0] * len(self.opts)
            if all_reduce:
                # The collectives are issued from the CALLER's stream, never from the stream that captures: ProcessGroupNCCL's
                # watchdog thread polls the end event of every pending collective, and on ROCm 7.2 that query fails with
                # hipErrorCapturedEvent (process abort) when the stream the collective was issued from has meanwhile begun a
                # capture -- reproduced in isolation by tools/rccl_capture_probe.py (issued from the capture stream: abort;
                # from another stream, or with the watchdog drained first: fine).
                main.wait_stream(side)
                scales = [o.all_reduce_grads() for o in self.opts]
                side.wait_stream(main)
            with torch.cuda.stream(side):
                for o, sc in zip(self.opts, scales):
                    o.stage_hyper(sc)
                update()
        main.wait_stream(side)
        from .streams import quiesce_collectives
        quiesce_collectives()
        for o in self.opts:
            o.stage_hyper(advance=False)                 # capture records the launches; it does not run a step
        self.graph = torch.cuda.CUDAGraph()
        self.graph_update = None
        if all_reduce:
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):   # (same stream as the warm-up and thread-local capture errors: see GraphedForward._capture)
                self.loss3 = fwd_bwd()
            self.graph_update = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_update, pool=self.graph.pool(), stream=side, capture_error_mode="thread_local"):
                update()
        else:
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
                self.loss3 = fwd_bwd()
                update()
        bump_weights_epoch()

    def step(self, x=None, target=None, grad_scale: float = 1.0):
        if x is not None:
            self.x.copy_(x)
        if target is not None:
            self.target.copy_(target)
        if self.graph_update is None:
            for o in self.opts:
                o.stage_hyper(grad_scale)
            self.graph.replay()
        else:
            self.graph.replay()
            for o in self.opts:
                o.stage_hyper(grad_scale * o.all_reduce_grads())     # the step's collectives, on the replay's stream
            self.graph_update.replay()
        bump_weights_epoch()
        return self.loss3
