"""SURVEY 8f-2 (config 5's training tail): mask loss fwd+bwd, PolynomialLR, parameter groups, fused AdamW, bucketed gradient
all-reduce.  Goldens in tests/golden/train_tail.npz come from the reference's own utils/loss.py and
utils/optimizer/scheduler.py (tests/golden/gen_train_goldens.py).  CPU tests pin the oracle and the host logic; the
`gpu` tests run the HIP kernels through the C ABI."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_input, rel_err
from oracle import mumpy_oracle as O
from weight_fill import seeded_randn

HAS_GPU = torch.cuda.is_available()
TAGS = ["toy", "odd", "full"]


def _loss_case(g, tag):
    ss = g[tag + "/seed_shape"]
    seed, b, hw = int(ss[0]), int(ss[1]), [int(v) for v in ss[2:]]
    z = seeded_randn(seed, b, *hw) * 2.0
    t = (seeded_randn(seed + 50, b, 1, hw[1] * hw[2]) > 0.8).float()
    return z, t


# ----------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("tag", TAGS)
def test_oracle_mask_loss_matches_reference(train_golden, tag):
    z, t = _loss_case(train_golden, tag)
    z.requires_grad_(True)
    tot, iou, foc = O.mask_loss(z, t)
    (tot / 2.0).backward()                                      # the golden used accumulation_steps = 2
    ref = train_golden[tag + "/loss3"]
    assert abs(float(tot.detach()) / 2.0 - ref[0]) < 2e-6 and abs(float(iou.detach()) - ref[1]) < 2e-6 and abs(float(foc.detach()) - ref[2]) < 2e-6
    assert rel_err(z.grad, train_golden[tag + "/dlogits"]) < 1e-5


@pytest.mark.parametrize("tag", ["sched_a", "sched_b"])
def test_polynomial_lr_matches_reference(train_golden, tag):
    from mumpy_hip.train import polynomial_lr
    base, iter_max = train_golden[tag + "/base_itermax"]
    ref = train_golden[tag + "/lrs"]
    lrs, lr = [base], base
    for it in range(1, len(ref)):
        lr = polynomial_lr(base, lr, it, int(iter_max))
        lrs.append(lr)
    assert np.allclose(lrs, ref, rtol=1e-12, atol=0)
    assert np.allclose(O.polynomial_lr_sequence(base, int(iter_max), len(ref) - 1), ref, rtol=1e-12, atol=0)


def test_param_groups_split_on_cva():
    """train.py:198-213: encoder parameters with "cva" in the name get their own optimizer."""
    from models.decoder.decoder import BaselineDecoder
    from mumpy_hip.train import split_param_groups

    class Enc(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.block = torch.nn.Linear(4, 4)
            self.cva = torch.nn.Linear(4, 2)
            self.frozen = torch.nn.Linear(2, 2)
            self.frozen.weight.requires_grad_(False)
    enc, dec = Enc(), BaselineDecoder(in_channels=32, features=[32] * 5)
    g = split_param_groups(enc, dec)
    assert [tuple(p.shape) for p in g["cva"]] == [(2, 4), (2,)]
    assert [tuple(p.shape) for p in g["enc"]] == [(4, 4), (4,), (2,)]            # frozen.weight is filtered (utils.py:258)
    assert len(g["dec"]) == len(list(dec.parameters()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import PKG  # noqa: F401
    from mumpy_hip import distributed as D
    from mumpy_hip.train import FlatAdamW
    D.init_process_group("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))    # odd sizes: exercises the 16-B slot padding
    opt = FlatAdamW(net.parameters(), lr=1e-3)
    for i, p in enumerate(net.parameters()):
        p.grad.copy_(torch.full_like(p, float((rank + 1) * (i + 1))))
    scale = opt.all_reduce_grads(bucket_bytes=64)                                # 16-float buckets: several collectives
    q.put((rank, scale, [float(p.grad.flatten()[0]) for p in net.parameters()],
           bool(all(p.grad.data_ptr() >= opt.grad.data_ptr() for p in net.parameters()))))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_allreduce_two_ranks():
    """world 2, gloo: the flat gradient buffer is summed bucket by bucket; AdamW then applies 1/world."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, scale, firsts, views in res:
        assert scale == 0.5 and views
        assert firsts == [3.0 * (i + 1) for i in range(4)]                        # (1 + 2) * (i + 1)


# ----------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_mask_loss_matches_reference(train_golden, tag):
    from mumpy_hip import ops
    z, t = _loss_case(train_golden, tag)
    loss3, dz = ops.mask_loss(z.cuda(), t.cuda(), loss_scale=0.5)
    ref = train_golden[tag + "/loss3"]
    assert np.allclose(loss3.cpu().numpy(), ref, rtol=2e-5, atol=0)              # fp32 sums of 1e3..1e5 terms
    assert dz.shape == z.shape
    assert rel_err(dz.cpu(), train_golden[tag + "/dlogits"]) < 2e-5
    loss_only, none = ops.mask_loss(z.cuda(), t.cuda(), need_grad=False, loss_scale=0.5)
    assert none is None and torch.equal(loss_only, loss3)                         # deterministic, same reduction order
    again, dz2 = ops.mask_loss(z.cuda(), t.cuda(), loss_scale=0.5)
    assert torch.equal(dz2, dz) and torch.equal(again, loss3)


@pytest.mark.gpu
def test_hip_mask_loss_extremes():
    """saturated logits (|z| = 40) and an all-background target: finite loss and gradient, equal to the oracle."""
    from mumpy_hip import ops
    z = seeded_randn(9, 2, 1, 32, 32) * 40.0
    t = torch.zeros(2, 1, 1024)
    t[1, 0, :100] = 1.0
    zo = z.clone().requires_grad_(True)
    tot, iou, foc = O.mask_loss(zo, t)
    tot.backward()
    loss3, dz = ops.mask_loss(z.cuda(), t.cuda())
    assert torch.isfinite(loss3).all() and torch.isfinite(dz).all()
    assert np.allclose(loss3.cpu().numpy(), [float(tot.detach()), float(iou.detach()), float(foc.detach())], rtol=3e-5)
    assert rel_err(dz.cpu(), zo.grad) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 4, 1023, 262147])
def test_hip_adamw_matches_torch(n):
    """torch.optim.AdamW (what utils/utils.py:258 builds) for 5 steps, weight decay and a gradient scale included."""
    from mumpy_hip import ops
    p0 = seeded_randn(21, n)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, weight_decay=1e-2)
    p, m, v = p0.cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 6):
        g = seeded_randn(100 + step, n)
        ref.grad = g * 0.5
        opt.step()
        ops.adamw_step(p, g.cuda(), m, v, step, lr=3e-3, weight_decay=1e-2, grad_scale=0.5)
    assert rel_err(p.cpu(), ref.data) < 2e-6
    st = opt.state[ref]
    assert rel_err(m.cpu(), st["exp_avg"]) < 2e-6 and rel_err(v.cpu(), st["exp_avg_sq"]) < 2e-6


@pytest.mark.gpu
def test_flat_adamw_trains_like_torch_adamw():
    """FlatAdamW over a module (params re-pointed at the flat buffer) == per-tensor torch.optim.AdamW, with the
    polynomial schedule stepping both."""
    from mumpy_hip.train import FlatAdamW, polynomial_lr
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).cuda()
    twin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3)).cuda()
    twin.load_state_dict(net.state_dict())
    ref = torch.optim.AdamW(twin.parameters(), lr=1e-2, weight_decay=1e-4)
    opt = FlatAdamW(net.parameters(), lr=1e-2, weight_decay=1e-4)
    x = seeded_randn(5, 16, 7).cuda()
    for it in range(1, 9):
        for model in (net, twin):
            model(x).square().mean().backward()                  # autograd writes into the flat gradient views
        assert all(p.grad.data_ptr() >= opt.grad.data_ptr() for p in net.parameters())
        opt.step()
        ref.step()
        opt.zero_grad()
        ref.zero_grad()
        lr = opt.scheduler_step(iter_max=6)
        for gparam in ref.param_groups:
            gparam["lr"] = polynomial_lr(1e-2, gparam["lr"], it, 6)
        assert lr == ref.param_groups[0]["lr"]
    for a, b in zip(net.parameters(), twin.parameters()):
        assert rel_err(a.detach().cpu(), b.detach().cpu()) < 1e-5


def test_flat_adamw_state_dict_is_torch_adamw_layout(tmp_path):
    """CPU: FlatAdamW.state_dict() has torch.optim.AdamW's layout (the reference's enc_opt / dec_opt files,
    utils/utils.py:266-271): torch's own AdamW loads it, and a file torch wrote loads back -- through the weights-only
    loader of mumpy_hip.checkpoint."""
    from mumpy_hip import checkpoint as C
    from mumpy_hip.train import FlatAdamW
    torch.manual_seed(4)
    net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2))
    opt = FlatAdamW(net.parameters(), lr=2e-3, weight_decay=1e-3)
    opt.exp_avg.normal_(); opt.exp_avg_sq.uniform_(); opt.steps, opt.sched_it, opt.lr = 11, 9, 1.5e-3
    enc = torch.nn.Linear(2, 2)
    C.save_checkpoint(str(tmp_path), enc, enc, epoch=3, optimizers={"enc": opt, "dec": opt, "cva": opt})
    for f in ("enc_opt_3.pt", "dec_opt_3.pt", "cva_opt_3.pt", "encoder_3.pt", "decoder_3.pt"):
        assert (tmp_path / f).exists()
    sds = C.load_optimizer_states(str(tmp_path), epoch=3)
    assert set(sds) == {"enc", "dec", "cva"}
    twin = torch.optim.AdamW(net.parameters(), lr=1.0)
    twin.load_state_dict({k: v for k, v in sds["enc"].items() if k != "mumpy"})             # torch accepts it as its own
    assert twin.param_groups[0]["lr"] == 1.5e-3 and twin.param_groups[0]["weight_decay"] == 1e-3
    for i, p in enumerate(net.parameters()):
        o = opt.offsets[i]
        assert torch.equal(twin.state[p]["exp_avg"].reshape(-1), opt.exp_avg[o:o + p.numel()])
        assert float(twin.state[p]["step"]) == 11.0
    fresh = FlatAdamW(torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Linear(4, 2)).parameters(), lr=9.0)
    fresh.load_state_dict(twin.state_dict())                                                # and back, from torch's own dict
    for i in range(len(opt.params)):                 # (the flat buffers' alignment padding between parameters is not state)
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(fresh.state_dict()["state"][i][k], opt.state_dict()["state"][i][k])
    assert (fresh.steps, fresh.lr, fresh.weight_decay) == (11, 1.5e-3, 1e-3)
    fresh.load_state_dict(sds["enc"])
    assert (fresh.steps, fresh.sched_it, fresh.base_lr) == (11, 9, 2e-3)


def test_flat_adamw_conv_weights_are_channels_last_views():
    """CPU: a spatial conv weight adopted by FlatAdamW keeps its logical (Cout,Cin,kh,kw) shape and values, but its memory (and
    its gradient's and the Adam moments') is the (Cout,kh,kw,Cin) image the HIP convolution reads; the optimizer state still
    round-trips through torch.optim.AdamW in logical order."""
    from mumpy_hip.train import FlatAdamW
    torch.manual_seed(6)
    net = torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.Conv2d(6, 2, 1), torch.nn.Conv2d(1, 3, 3))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt = FlatAdamW(net.parameters(), lr=1e-3)
    w = net[0].weight
    assert w.shape == (6, 4, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous() and w.grad.permute(0, 2, 3, 1).is_contiguous()
    assert w.data_ptr() == opt.param.data_ptr() and net[1].weight.is_contiguous() and net[2].weight.is_contiguous()
    assert all(torch.equal(v, before[k]) for k, v in net.state_dict().items())
    assert torch.equal(opt.param[:w.numel()].view(6, 3, 3, 4), before["0.weight"].permute(0, 2, 3, 1))
    net[0](torch.randn(2, 4, 5, 5)).square().sum().backward()    # autograd accumulates into the permuted views as well
    assert opt.grad[:w.numel()].abs().sum() > 0
    opt.exp_avg.normal_(); opt.exp_avg_sq.uniform_()
    sd = opt.state_dict()
    assert sd["state"][0]["exp_avg"].shape == (6, 4, 3, 3) and sd["state"][0]["exp_avg"].is_contiguous()
    assert torch.equal(sd["state"][0]["exp_avg"].permute(0, 2, 3, 1).reshape(-1), opt.exp_avg[:w.numel()])
    twin = torch.optim.AdamW(net.parameters(), lr=1.0)
    twin.load_state_dict({k: v for k, v in sd.items() if k != "mumpy"})
    fresh = FlatAdamW(torch.nn.Sequential(torch.nn.Conv2d(4, 6, 3, padding=1), torch.nn.Conv2d(6, 2, 1), torch.nn.Conv2d(1, 3, 3)).parameters(), lr=1.0)
    fresh.load_state_dict(twin.state_dict())
    for i, (p, o) in enumerate(zip(opt.params, opt.offsets)):
        assert torch.equal(fresh.exp_avg[o:o + p.numel()], opt.exp_avg[o:o + p.numel()])
        assert torch.equal(fresh.exp_avg_sq[o:o + p.numel()], opt.exp_avg_sq[o:o + p.numel()])


@pytest.mark.gpu
def test_checkpoint_resume_continues_the_same_trajectory(tmp_path):
    """save -> load -> continue == never having stopped: parameters, both moments, step and scheduler counters survive
    (train.py:179-188 resumes encoder, decoder and both optimizers)."""
    from mumpy_hip import checkpoint as C
    from mumpy_hip.train import FlatAdamW

    def make():
        torch.manual_seed(8)
        return torch.nn.Sequential(torch.nn.Linear(9, 8), torch.nn.GELU(), torch.nn.Linear(8, 4)).cuda()

    x = seeded_randn(6, 32, 9).cuda()

    def run(net, opt, n):
        for _ in range(n):
            net(x).square().mean().backward()
            opt.step()
            opt.zero_grad()
            opt.scheduler_step(iter_max=20)

    a = make(); oa = FlatAdamW(a.parameters(), lr=1e-2, weight_decay=1e-3)
    run(a, oa, 10)                                                        # the uninterrupted run
    b = make(); ob = FlatAdamW(b.parameters(), lr=1e-2, weight_decay=1e-3)
    run(b, ob, 4)
    C.save_checkpoint(str(tmp_path), b, b, epoch=0, optimizers={"enc": ob})
    c = make(); oc = FlatAdamW(c.parameters(), lr=123.0)                  # a fresh process: wrong rate, zero moments
    e, _, _ = C.load_checkpoint(str(tmp_path), epoch=0)
    c.load_state_dict(e, strict=True)
    oc.load_state_dict(C.load_optimizer_states(str(tmp_path), epoch=0)["enc"])
    run(c, oc, 6)
    assert torch.equal(oc.param, oa.param) and torch.equal(oc.exp_avg, oa.exp_avg) and torch.equal(oc.exp_avg_sq, oa.exp_avg_sq)
    assert (oc.steps, oc.sched_it, oc.lr) == (oa.steps, oa.sched_it, oa.lr)


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from conftest import PKG  # noqa: F401
    from weight_fill import fill_module_
    from models.modules.swinTransformer import SwinTransformerBlock
    from mumpy_hip import distributed as D
    from mumpy_hip.autograd import swin_block_train
    from mumpy_hip.train import FlatAdamW
    D.init_process_group("gloo")
    dev = torch.device("cuda:0")
    blk = fill_module_(SwinTransformerBlock(dim=96, input_resolution=(14, 14), num_heads=3, window_size=7, shift_size=3)).to(dev)
    opt = FlatAdamW(blk.parameters(), lr=1e-3, weight_decay=1e-4)
    x = seeded_randn(300 + rank, 2, 196, 96).to(dev)               # this rank's micro-batch
    g = seeded_randn(310 + rank, 2, 196, 96).to(dev)
    swin_block_train(blk, x).backward(g)
    scale = opt.all_reduce_grads(bucket_bytes=1 << 16)               # several buckets
    opt.step(grad_scale=scale)
    q.put((rank, opt.param.cpu().numpy()))            # numpy: a torch tensor in the queue is shared by fd and can outlive its owner
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_training_step_matches_accumulated_single_process():
    """Two ranks (sharing the one GPU of the test box, gloo standing in for RCCL), one Swin block, one step: after the bucketed
    gradient all-reduce and the fused AdamW both ranks hold the same parameters, equal to one process that accumulated the
    two micro-batches' gradients and stepped with grad_scale = 1/2."""
    from weight_fill import fill_module_
    from models.modules.swinTransformer import SwinTransformerBlock
    from mumpy_hip.autograd import swin_block_train
    from mumpy_hip.train import FlatAdamW
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: torch.from_numpy(a) for r, a in (q.get(timeout=300) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])
    dev = torch.device("cuda:0")
    blk = fill_module_(SwinTransformerBlock(dim=96, input_resolution=(14, 14), num_heads=3, window_size=7, shift_size=3)).to(dev)
    opt = FlatAdamW(blk.parameters(), lr=1e-3, weight_decay=1e-4)
    for rank in range(2):                                            # autograd accumulates into the flat gradient views
        swin_block_train(blk, seeded_randn(300 + rank, 2, 196, 96).to(dev)).backward(seeded_randn(310 + rank, 2, 196, 96).to(dev))
    opt.step(grad_scale=0.5)
    assert rel_err(res[0], opt.param.cpu()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("b,cin,cout,kh,kw,h,w", [(2, 64, 32, 3, 3, 9, 11), (2, 128, 128, 3, 3, 28, 28), (1, 32, 64, 7, 1, 14, 14),
                                                  (3, 96, 32, 1, 7, 7, 5), (2, 256, 128, 3, 3, 56, 56), (2, 32, 128, 3, 3, 112, 112)])
def test_hip_conv2d_wgrad_one_launch(b, cin, cout, kh, kw, h, w):
    """mumpy_conv2d_wgrad_nhwc (all taps in one launch, borders and shifts by address arithmetic, split pixel ranges reduced in
    a fixed order) against torch autograd of F.conv2d in float64; accumulate mode adds into an existing buffer."""
    from mumpy_hip import ops
    x, dy = seeded_randn(1, b, cin, h, w), seeded_randn(2, b, cout, h, w)
    wt = torch.zeros(cout, cin, kh, kw, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, None, padding=(kh // 2, kw // 2)).backward(dy.double())
    ref = wt.grad.permute(0, 2, 3, 1)                                   # (Cout, kh, kw, Cin)
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    dw = ops.conv2d_wgrad(xd, dyd, kh, kw)
    assert rel_err(dw.cpu(), ref) < 1e-5
    assert torch.equal(dw, ops.conv2d_wgrad(xd, dyd, kh, kw))
    acc = torch.full((cout, kh, kw, cin), 0.25, device="cuda")
    assert ops.conv2d_wgrad(xd, dyd, kh, kw, dw_out=acc) is None
    assert rel_err(acc.cpu() - 0.25, ref) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("two_graphs", [False, True])
def test_graphed_train_step_equals_eager_steps(two_graphs):
    """GraphedTrainStep (hipGraph replay of forward + loss + backward + AdamW, step constants staged in device memory) walks the
    same parameter trajectory as eager steps, learning-rate changes included.  two_graphs: the data-parallel form -- backward
    and update captured separately, the gradient all-reduce (a no-op at world size 1) issued between the replays."""
    from weight_fill import fill_module_
    from models.decoder.decoder import BaselineDecoder
    from mumpy_hip import ops
    from mumpy_hip.autograd import baseline_decoder_train
    from mumpy_hip.train import FlatAdamW, GraphedTrainStep
    dev = torch.device("cuda:0")
    x = seeded_randn(400, 2, 64, 7, 7).to(dev)
    target = (seeded_randn(401, 2, 1, 224, 224) > 1.0).float().to(dev)
    lrs = [1e-3, 1e-3, 1e-3, 5e-4, 2.5e-4, 1e-4]

    def make():
        dec = fill_module_(BaselineDecoder(in_channels=64, features=[128] * 5)).eval().to(dev)
        return dec, FlatAdamW(dec.parameters(), lr=lrs[0], weight_decay=1e-4)
    dec_e, opt_e = make()
    for lr in lrs:                                                        # eager reference trajectory
        opt_e.lr = lr
        logits = baseline_decoder_train(dec_e, x)
        loss3, dl = ops.mask_loss(logits.detach(), target)
        logits.backward(dl)
        opt_e.step()
        opt_e.zero_grad()
    dec_g, opt_g = make()
    gs = GraphedTrainStep(lambda xx: baseline_decoder_train(dec_g, xx), [opt_g], x, target, warmup=3,
                          all_reduce=two_graphs)                             # 3 real steps at lrs[0..2]
    for lr in lrs[3:]:
        opt_g.lr = lr
        gs.step()
    torch.cuda.synchronize()
    assert opt_g.steps == opt_e.steps == len(lrs)
    assert rel_err(opt_g.param.cpu(), opt_e.param.cpu()) < 1e-5
    assert rel_err(opt_g.exp_avg_sq.cpu(), opt_e.exp_avg_sq.cpu()) < 1e-4
