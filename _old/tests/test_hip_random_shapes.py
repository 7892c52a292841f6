"""Seeded random-shape sweeps of the two kernels with the most intricate indexing (GEMM tiles / split-K / strided rows; window
attention forward + backward with arbitrary grids and shifts) against plain torch / the oracle.  GPU only."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import mumpy_oracle as O
from weight_fill import seeded_randn

pytestmark = pytest.mark.gpu


def _linear_cases(n, seed):
    rng = random.Random(seed)
    for i in range(n):
        m = rng.choice([1, 7, 33, 64, 100, 129, 392, 1000, 1961, 5000, 12345])
        nn_ = 32 * rng.randint(1, 40)
        k = 32 * rng.randint(1, 48)
        yield i, m, nn_, k, rng.random() < 0.5, rng.random() < 0.5, rng.random() < 0.7


@pytest.mark.parametrize("case", list(_linear_cases(24, 2024)), ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}")
def test_linear_random_shapes(case):
    from mumpy_hip import ops
    i, m, n, k, gelu, res, bias = case
    x, w = seeded_randn(1000 + i, m, k), seeded_randn(2000 + i, n, k) / k ** 0.5
    b = seeded_randn(3000 + i, n) if bias else None
    r = seeded_randn(4000 + i, m, n) if res else None
    ref = F.linear(x.double(), w.double(), None if b is None else b.double())
    if gelu:
        ref = F.gelu(ref)
    if r is not None:
        ref = ref + r.double()
    y = ops.linear(x.cuda(), w.cuda(), None if b is None else b.cuda(), act=ops.ACT_GELU if gelu else ops.ACT_NONE,
                   residual=None if r is None else r.cuda())
    assert rel_err(y.cpu().double(), ref) < 5e-6


@pytest.mark.parametrize("case", list(_linear_cases(24, 4048)), ids=lambda c: f"{c[1]}x{c[2]}x{c[3]}")
def test_linear_bf16x3_random_shapes(case):
    """The same sweep in split-precision mode (both tiles, every K-split plan, ragged M, epilogues): fp32-level error."""
    from mumpy_hip import ops
    i, m, n, k, gelu, res, bias = case
    x, w = seeded_randn(1100 + i, m, k), seeded_randn(2100 + i, n, k) / k ** 0.5
    b = seeded_randn(3100 + i, n) if bias else None
    r = seeded_randn(4100 + i, m, n) if res else None
    ref = F.linear(x.double(), w.double(), None if b is None else b.double())
    if gelu:
        ref = F.gelu(ref)
    if r is not None:
        ref = ref + r.double()
    ops.set_matrix_math("bf16x3")
    try:
        y = ops.linear(x.cuda(), w.cuda(), None if b is None else b.cuda(), act=ops.ACT_GELU if gelu else ops.ACT_NONE,
                       residual=None if r is None else r.cuda())
    finally:
        ops.set_matrix_math("fp32")
    assert rel_err(y.cpu().double(), ref) < 5e-6


def _wa_cases(n, seed):
    rng = random.Random(seed)
    for i in range(n):
        yield i, rng.randint(1, 3), 7 * rng.randint(1, 5), 7 * rng.randint(1, 4), 32 * rng.randint(1, 6), rng.randint(0, 6)


@pytest.mark.parametrize("case", list(_wa_cases(16, 7)), ids=lambda c: f"b{c[1]}_{c[2]}x{c[3]}_c{c[4]}_s{c[5]}")
def test_window_attention_fwd_bwd_random_grids(case):
    """any (B, Hs, W) grid of 7x7 windows, any cyclic shift 0..6 (with the matching region mask), 1..6 heads."""
    from models.modules.swinTransformer import relative_position_index
    from mumpy_hip import ops
    from mumpy_hip.autograd import WindowAttentionFn
    i, b, hs, w, c, shift = case
    if min(hs, w) <= 7:
        shift = 0
    qkv, table, dout = seeded_randn(5000 + i, b, hs * w, 3 * c), seeded_randn(6000 + i, 169, c // 32) * 0.3, seeded_randn(7000 + i, b, hs * w, c)
    idx = relative_position_index(7, 7)
    mask = O.shift_attn_mask(hs, w, shift) if shift else None
    qr, tr = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
    out = O.window_attention_core(qr, tr, idx, hs, w, shift, mask)
    out.backward(dout)
    qg, tg = qkv.cuda().requires_grad_(True), table.cuda().requires_grad_(True)
    tab, ids = ops.compact_attn_mask(mask.cuda()) if mask is not None else (None, None)
    y = WindowAttentionFn.apply(qg, tg, idx.cuda(), (b, hs, w, c, shift, 32 ** -0.5), tab, ids)
    y.backward(dout.cuda())
    assert rel_err(y.detach().cpu(), out.detach()) < 1e-5
    assert rel_err(qg.grad.cpu(), qr.grad) < 3e-5 and rel_err(tg.grad.cpu(), tr.grad) < 3e-5
