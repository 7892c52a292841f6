"""torch.autograd glue for TRAINING the Swin block on the HIP kernels (first slice: rows 5-7 of 8a).

Every Function's forward and backward is a C-ABI kernel call; torch contributes the tape, the tensors and the shape
bookkeeping only.  Linear backward is one call, mumpy_linear_bwd (csrc/gemm_bwd.hip), on the row-major tensors as they are:
    dX = dY W,  dW (+)= dY^T X,  db (+)= column sums of dY
with dW / db accumulated straight into the parameter's `.grad` when that is a view of FlatAdamW's flat gradient buffer.
(`set_matrix_math("bf16")` runs the same call with bf16 operands on the bf16 MFMA; only the split-precision modes keep the
first version's route: the forward GEMM on transposed copies.)
`swin_block_train(block, x)` runs a `models.modules.swinTransformer.SwinTransformerBlock` through these Functions, with the
same maths as its inference forward (swin:259-307); in train mode stochastic depth draws a per-sample mask per branch.
"""
import os

import torch

from . import ops


def _grad_slot(p):
    """The buffer a parameter's gradient is accumulated into by the backward kernels themselves, or None.  Opt-in: only a
    parameter that FlatAdamW has adopted (it points `.grad` at a view of its flat gradient buffer and records that view as
    `p._mumpy_flat_grad`) and whose `.grad` is STILL that view qualifies -- autograd then receives None for it and launches no
    add.  Resolved at BACKWARD time (the Functions keep the parameter objects, not the slots): a `.grad` that was re-pointed
    or set to None between forward and backward and parameters with hooks take the ordinary route in which the kernel returns
    dW / db to autograd.  `torch.autograd.grad(inputs=[adopted_param])` cannot be told apart from `.backward()` inside a
    Function: it fails loudly ("appears to not have been used in the graph") unless run under `grad_slots(False)`."""
    if p is None or not p.is_leaf or not _SLOTS_ON[0]:
        return None
    g = p.grad
    if g is None or g is not getattr(p, "_mumpy_flat_grad", None):
        return None
    if getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
        return None
    dense = g.is_contiguous() or (g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last))
    if g.dtype == torch.float32 and g.is_cuda and dense and g.shape == p.shape and not g.requires_grad:
        return g
    return None


_SLOTS_ON = [True]


class grad_slots:
    """Context manager: `with grad_slots(False):` makes every backward return its parameter gradients to autograd (no in-kernel
    accumulation into FlatAdamW's buffer) -- for torch.autograd.grad(inputs=[param]) and gradient inspection."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _SLOTS_ON[0] = _SLOTS_ON[0], self.on
        return self

    def __exit__(self, *exc):
        _SLOTS_ON[0] = self.prev
        return False


class LinearFn(torch.autograd.Function):
    """y = x W^T + b [+ residual]: with a residual the add rides in the GEMM epilogue (as in the inference forward) and its
    gradient is dy itself."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual=None):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.params = (weight, bias)
        return ops.linear(x, weight, bias, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        wshape = weight.shape                                     # (N, K), or a 1x1-conv kernel (N, K, 1, 1): the same memory
        n, k = wshape[0], weight.numel() // wshape[0]
        weight = weight.reshape(n, k)
        dy2, x2 = dy.reshape(-1, n).contiguous(), x.reshape(-1, k)
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if ops.matrix_math() not in ("fp32", "bf16") or LEGACY_LINEAR_BWD:
            # split-precision modes: the forward GEMM on transposed copies (its operand modes apply to the backward products too)
            dx = ops.linear(dy2, ops.transpose(weight)).reshape(x.shape) if need_dx else None
            dw = ops.linear(ops.transpose(dy2, 32), ops.transpose(x2, 32)).reshape(wshape) if need_dw else None
            db = ops.col_sum(dy2) if need_db else None
            return dx, dw, db, (dy if ctx.has_res and ctx.needs_input_grad[3] else None)
        m = dy2.shape[0]
        wslot, bslot = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        # large token counts: dX on the forward GEMM (the wave-specialised kernel) against a transposed copy of W -- the
        # copy is weight-sized and the product runs at 100+ TFLOP/s; everything else in one call, no copies
        big = need_dx and m >= BIG_DGRAD_ROWS
        dx = dw = db = None
        if (need_dx and not big) or need_dw or need_db:          # (a frozen Linear with a big input needs only the GEMM below)
            dx, dw, db = ops.linear_bwd(x2.contiguous(), weight, dy2, need_dx=need_dx and not big, need_dw=need_dw, need_db=need_db,
                                        dw_out=wslot.reshape(n, k) if (need_dw and wslot is not None) else None,
                                        db_out=bslot if need_db else None)
        if big:
            dx = ops.linear(dy2, ops.transpose(weight))
        if dw is not None:
            dw = dw.reshape(wshape)
        return (dx.reshape(x.shape) if need_dx else None), dw, db, (dy if ctx.has_res and ctx.needs_input_grad[3] else None)


LEGACY_LINEAR_BWD = os.environ.get("MUMPY_LEGACY_LINEAR_BWD", "0") != "0"   # the first version's route (transposes + forward GEMM), for A/B runs
BIG_DGRAD_ROWS = int(os.environ.get("MUMPY_BIG_DGRAD_ROWS", "16384"))     # tools/big_dgrad_sweep.sh: at B = 2 the step is flat in this (4096 .. never)


def _residual_linear(x_res, drop_path, inp, lin_w, lin_b):
    """x_res + drop_path(Linear(inp)): one launch when stochastic depth is inactive (eval mode / rate 0), else the taped
    Linear, the per-sample scale and the add."""
    if not drop_path.training or getattr(drop_path, "drop_prob", 0.0) <= 0.0:
        return LinearFn.apply(inp, lin_w, lin_b, x_res)
    return AddFn.apply(x_res, drop_path_train(drop_path, LinearFn.apply(inp, lin_w, lin_b)))


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.params = (gamma, beta)
        return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = _ln_backward(ctx, x, gamma, dy, None)
        return dx, dg, db, None


def _ln_backward(ctx, x, gamma, dy, dx_add):
    gslot, bslot = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
    both = gslot is not None and bslot is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
    return ops.layernorm_bwd(x, gamma, dy.contiguous(), ctx.eps, dx_add=None if dx_add is None else dx_add.contiguous(),
                             dg_out=gslot if both else None, db_out=bslot if both else None)


class ResidualLayerNormFn(torch.autograd.Function):
    """x -> (x, LayerNorm(x)) for the pre-norm residual pattern `x + f(norm(x))` (swin:302-305, blocks:86-92): the gradient that
    arrives over the residual output is added to the LayerNorm's dx inside the backward kernel (no separate add launch)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.params = (gamma, beta)
        return x.view_as(x), ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dx, dg, db = _ln_backward(ctx, x, gamma, dy, dres)
        return dx, dg, db, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x, dy.contiguous())


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class DropPathFn(torch.autograd.Function):
    """x[b] * scale[b] with scale = Bernoulli(keep)/keep per sample (timm's drop_path, swin:302,305)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return ops.scale_samples(x, scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        return ops.scale_samples(dy.contiguous(), scale), None


def drop_path_train(module, x):
    """The block's `drop_path` in training: identity for nn.Identity / eval mode / rate 0, else a fresh per-sample mask drawn
    with torch's generator (as timm does: `x.new_empty(B).bernoulli_(keep)`), scaled by 1/keep."""
    p = getattr(module, "drop_prob", 0.0)
    if not module.training or p <= 0.0:
        return x
    keep = 1.0 - p
    scale = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
    if keep > 0.0:
        scale.div_(keep)
    return DropPathFn.apply(x, scale)


class WindowAttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale + bias + mask) v on raster-ordered qkv; differentiable in qkv and the bias table."""

    @staticmethod
    def forward(ctx, qkv, table, rel_index, dims, mask_tab, mask_id):
        b, hs, w, c, shift, scale = dims
        idx32 = ops.rel_index32(rel_index)
        bias_pad = ops.expand_relpos_bias(table.detach(), idx32)
        ctx.save_for_backward(qkv, bias_pad, idx32, ops.rel_index_csr(rel_index))      # (both index images are cached on the buffer)
        ctx.dims, ctx.mask = dims, (mask_tab, mask_id)
        ctx.table = table
        return ops.window_attention(qkv, bias_pad, b, hs, w, c, shift, scale, mask_tab, mask_id)

    @staticmethod
    def backward(ctx, dout):
        qkv, bias_pad, idx32, csr = ctx.saved_tensors
        b, hs, w, c, shift, scale = ctx.dims
        dqkv, dtable = ops.window_attention_bwd(qkv, dout.contiguous(), bias_pad, idx32, b, hs, w, c, shift, scale, *ctx.mask,
                                                dtable_out=_grad_slot(ctx.table) if ctx.needs_input_grad[1] else None, rel_csr=csr)
        return dqkv, dtable, None, None, None, None


def swin_block_train(block, x):
    """SwinTransformerBlock.forward (swin:259-307) with a backward: x (B, L, C) -> (B, L, C), gradients reach x and every
    parameter of the block (norm1/2, qkv, relative_position_bias_table, proj, fc1, fc2)."""
    h, w = block.input_resolution
    b, l, c = x.shape
    hs = l // w
    att = block.attn
    tab, ids = att.mask_pack(block.attn_mask)
    x, y = ResidualLayerNormFn.apply(x, block.norm1.weight, block.norm1.bias, block.norm1.eps)
    qkv = LinearFn.apply(y, att.qkv.weight, att.qkv.bias)
    a = WindowAttentionFn.apply(qkv, att.relative_position_bias_table, att.relative_position_index,
                                (b, hs, w, block.dim, block.shift_size, att.scale), tab, ids)
    x = _residual_linear(x, block.drop_path, a, att.proj.weight, att.proj.bias)
    x, z = ResidualLayerNormFn.apply(x, block.norm2.weight, block.norm2.bias, block.norm2.eps)
    hmid = GeluFn.apply(LinearFn.apply(z, block.mlp.fc1.weight, block.mlp.fc1.bias))
    return _residual_linear(x, block.drop_path, hmid, block.mlp.fc2.weight, block.mlp.fc2.bias)


class PatchGatherFn(torch.autograd.Function):
    """x (B, H*W, C) -> (B, H/2 * W/2, 4C): the 2x2 gather of PatchMerging, a permutation in one launch each way."""
    @staticmethod
    def forward(ctx, x, h, w):
        b, l, c = x.shape
        if l != h * w or h % 2 or w % 2:
            raise RuntimeError(f"PatchMerging: {l} tokens are not an even {h} x {w} grid")
        ctx.geom = (b, h, w, c)
        return ops.patch_gather(x.contiguous(), *ctx.geom)

    @staticmethod
    def backward(ctx, dy):
        return ops.patch_gather(dy.contiguous(), *ctx.geom, inverse=True), None, None


def patch_merging_train(pm, x):
    """PatchMerging.forward (swin:344-367) with a backward: 2x2 gather in the order (0,0),(1,0),(0,1),(1,1) (swin:357-361,
    a pure permutation), LayerNorm(4C) and the bias-free reduction on the HIP kernels."""
    h, w = pm.input_resolution
    g = PatchGatherFn.apply(x, h, w)
    return LinearFn.apply(LayerNormFn.apply(g, pm.norm.weight, pm.norm.bias, pm.norm.eps), pm.reduction.weight, None)


def baseline_tokenize_train(tok, x):
    """BaselineTokenize.forward (swin:11-32) with a backward: the Conv3d with kernel = stride = (T,4,4) is a per-patch Linear
    over (c, t, ky, kx); the patch gather is a reshape/permute, the product and the LayerNorm run on the HIP kernels.
    K = 3*T*16 is zero-padded to a multiple of 32 for the GEMM (the pad columns carry no gradient)."""
    w = tok.proj.weight                                        # (Cout, 3, T, 4, 4)
    b, t, ch, hh, ww = x.shape
    if t != w.shape[2]:
        raise RuntimeError("BaselineTokenize expects the tubelet to span the whole clip (squeeze(-3), swin:29)")
    p = w.shape[3]
    cols = x.permute(0, 2, 1, 3, 4).reshape(b, ch, t, hh // p, p, ww // p, p).permute(0, 3, 5, 1, 2, 4, 6)
    cols = cols.reshape(b * (hh // p) * (ww // p), ch * t * p * p)
    k = cols.shape[1]
    pad = (-k) % 32
    cols = torch.nn.functional.pad(cols, (0, pad)).contiguous()
    wmat = torch.nn.functional.pad(w.reshape(w.shape[0], k), (0, pad))
    y = LinearFn.apply(cols, wmat, tok.proj.bias)
    return LayerNormFn.apply(y, tok.norm.weight, tok.norm.bias, tok.norm.eps).reshape(b, -1, w.shape[0])


def baseline_encoder_train(enc, x):
    """BaselineEncoder.forward (encoder.py:22-30; SwinTransformer.forward_features swin:604-625) with a backward:
    x (B,3,3,224,224) -> (B,1024,7,7).  Stochastic depth is active when the module is in train mode."""
    m = enc.base
    y = baseline_tokenize_train(m.patch_embed, x)
    for layer in m.layers:
        for blk in layer.blocks:
            y = swin_block_train(blk, y)
        if layer.downsample is not None:
            y = patch_merging_train(layer.downsample, y)
    y = LayerNormFn.apply(y, m.norm.weight, m.norm.bias, m.norm.eps)
    b, _, c = y.shape
    return y.reshape(b, 7, 7, c).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------- BaselineDecoder (config 1)
class Conv2dFn(torch.autograd.Function):
    """3x3 / kxk same-padding convolution on NHWC (mumpy_conv2d_nhwc_fwd).  Backward:
       dX = conv(dY, W flipped and transposed)  (the same implicit-GEMM kernel),
       dW[tap] = dY^T X_shifted(tap)            (mumpy_conv2d_wgrad_nhwc: ONE launch over all taps on the NHWC tensors as
                                                 they are, the shift and the zero border are address arithmetic),
       db = column sums of dY."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        """weight: the nn.Conv2d parameter (Cout,Cin,kh,kw) as it is -- channels_last memory when FlatAdamW owns it, so that its
        (Cout,kh,kw,Cin) image is a view (no copy per step) and the weight gradient accumulates into the grad slot -- or any
        tensor of that logical shape."""
        w_krsc = weight.permute(0, 2, 3, 1)
        if not w_krsc.is_contiguous():
            w_krsc = w_krsc.contiguous()
        ctx.save_for_backward(x, w_krsc)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        return ops.conv2d_nhwc(x, w_krsc, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors                                   # x logical (B,Cin,H,W) NHWC; w (Cout,kh,kw,Cin)
        cout, kh, kw, cin = w.shape
        b, _, h, wd = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last)
        dy2 = dy.permute(0, 2, 3, 1).reshape(-1, cout)             # (P, Cout) view of the NHWC memory
        dx = dw = db = None
        fast = ops.matrix_math() in ("fp32", "bf16") and not LEGACY_LINEAR_BWD
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_nhwc(dy, ops.conv_weight_dgrad(w) if fast else w.permute(3, 1, 2, 0).flip(1, 2).contiguous())
        if ctx.needs_input_grad[1]:
            if fast:                                               # one launch over all taps, no shifted copies (fp32 or bf16 operands)
                wslot = _grad_slot(ctx.params[0])                  # channels_last 4-D slot: its KRSC image is a contiguous view
                if wslot is not None and wslot.dim() == 4 and wslot.permute(0, 2, 3, 1).is_contiguous():
                    ops.conv2d_wgrad(x, dy, kh, kw, dw_out=wslot.permute(0, 2, 3, 1))
                else:
                    dw = ops.conv2d_wgrad(x, dy, kh, kw).permute(0, 3, 1, 2)       # logical OIHW for autograd
            else:                                                  # split-precision modes: one forward GEMM per tap on copies
                dyt = ops.transpose(dy2.contiguous(), 32)          # (Cout, Ppad)
                xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, kw // 2, kw // 2, kh // 2, kh // 2))   # (B,H+2,W+2,Cin)
                taps = []
                for ky in range(kh):
                    for kx in range(kw):
                        xs = xp[:, ky:ky + h, kx:kx + wd, :].reshape(-1, cin).contiguous()
                        taps.append(ops.linear(dyt, ops.transpose(xs, 32)))                          # (Cout, Cin)
                dw = torch.stack(taps, dim=1).reshape(cout, kh, kw, cin).permute(0, 3, 1, 2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            bslot = _grad_slot(ctx.params[1]) if fast else None
            if bslot is not None:                                  # column sums of dY straight into the slot (no add launch)
                ops.linear_bwd(None, None, dy2.contiguous(), need_dx=False, need_dw=False, need_db=True, db_out=bslot)
            else:
                db = ops.col_sum(dy2.contiguous())
        return dx, dw, db


class GroupNormActFn(torch.autograd.Function):
    """GroupNorm followed by act in {0, ops.ACT_RELU, ops.ACT_SIGMOID} on NHWC."""

    @staticmethod
    def forward(ctx, z, gamma, beta, groups, eps, act):
        z, partial, nsplit = ops.gn_stats(z, groups)
        ctx.save_for_backward(z, partial, gamma, beta)
        ctx.cfg = (nsplit, groups, eps, act)
        ctx.params = (gamma, beta)
        return ops.gn_apply_resample(z, (partial, nsplit, gamma, beta, groups, eps), act=act)

    @staticmethod
    def backward(ctx, dy):
        z, partial, gamma, beta = ctx.saved_tensors
        nsplit, groups, eps, act = ctx.cfg
        gs, bs = _grad_slot(ctx.params[0]), _grad_slot(ctx.params[1])
        if gs is None or bs is None:
            gs = bs = None
        dz, dg, db = ops.gn_bwd(z, (partial, nsplit), gamma, beta, dy, groups, eps, act, dg_out=gs, db_out=bs)
        return dz, dg, db, None, None, None



class UpsampleFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=scale, mode="bilinear", align_corners=...) on NHWC, scale in {2, 4}."""

    @staticmethod
    def forward(ctx, x, scale, align_corners):
        ctx.cfg = (scale, align_corners)
        return ops.gn_apply_resample(x, None, scale=scale, align_corners=align_corners)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample_bwd(dy, *ctx.cfg), None, None



class FinalConvFn(torch.autograd.Function):
    """Conv2d(C -> 1, 3x3) (mumpy_final_conv_fwd).  Backward: mumpy_final_conv_bwd for C = 32 (the three-view Decoder); other widths
    go through the generic conv path with the single output channel embedded in a 32-channel gradient image (the implicit-GEMM
    kernel wants channel counts in multiples of 32) and one M = 1 GEMM per tap -- fine for a test, 1.6 ms per step at 224 x 224."""

    @staticmethod
    def forward(ctx, x, w_krsc, bias):
        ctx.save_for_backward(x, w_krsc)
        return ops.final_conv(x, w_krsc, bias)

    @staticmethod
    def backward(ctx, dy):                                         # dy (B,1,H,W)
        x, w = ctx.saved_tensors                                   # w (1,3,3,C)
        b, c, h, wd = x.shape
        if c == 32 and not LEGACY_LINEAR_BWD:
            return ops.final_conv_bwd(x, w, dy.contiguous())
        dyp = torch.zeros(b, h, wd, 32, device=dy.device, dtype=torch.float32)
        dyp[..., 0] = dy[:, 0]
        wt = torch.zeros(c, 3, 3, 32, device=dy.device, dtype=torch.float32)
        wt[..., 0] = w[0].permute(2, 0, 1).flip(1, 2)
        dx = ops.conv2d_nhwc(dyp.permute(0, 3, 1, 2), wt)
        dyt = ops.transpose(dy.reshape(-1, 1).contiguous(), 32)    # (1, Ppad)
        xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
        taps = [ops.linear(dyt, ops.transpose(xp[:, ky:ky + h, kx:kx + wd, :].reshape(-1, c).contiguous(), 32))
                for ky in range(3) for kx in range(3)]
        dw = torch.stack(taps, dim=1).reshape(1, 3, 3, c)
        return dx, dw, dy.sum().reshape(1)


def baseline_decoder_train(dec, x):
    """BaselineDecoder.forward (decoder.py:277-284) with a backward: x (B,in_channels,7,7) -> logits (B,1,224,224)."""
    x = x.contiguous(memory_format=torch.channels_last)
    for i in range(5):
        conv, gn = getattr(dec, f"decoder_{i + 1}")[0], getattr(dec, f"decoder_{i + 1}")[1]
        z = Conv2dFn.apply(x, conv.weight, conv.bias)
        a = GroupNormActFn.apply(z, gn.weight, gn.bias, gn.num_groups, gn.eps, ops.ACT_RELU)
        x = UpsampleFn.apply(a, 2, True)
    return FinalConvFn.apply(x, dec.final_out.weight.permute(0, 2, 3, 1).contiguous(), dec.final_out.bias)


# ---------------------------------------------------------------------------------------------- global temporal blocks (row 13)
class TemporalAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads, scale):
        s_, t, c3 = qkv.shape
        ctx.save_for_backward(qkv)
        ctx.cfg = (s_, t, c3 // 3, heads, scale)
        return ops.temporal_attention(qkv, s_, t, c3 // 3, heads, scale)

    @staticmethod
    def backward(ctx, dout):
        (qkv,) = ctx.saved_tensors
        return ops.temporal_attention_bwd(qkv, dout.contiguous(), *ctx.cfg), None, None


def global_block_train(block, x):
    """blocks.Block.forward (blocks:77-92) with a backward: x (S, T, C) sites x temporal tokens."""
    att = block.attn.unwrapped
    x, y = ResidualLayerNormFn.apply(x, block.norm1.weight, block.norm1.bias, block.norm1.eps)
    a = TemporalAttentionFn.apply(LinearFn.apply(y, att.qkv.weight, att.qkv.bias), att.heads, att.scale)
    x = _residual_linear(x, block.drop_path, a, att.proj.weight, att.proj.bias)
    x, z = ResidualLayerNormFn.apply(x, block.norm2.weight, block.norm2.bias, block.norm2.eps)
    mlp = block.mlp.unwrapped
    hmid = GeluFn.apply(LinearFn.apply(z, mlp.fc1.weight, mlp.fc1.bias))
    return _residual_linear(x, block.drop_path, hmid, mlp.fc2.weight, mlp.fc2.bias)


# ---------------------------------------------------------------------------------------------- pyramid Decoder (row 15)
def _conv_train(x, conv):
    """nn.Conv2d (stride 1, same padding) through Conv2dFn; Cin is zero-padded to a multiple of 32 (the 9-channel DCT input)."""
    w = conv.weight                                                       # (Cout, Cin, kh, kw)
    pad = (-w.shape[1]) % 32
    if pad:                                                               # (a derived tensor: its gradient goes through autograd)
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad))
        x = torch.cat([x, x.new_zeros(x.shape[0], pad, x.shape[2], x.shape[3])], dim=1)
    return Conv2dFn.apply(x.contiguous(memory_format=torch.channels_last), w, conv.bias)


def _gcm_train(m, x):
    return _conv_train(_conv_train(x, m.conv_l1), m.conv_l2) + _conv_train(_conv_train(x, m.conv_r1), m.conv_r2)


def decoder_train(dec, x, view_x, ffinfo):
    """Decoder.forward (decoder.py:183-225) with a backward: x (B,2304,7,7), view_x[4][3] of (B,1,L,C), ffinfo (B,9,224,224)
    -> (logits (B,1,224,224), feats (B,32,224,224)).  Convolutions, GroupNorm(+ReLU/Sigmoid), the bilinear resamplings and
    the temporal heads (Conv3d with kernel = stride = (T,1,1), i.e. a per-pixel Linear over (C,T)) run on the HIP kernels
    in both directions; the wiring in between (products, sums, concatenation, PixelShuffle, 2x2 average pooling) is left
    to torch's own autograd in this first version."""
    tdims = dec.input_token_temporal_dims
    tmax = max(tdims)
    rgb = []
    for s_ in range(4):
        seq = getattr(dec, f"rgb_decoder_{s_ + 1}")
        conv, gn = seq[0], seq[1]
        parts = []
        for v, t in enumerate(view_x[s_]):                                # merge_views_along_channel_axis (decoder.py:43-53)
            b, tt, n, c = t.shape
            tv = tdims[v]
            parts.append(t.reshape(b, tv, (tt * n) // tv, c).repeat(1, tmax // tv, 1, 1))
        m = torch.cat(parts, dim=-1)                                      # (B, T, n, C')
        b, t, n, c = m.shape
        cols = m.permute(0, 2, 3, 1).reshape(b * n, c * t).contiguous()   # per pixel: features ordered (C', T) like the weight
        y = LinearFn.apply(cols, conv.weight.reshape(conv.weight.shape[0], c * t), conv.bias)
        side = dec.shape[s_]
        y = y.reshape(b, side, side, -1).permute(0, 3, 1, 2)              # logical NCHW over NHWC memory
        rgb.append(GroupNormActFn.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, ops.ACT_RELU))
    rgb1, rgb2, rgb3, rgb4 = rgb
    freq, f = [], ffinfo
    for i in range(5):
        seq = getattr(dec, f"decoder_frequency_{i}")
        z = _conv_train(torch.nn.functional.avg_pool2d(f, 2), seq[1])
        f = GroupNormActFn.apply(z, seq[2].weight, seq[2].bias, seq[2].num_groups, seq[2].eps, ops.ACT_SIGMOID)
        freq.append(f)
    up = UpsampleFn.apply
    out1 = torch.nn.functional.pixel_shuffle(_gcm_train(dec.gcm1, torch.cat([rgb4, x], 1)) * freq[4], 2)
    gcn1 = _gcm_train(dec.gcm2, rgb3 * up(_conv_train(rgb4, dec.seb1.conv), 2, False))
    gcn2 = _gcm_train(dec.gcm3, rgb2 * up(_conv_train(torch.cat([rgb3, up(rgb4, 2, False)], 1), dec.seb2.conv), 2, False))
    gcn3 = _gcm_train(dec.gcm4, rgb1 * up(_conv_train(torch.cat([rgb2, up(rgb3, 2, False), up(rgb4, 4, False)], 1),
                                                      dec.seb3.conv), 2, False))

    def block(z, seq):
        a = GroupNormActFn.apply(_conv_train(z, seq[0]), seq[1].weight, seq[1].bias, seq[1].num_groups, seq[1].eps, ops.ACT_RELU)
        return up(a, 2, True)

    z = block(gcn1 * freq[3] + out1, dec.decoder_2)
    z = block(z + gcn2 * freq[2], dec.decoder_3)
    z = block(z + gcn3 * freq[1], dec.decoder_4)
    z = block(z * freq[0], dec.decoder_5)
    feats = torch.nn.functional.avg_pool2d(torch.nn.functional.pixel_shuffle(z, 2), 2)
    logits = FinalConvFn.apply(feats.contiguous(memory_format=torch.channels_last),
                               dec.final_out.weight.permute(0, 2, 3, 1).contiguous(), dec.final_out.bias)
    return logits, feats


# ---------------------------------------------------------------------------------------------- SwinDAttention (row 10)
class DWConv5Fn(torch.autograd.Function):
    """Depthwise 5x5 conv (padding 2) inside 7x7 windows: x (N,49,C) token-major, weight (C,1,5,5), bias (C)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w25 = weight.reshape(weight.shape[0], 25).contiguous()
        ctx.save_for_backward(x, w25)
        ctx.wshape = weight.shape
        return ops.dwconv5_window(x, w25, bias)

    @staticmethod
    def backward(ctx, du):
        x, w25 = ctx.saved_tensors
        dx, dw, db = ops.dwconv5_window_bwd(x, w25, du.contiguous())
        return dx, dw.reshape(ctx.wshape), db


class DeformSampleFn(torch.autograd.Function):
    """Bilinear sampling of kv windows at the learned positions (grid_sample semantics of deform:353-356), window form:
    x2w (B2,49,C), pos (nq,3,49,2) -> (B2,49,C); kv window b2 uses pos[b2 % nq]."""

    @staticmethod
    def forward(ctx, x2w, pos):
        ctx.save_for_backward(x2w, pos)
        b2, _, c = x2w.shape
        return ops.deform_sample(x2w.reshape(b2, 49, c), pos, b2, 7, 7, c, pos.shape[0])

    @staticmethod
    def backward(ctx, ds):
        x2w, pos = ctx.saved_tensors
        return ops.deform_sample_bwd(x2w, pos, ds.contiguous())


class DeformAttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale) v per kv window with the reference's pairing (q window = kv window mod B1, deform:330) and the sum
    over adjacent r-tuples (deform:394-395), window form: q (B1,49,C), kv (B1*r,49,2C) -> (B1,49,C)."""

    @staticmethod
    def forward(ctx, q, kv, scale):
        b1, _, c = q.shape
        r = kv.shape[0] // b1
        ctx.save_for_backward(q, kv)
        ctx.cfg = (r, scale)
        return ops.deform_attention(q, kv, ops.pad_mask(q.device), b1, 7, 7, c, r, scale)

    @staticmethod
    def backward(ctx, dout):
        q, kv = ctx.saved_tensors
        dq, dkv = ops.deform_attention_bwd(q, kv, dout.contiguous(), *ctx.cfg)
        return dq, dkv, None


_REF_POINTS = {}


def _ref_points(device):
    key = str(device)
    if key not in _REF_POINTS:
        r = (torch.linspace(0.5, 6.5, 7) / 7.0) * 2.0 - 1.0                # deform:313-319
        _REF_POINTS[key] = torch.stack(torch.meshgrid(r, r, indexing="ij"), -1).reshape(49, 2).to(device)
    return _REF_POINTS[key]


def swin_dattention_train(att, x1w, x2w):
    """SwinDAttention.forward (deform:324-405) with a backward, window form: x1w (B1,49,C) q windows, x2w (B2,49,C) kv windows
    (already through `pre`), B2 = r*B1 -> y (B1,49,C) including the un-permuted (C,49) -> (49,C) reshape of deform:403.
    The offset network runs unfused (depthwise conv, LayerNorm, GELU, 1x1 conv as kernels; tanh / scaling / reference points
    on the (B1,3,49,2) positions are a few KB of torch arithmetic)."""
    b1, _, c = x1w.shape
    g, cg = att.n_groups, att.n_group_channels
    q = LinearFn.apply(x1w, att.proj_q.weight, att.proj_q.bias)          # (C,C,1,1) leaf: dW accumulates in its grad slot
    off = att.conv_offset
    qg = q.reshape(b1, 49, g, cg).permute(0, 2, 1, 3).reshape(b1 * g, 49, cg).contiguous()
    u = DWConv5Fn.apply(qg, off[0].weight, off[0].bias)
    a = GeluFn.apply(LayerNormFn.apply(u, off[1].norm.weight, off[1].norm.bias, off[1].norm.eps))
    wpw = torch.nn.functional.pad(off[3].weight.reshape(2, cg), (0, 0, 0, 30))            # N = 2 padded to the GEMM's multiple of 32
    o2 = LinearFn.apply(a.reshape(-1, cg), wpw, None)[:, :2].reshape(b1, g, 49, 2)
    pos = torch.tanh(o2) * (2.0 / 7.0) + _ref_points(x1w.device)                           # deform:339-349, (y, x)
    samp = DeformSampleFn.apply(x2w.contiguous(), pos.contiguous())
    wkv = torch.cat([att.proj_k.weight.reshape(c, c), att.proj_v.weight.reshape(c, c)], 0)
    bkv = torch.cat([att.proj_k.bias, att.proj_v.bias])
    kv = LinearFn.apply(samp, wkv, bkv)
    o = DeformAttentionFn.apply(q, kv, att.scale)
    yt = LinearFn.apply(o, att.proj_out.weight, att.proj_out.bias)
    return yt.transpose(1, 2).reshape(b1, 49, c)


# ---------------------------------------------------------------------------------------------- the three-view encoder
def _tubelet_tokens(x, proj, norm):
    """One view of CrossThreeViewTokenize (mTVE:605-618): Conv3d(3 -> C, k = s = (t,4,4)) as a per-tubelet Linear + LayerNorm.
    x (B,T,3,H,W) -> (B, (T//t)*(H/4)*(W/4), C), frames stacked on the token axis."""
    w = proj.weight                                                       # (C, 3, t, 4, 4)
    cout, ch, t, p, _ = w.shape
    b, T, _, hh, ww = x.shape
    tt = (T - t) // t + 1
    cols = x[:, :tt * t].permute(0, 2, 1, 3, 4).reshape(b, ch, tt, t, hh // p, p, ww // p, p).permute(0, 2, 4, 6, 1, 3, 5, 7)
    cols = cols.reshape(b * tt * (hh // p) * (ww // p), ch * t * p * p)
    k = cols.shape[1]
    pad = (-k) % 32
    y = LinearFn.apply(torch.nn.functional.pad(cols, (0, pad)).contiguous(), torch.nn.functional.pad(w.reshape(cout, k), (0, pad)),
                       proj.bias)
    return LayerNormFn.apply(y, norm.weight, norm.bias, norm.eps).reshape(b, -1, cout)


def _windows(x, hs, w):
    """(B, hs*w, C) raster -> (B*nW, 49, C) window-major (window_partition, swin:54-66; a pure permutation)."""
    b, _, c = x.shape
    return x.view(b, hs // 7, 7, w // 7, 7, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, 49, c)


def cross_swin_block_train(blk, x1, x2):
    """CrossSwinBlock.forward (mTVE:228-291) with a backward -> (x1_new, out); `out` is the W-MSA output before the residual,
    which the next view's cross attention consumes (mTVE:275, 347-349).  x2 is ignored for the last view."""
    _, w = blk.input_resolution
    b, l1, c1 = x1.shape
    hs1 = l1 // w
    att = blk.attn
    x1, y = ResidualLayerNormFn.apply(x1, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
    a = WindowAttentionFn.apply(LinearFn.apply(y, att.qkv.weight, att.qkv.bias), att.relative_position_bias_table,
                                att.relative_position_index, (b, hs1, w, c1, 0, att.scale), None, None)
    out = LinearFn.apply(a, att.proj.weight, att.proj.bias)
    x1 = AddFn.apply(x1, drop_path_train(blk.drop_path, out))
    if not blk.last_view:
        hs2 = x2.shape[1] // w
        x1w = _windows(x1, hs1, w).contiguous()
        x2w = LinearFn.apply(_windows(x2, hs2, w).contiguous(), blk.pre.weight, blk.pre.bias)
        # stochastic depth applies twice on this branch, as in the reference: CVAModule drops D per WINDOW (its input's
        # leading axis is B*nW, mTVE:138), the block drops the window-major y per clip (mTVE:286); rates dpr[2,4,22] of
        # stages 1-3 are non-zero (mTVE:553)
        d = drop_path_train(blk.cva.drop_path, swin_dattention_train(blk.cva.crossattn, x1w, x2w))
        yw = AddFn.apply(x1w, d)                                          # CVAModule: x1 + drop_path(D) (mTVE:138)
        x1 = AddFn.apply(x1, drop_path_train(blk.drop_path, yw.reshape(b, l1, c1)))       # window-major y added to raster x1 (mTVE:285-286)
    x1, z = ResidualLayerNormFn.apply(x1, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
    hmid = GeluFn.apply(LinearFn.apply(z, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
    return _residual_linear(x1, blk.drop_path, hmid, blk.mlp.fc2.weight, blk.mlp.fc2.bias), out


def encoder_train(enc, x):
    """Encoder.forward (encoder.py:11-18; ThreeViewSwinTransformer.forward mTVE:732-746) with a backward:
    x (B,T,3,224,224) -> (final_x (B,2304,7,7), view_x[4][3] of (B,1,L,C), dct_x (B,9,224,224)).  The DCT features carry no
    parameters and are computed without a tape."""
    m = enc.base
    b = x.shape[0]
    with torch.no_grad():
        dct_x = m.faf.forward_frame(x, 1)
    tok = m.tokenize
    xs = [_tubelet_tokens(x, getattr(tok, f"project{v + 1}"), getattr(tok, f"norm{v + 1}")) for v in range(3)]
    tdims = m.input_token_temporal_dims
    view_x = []
    for layer in m.layers.layers:
        for i, blk in enumerate(layer.blocks):
            if i == 0:                                                    # cross block: view 3 -> 2 -> 1 (mTVE:345-350)
                xs[2], out2 = cross_swin_block_train(blk.block3, xs[2], None)
                xs[1], out1 = cross_swin_block_train(blk.block2, xs[1], out2)
                xs[0], _ = cross_swin_block_train(blk.block1, xs[0], out1)
            else:
                for v in range(3):
                    sub = getattr(blk, f"block{v + 1}")
                    if not isinstance(sub, torch.nn.Identity):
                        xs[v] = swin_block_train(sub, xs[v])
        view_x.append([t.unsqueeze(1) for t in xs])                       # captured before the downsample (mTVE:535)
        if layer.downsample is not None:
            xs = [patch_merging_train(getattr(layer.downsample, f"downsample{v + 1}"), xs[v]) for v in range(3)]
    tmax = max(tdims)
    parts = []
    for v, t in enumerate(xs):                                            # merge_views_along_channel_axis (mTVE:710-718)
        _, l, c = t.shape
        parts.append(t.reshape(b, tdims[v], l // tdims[v], c).repeat(1, tmax // tdims[v], 1, 1))
    g = torch.cat(parts, -1)                                              # (B, T, 49, 2560)
    g = LinearFn.apply(g.contiguous(), m.globalembedding.weight, m.globalembedding.bias)
    g = g.permute(0, 2, 1, 3).reshape(b * 49, tmax, g.shape[-1]).contiguous()               # site-major sequences of T tokens
    for gb in m.globalblocks.blocks:
        g = global_block_train(gb, g)
    g = g.reshape(b, 49, tmax, -1)
    final = torch.cat([g[:, :, 0], g[:, :, 1], g[:, :, 2]], -1)           # temporal slices 0,1,2 only (mTVE:745)
    return final.reshape(b, 7, 7, -1).permute(0, 3, 1, 2), view_x, dct_x
