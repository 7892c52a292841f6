"""torch.autograd glue for TRAINING the Swin block on the HIP kernels (first slice: rows 5-7 of 8a).

Every Function's forward and backward is a C-ABI kernel call; torch contributes the tape, the tensors and the shape
bookkeeping only.  Linear backward reuses the forward GEMM (y = x W^T):
    dX = dY W      = linear(dY, W^T)           W^T via mumpy_transpose_fwd
    dW = dY^T X    = linear(dY^T, X^T)         both operands transposed, token dim zero-padded to a multiple of 32
    db = column sums of dY (fixed-order two-stage reduction)
`swin_block_train(block, x)` runs a `models.modules.swinTransformer.SwinTransformerBlock` through these Functions, with the
same maths as its inference forward (swin:259-307; DropPath must be the identity, i.e. drop_path = 0 or eval mode).
"""
import torch

from . import ops


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n, k = weight.shape
        dy2, x2 = dy.reshape(-1, n).contiguous(), x.reshape(-1, k)
        dx = ops.linear(dy2, ops.transpose(weight)).reshape(x.shape) if ctx.needs_input_grad[0] else None
        dw = ops.linear(ops.transpose(dy2, 32), ops.transpose(x2, 32)) if ctx.needs_input_grad[1] else None
        db = ops.col_sum(dy2) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return ops.layernorm(x, gamma, beta, eps)

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, gamma, dy.contiguous(), ctx.eps)
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


class WindowAttentionFn(torch.autograd.Function):
    """softmax(q k^T * scale + bias + mask) v on raster-ordered qkv; differentiable in qkv and the bias table."""

    @staticmethod
    def forward(ctx, qkv, table, rel_index, dims, mask_tab, mask_id):
        b, hs, w, c, shift, scale = dims
        bias_pad = ops.expand_relpos_bias(table.detach(), rel_index)
        ctx.save_for_backward(qkv, bias_pad, rel_index.to(torch.int32).reshape(-1).contiguous())
        ctx.dims, ctx.mask = dims, (mask_tab, mask_id)
        return ops.window_attention(qkv, bias_pad, b, hs, w, c, shift, scale, mask_tab, mask_id)

    @staticmethod
    def backward(ctx, dout):
        qkv, bias_pad, idx32 = ctx.saved_tensors
        b, hs, w, c, shift, scale = ctx.dims
        dqkv, dtable = ops.window_attention_bwd(qkv, dout.contiguous(), bias_pad, idx32, b, hs, w, c, shift, scale, *ctx.mask)
        return dqkv, dtable, None, None, None, None


def swin_block_train(block, x):
    """SwinTransformerBlock.forward (swin:259-307) with a backward: x (B, L, C) -> (B, L, C), gradients reach x and every
    parameter of the block (norm1/2, qkv, relative_position_bias_table, proj, fc1, fc2)."""
    if isinstance(block.drop_path, torch.nn.Module) and not isinstance(block.drop_path, torch.nn.Identity) and block.training:
        raise NotImplementedError("swin_block_train: stochastic depth (DropPath > 0 in train mode) is not implemented")
    h, w = block.input_resolution
    b, l, c = x.shape
    hs = l // w
    att = block.attn
    tab, ids = att.mask_pack(block.attn_mask)
    y = LayerNormFn.apply(x, block.norm1.weight, block.norm1.bias, block.norm1.eps)
    qkv = LinearFn.apply(y, att.qkv.weight, att.qkv.bias)
    a = WindowAttentionFn.apply(qkv, att.relative_position_bias_table, att.relative_position_index,
                                (b, hs, w, block.dim, block.shift_size, att.scale), tab, ids)
    x = AddFn.apply(x, LinearFn.apply(a, att.proj.weight, att.proj.bias))
    z = LayerNormFn.apply(x, block.norm2.weight, block.norm2.bias, block.norm2.eps)
    hmid = GeluFn.apply(LinearFn.apply(z, block.mlp.fc1.weight, block.mlp.fc1.bias))
    return AddFn.apply(x, LinearFn.apply(hmid, block.mlp.fc2.weight, block.mlp.fc2.bias))
