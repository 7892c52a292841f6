"""Global temporal ViT blocks (reference models/modules/blocks.py) on the HIP kernels.

The encoder applies these with the 49 spatial sites as the vmapped axis (mTVE:741), i.e. every site is an independent
sequence of T temporal tokens.  Here that is simply a (B*49, T, 768) batch: LN -> QKV GEMM -> one T x T attention
launch for all sites and heads -> proj GEMM(+residual) -> LN -> fc1 GEMM(+GELU) -> fc2 GEMM(+residual).
"""
import torch
import torch.nn as nn

from models.modules.layers import Derived, DropPath, refuse_stochastic_depth
from mumpy_hip import ops


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout, out_dim=None):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden_dim)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_dim, dim if out_dim is None else out_dim)
        self.drop = nn.Dropout(dropout)

    @property
    def unwrapped(self):
        return self

    def forward(self, x, residual=None, emit_stats=False):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, act=ops.ACT_GELU)
        return ops.linear(h, self.fc2.weight, self.fc2.bias, residual=residual, emit_stats=emit_stats)

    def forward_bf16(self, x16, residual):
        """bf16 storage: bf16 LayerNorm output in, bf16 hidden tensor, fc2 adds into the fp32 residual stream."""
        from models.modules.swinTransformer import _w16
        h = ops.linear_bf16s(x16, _w16(self, "fc1"), self.fc1.bias, act=ops.ACT_GELU, out_bf16=True)
        return ops.linear_bf16s(h, _w16(self, "fc2"), self.fc2.bias, residual=residual, out_bf16=False)


class Attention(nn.Module):
    def __init__(self, dim, heads, dropout):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        if dim // heads != 64:
            raise NotImplementedError("HIP temporal attention is built for 64-wide heads")
        self.attn = None
        self.qkv = nn.Linear(dim, dim * 3)
        self.attn_drop = nn.Dropout(dropout)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(dropout)

    @property
    def unwrapped(self):
        return self

    def forward(self, x, mask=None, residual=None, ln=None, keep_t=None):
        """x (S,T,C) -> (attention output after proj (+residual), None); the TxT map is not materialised.
        ln = (stats, norm): x is the RAW block input and the LayerNorm is folded into the qkv GEMM (ops.linear_ln).
        keep_t: only the first keep_t temporal tokens are queries -> (S, keep_t, C); residual must have that shape."""
        s, t, c = x.shape
        if ln is not None:
            stats, norm = ln
            d = self.__dict__.setdefault("_ln_fold", Derived())
            wg, cs, bp = d.get((self.qkv.weight, self.qkv.bias, norm.weight, norm.bias),
                               lambda: ops.fold_ln_weights(self.qkv.weight, self.qkv.bias, norm.weight, norm.bias))
            qkv = ops.linear_ln(x, stats, wg, cs, bp, norm.eps)
        else:
            qkv = ops.linear(x, self.qkv.weight, self.qkv.bias)
        a = ops.temporal_attention(qkv, s, t, c, self.heads, self.scale, tq=keep_t)
        return ops.linear(a, self.proj.weight, self.proj.bias, residual=residual), None


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_dim, dropout, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn = Attention(dim, heads, dropout)
        self.mlp = FeedForward(dim, mlp_dim, dropout)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x, mask=None, return_attention=False, keep_t=None):
        """keep_t (internal, the LAST global block only): the caller uses temporal tokens 0 .. keep_t-1 of the result alone
        (mTVE:745 keeps slices 0..2), so queries, projection, residual and MLP run on those tokens only -> (S, keep_t, C).
        Identical values for the kept tokens; 40 % of the block's proj / MLP work saved at T = 5."""
        if return_attention:          # blocks:85-87: the (S, heads, T, T) map of this block's attention, nothing else
            s_, t_, c_ = x.shape
            qkv = ops.linear(ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), self.attn.qkv.weight, self.attn.qkv.bias)
            return ops.attention_probs(qkv, qkv[..., c_:], s_, self.attn.heads, t_, t_, c_ // self.attn.heads,
                                       (t_ * 3 * c_, 3 * c_), (t_ * 3 * c_, 3 * c_), self.attn.scale)
        if self.training:
            refuse_stochastic_depth(self)
        if keep_t is not None and keep_t < x.shape[1] and ops.storage() == "fp32":
            s_, t_, c_ = x.shape
            xr = torch.empty(s_, keep_t, c_, device=x.device, dtype=torch.float32)
            ops.copy_rows(x, t_ * c_, xr, keep_t * c_, s_, keep_t * c_)            # the kept tokens of the residual stream, compact
            st = ops.ln_stats_of(x)
            ln = (st, self.norm1) if st is not None and ops.linear_ln_tiles(s_ * t_, 3 * c_, c_) > 0 else None
            xn = x if ln is not None else ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            y, _ = self.attn(xn, mask, residual=xr, ln=ln, keep_t=keep_t)
            return self.mlp(ops.layernorm(y, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=y)
        # norm1 folds into the qkv GEMM when the previous block's fc2 left row statistics on x (see SwinTransformerBlock.forward)
        m, c = x.numel() // x.shape[-1], x.shape[-1]
        st = ops.ln_stats_of(x)
        if st is not None and ops.linear_ln_tiles(m, 3 * c, c) > 0:
            x, _ = self.attn(x, mask, residual=x, ln=(st, self.norm1))
        else:
            x, _ = self.attn(ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), mask, residual=x)
        if ops.storage() == "bf16" and isinstance(self.mlp, FeedForward):
            return self.mlp.forward_bf16(ops.layernorm_bf16(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), x)
        emit = isinstance(self.mlp, FeedForward) and ops.linear_ln_tiles(m, c, self.mlp.fc1.out_features) > 0 and \
            ops.linear_ln_tiles(m, 3 * c, c) > 0
        if emit:
            return self.mlp(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=x, emit_stats=True)
        return self.mlp(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=x)
