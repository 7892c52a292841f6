"""Swin operators of the Mumpy encoder on the MI355X HIP kernels.

Same public names, constructor arguments and state_dict keys as the reference's models/modules/swinTransformer.py
(so its checkpoints load strictly), different execution: tokens stay in RASTER order end to end.  LayerNorm and the
QKV / proj / MLP Linears are per-token, so they never need the window layout; the window gather, the cyclic shift and
the inverse scatter are folded into the addressing of one fused attention kernel (mumpy_window_attention_fwd), and
bias+GELU / bias+residual ride in the GEMM epilogue (mumpy_linear_fwd).  A Swin block is 7 launches:
LN, QKV GEMM, window attention, proj GEMM(+residual), LN, fc1 GEMM(+GELU), fc2 GEMM(+residual).
"""
import torch
import torch.nn as nn

from models.modules.layers import Derived, DropPath, refuse_stochastic_depth, to_2tuple, trunc_normal_
from mumpy_hip import ops


def window_partition(x, window_size):
    """(B,H,W,C) -> (B*nW, ws, ws, C).  API parity with swin:54-66; the HIP forward never materialises this."""
    b, h, w, c = x.shape
    x = x.reshape(b, h // window_size, window_size, w // window_size, window_size, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, window_size, window_size, c)


def window_reverse(windows, window_size, H, W):
    """(B*nW, ws, ws, C) -> (B,H,W,C).  API parity with swin:69-83."""
    H, W = int(H), int(W)
    b = windows.shape[0] // ((H // window_size) * (W // window_size))
    x = windows.reshape(b, H // window_size, W // window_size, window_size, window_size, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, H, W, -1)


def build_shift_mask(hs, w, window_size, shift_size):
    """attn_mask buffer of a shifted block: 0 / -100, regions cut on the stacked (t*H, W) grid (swin:233-252)."""
    region = torch.zeros(hs, w)
    cuts = (slice(0, -window_size), slice(-window_size, -shift_size), slice(-shift_size, None))
    k = 0
    for rs in cuts:
        for cs in cuts:
            region[rs, cs] = k
            k += 1
    ids = window_partition(region.view(1, hs, w, 1), window_size).reshape(-1, window_size * window_size)
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def relative_position_index(ws_h, ws_w):
    ys, xs = torch.meshgrid(torch.arange(ws_h), torch.arange(ws_w), indexing="ij")
    coords = torch.stack([ys.flatten(), xs.flatten()])                       # (2, N)
    rel = coords[:, :, None] - coords[:, None, :]                             # (2, N, N)
    return (rel[0] + ws_h - 1) * (2 * ws_w - 1) + (rel[1] + ws_w - 1)


def _w16(module, name):
    """bf16 copy of module.<name>.weight, cached per weight version (Derived)."""
    lin = getattr(module, name)
    cache = module.__dict__.setdefault("_w16_cache", {})
    d = cache.get(name)
    if d is None:
        d = cache[name] = Derived()
    return d.get((lin.weight,), lambda: lin.weight.detach().to(torch.bfloat16).contiguous())


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if act_layer is not nn.GELU:
            raise NotImplementedError("the GEMM epilogue implements exact-erf GELU only")
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, residual=None, emit_stats=False):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias, act=ops.ACT_GELU)
        return ops.linear(h, self.fc2.weight, self.fc2.bias, residual=residual, emit_stats=emit_stats)

    def forward_ln(self, x, stats, norm, emit_stats=False):
        """x + fc2(GELU(fc1(LayerNorm(x)))) with the LayerNorm folded into fc1 (x raw, stats from x's producer)."""
        d = self.__dict__.setdefault("_ln_fold", Derived())
        wg, cs, bp = d.get((self.fc1.weight, self.fc1.bias, norm.weight, norm.bias),
                           lambda: ops.fold_ln_weights(self.fc1.weight, self.fc1.bias, norm.weight, norm.bias))
        h = ops.linear_ln(x, stats, wg, cs, bp, norm.eps, act=ops.ACT_GELU)
        return ops.linear(h, self.fc2.weight, self.fc2.bias, residual=x, emit_stats=emit_stats)

    def forward_bf16(self, x16, residual):
        """bf16 storage (ops.set_storage("bf16")): x16 is the bf16 LayerNorm output, the 4C hidden tensor stays bf16, fc2
        adds into the fp32 residual stream."""
        w1, w2 = _w16(self, "fc1"), _w16(self, "fc2")
        h = ops.linear_bf16s(x16, w1, self.fc1.bias, act=ops.ACT_GELU, out_bf16=True)
        return ops.linear_bf16s(h, w2, self.fc2.bias, residual=residual, out_bf16=False)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        if self.window_size != (7, 7) or dim // num_heads != 32:
            raise NotImplementedError("HIP window attention is built for 7x7 windows and 32-wide heads "
                                      f"(got window {self.window_size}, head width {dim // num_heads})")
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros(13 * 13, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(7, 7))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=0.02)
        self._bias = Derived()
        self._mask = Derived()

    def padded_bias(self):
        return self._bias.get((self.relative_position_bias_table, self.relative_position_index),
                              lambda: ops.expand_relpos_bias(self.relative_position_bias_table,
                                                             self.relative_position_index))

    def mask_pack(self, mask):
        if mask is None:
            return None, None
        return self._mask.get((mask,), lambda: ops.compact_attn_mask(mask))

    def attend(self, x_normed, b, hs, w, shift, mask):
        """x_normed (B, hs*w, C) raster -> projected W-MSA output (B, hs*w, C) raster, optional fused residual later."""
        tab, ids = self.mask_pack(mask)
        if x_normed.dtype == torch.bfloat16:                      # bf16 storage: bf16 in, bf16 qkv, bf16 out
            qkv = ops.linear_bf16s(x_normed, _w16(self, "qkv"), self.qkv.bias, out_bf16=True)
            return ops.window_attention_bf16(qkv, self.padded_bias(), b, hs, w, self.dim, shift, self.scale, tab, ids)
        qkv = ops.linear(x_normed, self.qkv.weight, self.qkv.bias)
        return ops.window_attention(qkv, self.padded_bias(), b, hs, w, self.dim, shift, self.scale, tab, ids)

    def attend_ln(self, x, stats, norm, b, hs, w, shift, mask):
        """attend(LayerNorm(x)) with the LayerNorm folded into the qkv GEMM (x raw, stats from x's producer)."""
        tab, ids = self.mask_pack(mask)
        d = self.__dict__.setdefault("_ln_fold", Derived())
        wg, cs, bp = d.get((self.qkv.weight, self.qkv.bias, norm.weight, norm.bias),
                           lambda: ops.fold_ln_weights(self.qkv.weight, self.qkv.bias, norm.weight, norm.bias))
        qkv = ops.linear_ln(x, stats, wg, cs, bp, norm.eps)
        return ops.window_attention(qkv, self.padded_bias(), b, hs, w, self.dim, shift, self.scale, tab, ids)

    def forward(self, x, mask=None):
        """Reference signature (swin:134): x (num_windows*B, 49, C) already partitioned; mask (nW,49,49) or None.
        Each window is treated as its own 7x7 image, so the same fused kernel serves this entry point."""
        b_ = x.shape[0]
        a = self.attend(x, b_, 7, 7, 0, mask)
        return ops.linear(a, self.proj.weight, self.proj.bias)

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 temporal_dim=1, fused_window_process=False):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size, self.mlp_ratio, self.temporal_dim = window_size, shift_size, mlp_ratio, temporal_dim
        if min(self.input_resolution) <= self.window_size:      # one window per frame: no shift (swin:217-220)
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, to_2tuple(self.window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        mask = None
        if self.shift_size > 0:
            h, w = self.input_resolution
            mask = build_shift_mask(h * temporal_dim, w, self.window_size, self.shift_size)
        self.register_buffer("attn_mask", mask)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        assert l % (h * w) == 0, "input feature has wrong size"
        hs = l // w                                              # frames stacked on rows (swin:267)
        if self.training:
            refuse_stochastic_depth(self)
        if ops.storage() == "bf16":                              # config 3: bf16 tensors between the kernels of the block
            a = self.attn.attend(ops.layernorm_bf16(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), b, hs, w,
                                 self.shift_size, self.attn_mask)
            x = ops.linear_bf16s(a, _w16(self.attn, "proj"), self.attn.proj.bias, residual=x, out_bf16=False)
            return self.mlp.forward_bf16(ops.layernorm_bf16(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), x)
        # LayerNorm folding (round 3): when the GEMM that produced x ran on the persistent kernel it left per-tile row statistics
        # on x, and when the consuming GEMM runs there too it takes the RAW x and finishes the normalisation in its epilogue --
        # no LayerNorm launch, no normalised copy of x (ops.linear_ln).  Anything else takes the two-launch route.
        m = b * l
        st = ops.ln_stats_of(x)
        fold1 = st is not None and ops.linear_ln_tiles(m, 3 * c, c) > 0
        fold2 = ops.linear_ln_tiles(m, c, c) > 0 and ops.linear_ln_tiles(m, self.mlp.fc1.out_features, c) > 0
        if fold1:
            a = self.attn.attend_ln(x, st, self.norm1, b, hs, w, self.shift_size, self.attn_mask)
        else:
            a = self.attn.attend(ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), b, hs, w,
                                 self.shift_size, self.attn_mask)
        x = ops.linear(a, self.attn.proj.weight, self.attn.proj.bias, residual=x, emit_stats=fold2)
        st = ops.ln_stats_of(x)
        # (the block's output feeds the next block's norm1 -- or a patch merging, which ignores the statistics)
        emit = ops.linear_ln_tiles(m, c, self.mlp.fc2.in_features) > 0 and ops.linear_ln_tiles(m, 3 * c, c) > 0
        if st is not None:
            return self.mlp.forward_ln(x, st, self.norm2, emit_stats=emit)
        return self.mlp(ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=x, emit_stats=emit)

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, temporal_dim={self.temporal_dim}")


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = tuple(input_resolution), dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        assert l == h * w, "input feature has wrong size"
        assert h % 2 == 0 and w % 2 == 0, f"x size ({h}*{w}) are not even."
        g = ops.patch_merge_ln(x, self.norm.weight, self.norm.bias, b, h, w, c, self.norm.eps)
        return ops.linear(g, self.reduction.weight)

    def extra_repr(self):
        return f"input_resolution={self.input_resolution}, dim={self.dim}"


class ThreeViewPatchMerging(nn.Module):
    def __init__(self, view_configs, cur_stage):
        super().__init__()
        for v in range(3):
            side = view_configs[v]["input_resolution"][cur_stage][0]      # square grids only (swin:640-643)
            merge = PatchMerging((view_configs[v]["temporal_dim"] * side, side), view_configs[v]["hidden_size"][cur_stage])
            setattr(self, f"downsample{v + 1}", merge)

    def forward(self, x):
        return [self.downsample1(x[0]), self.downsample2(x[1]), self.downsample3(x[2])]


class BaselineTokenize(nn.Module):
    """Conv3d(k=s=(t,4,4)) squeezing T + LayerNorm (swin:11-32) on the implicit-GEMM kernel."""

    def __init__(self, view_configs):
        super().__init__()
        size = view_configs["patches"].size
        self.patch_size = size[0]
        self.patches_resolution = view_configs["input_resolution"][0]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        k = (size[-1], size[0], size[1])
        self.proj = nn.Conv3d(3, view_configs["hidden_size"][0], kernel_size=k, stride=k, padding=0)
        self.norm = nn.LayerNorm(view_configs["hidden_size"][0])
        self._wt = Derived()

    def forward(self, x):
        w = self.proj.weight
        wt = self._wt.get((w,), lambda: w.reshape(w.shape[0], -1).t().contiguous())
        y = ops.patch_embed(x, wt, self.proj.bias, self.norm.weight, self.norm.bias, w.shape[2], self.norm.eps)
        if y.shape[1] != self.num_patches:
            raise RuntimeError("BaselineTokenize expects the tubelet to span the whole clip (squeeze(-3), swin:29)")
        return y


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 fused_window_process=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 mlp_ratio, qkv_bias, qk_scale, drop, attn_drop,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


class SwinTransformer(nn.Module):
    """Single-view Swin-B used by create_baseline() (config 1; swin:502-634)."""

    def __init__(self, view_configs, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96,
                 depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, use_checkpoint=False, fused_window_process=False, **kwargs):
        super().__init__()
        if ape:
            raise NotImplementedError("absolute position embedding is off in every reference config")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.patch_embed = BaselineTokenize(view_configs)
        self.patches_resolution = res = self.patch_embed.patches_resolution
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), (res[0] // 2 ** i, res[1] // 2 ** i), depths[i],
                                          num_heads[i], window_size, mlp_ratio, qkv_bias, qk_scale, drop_rate,
                                          attn_drop_rate, dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer,
                                          PatchMerging if i < self.num_layers - 1 else None))
        self.norm = norm_layer(self.num_features)
        self.avgpool = nn.AdaptiveAvgPool1d(1)
        self.head = nn.Linear(self.num_features, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.LayerNorm):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    def forward_features(self, x):
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x)
        return ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps)

    def forward(self, x):
        x = self.forward_features(x)
        if isinstance(self.head, nn.Linear):
            x = ops.linear(x, self.head.weight, self.head.bias)
        return x
