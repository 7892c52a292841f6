"""ORACLE — CPU restatement of the Mumpy forward hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product package never does; it fails loudly when the HIP library is missing.

What it is: a from-scratch, batched, list-free, vmap-free restatement in plain torch CPU ops
(fp32) of the reference's encoder + decoder forward, written as pure functions over a
state_dict (name -> tensor) so that it shares no code with the product's nn.Modules.
Each function cites the reference file:line it follows (paths relative to /root/reference).

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against the
fixtures in tests/golden/ that tests/golden/gen_goldens.py produced by running the real
reference in the build container (per-operator outputs, integer index maps, and whole-model
outputs for B=1/T=3, B=2/T=3 (cross-sample coupling), B=1/T=5, and config 1's encoder + decoder).

Backward: the functions are differentiable torch code, so torch autograd on them is the gradient oracle of the training
kernels.  That use is pinned as well: tests/test_swin_backward.py::test_oracle_autograd_matches_reference_block checks
output, input gradient and all parameter gradients of `swin_block` against the reference's own SwinTransformerBlock run
under autograd, and tests/test_train_tail.py checks `mask_loss` / `polynomial_lr_sequence` against the reference's
utils/loss.py and utils/optimizer/scheduler.py (fixtures: tests/golden/train_tail.npz, tests/golden/gen_train_goldens.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
WS = 7            # window size everywhere (factory:26)
HEAD_DIM = 32     # every Swin / deformable head is 32 wide (hidden/num_heads)


# ----------------------------------------------------------------------------------------------
# configuration (factory:38-62 restated as plain data)
# ----------------------------------------------------------------------------------------------
@dataclass
class MumpyConfig:
    frames: int = 3
    hidden: List[List[int]] = field(default_factory=lambda: [[96, 192, 384, 768], [96, 192, 384, 768],
                                                              [128, 256, 512, 1024]])
    heads: List[List[int]] = field(default_factory=lambda: [[3, 6, 12, 24], [3, 6, 12, 24], [4, 8, 16, 32]])
    view_depths: List[List[int]] = field(default_factory=lambda: [[2, 2, 6, 2], [2, 2, 18, 2], [2, 2, 18, 2]])
    depths: List[int] = field(default_factory=lambda: [2, 2, 18, 2])       # mTVE:676 ctor default
    res: List[int] = field(default_factory=lambda: [56, 28, 14, 7])
    global_heads: int = 12
    global_layers: int = 12

    @property
    def tubelets(self) -> List[int]:          # (T, T-1, 1): factory:39,41,43 for T=3; SURVEY 8d for T=5/9
        return [self.frames, self.frames - 1, 1]

    @property
    def token_t(self) -> List[int]:           # input_token_temporal_dims (factory:48)
        return [1, 1, self.frames]


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _ln(x, sd: SD, p: str):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _lin(x, sd: SD, p: str):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def window_token_index(hs: int, w: int, shift: int) -> torch.Tensor:
    """idx[n*49 + p] = raster token (row-major over the stacked (hs, w) grid) that sits at in-window
    position p of window n after roll(-shift) + window_partition (swin:54-66, 273-275).  The same map
    is the scatter map of window_reverse + roll(+shift) (swin:69-83, 294-295)."""
    ys = torch.arange(hs).view(hs // WS, 1, WS, 1)
    xs = torch.arange(w).view(1, w // WS, 1, WS)
    src = ((ys + shift) % hs) * w + (xs + shift) % w          # (nWy, nWx, 7, 7)
    return src.reshape(-1)


def shift_attn_mask(hs: int, w: int, shift: int) -> torch.Tensor:
    """(nW,49,49) 0/-100 mask; regions cut on the STACKED t*H axis (swin:233-252)."""
    img = torch.zeros(hs, w)
    cnt = 0
    for h in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
        for ww in (slice(0, -WS), slice(-WS, -shift), slice(-shift, None)):
            img[h, ww] = cnt
            cnt += 1
    mw = img.view(hs // WS, WS, w // WS, WS).permute(0, 2, 1, 3).reshape(-1, WS * WS)
    d = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))


# ----------------------------------------------------------------------------------------------
# row 1: FAF  (dct:42-49, 56-79; only frame index 1 is consumed, mTVE:734)
# ----------------------------------------------------------------------------------------------
def dct_matrix(n: int = 224) -> torch.Tensor:
    i = np.arange(n)[:, None].astype(np.float64)
    j = np.arange(n)[None, :].astype(np.float64)
    m = np.sqrt(2.0 / n) * np.cos((j + 0.5) * np.pi * i / n)
    m[0, :] = np.sqrt(1.0 / n)
    return torch.tensor(m).float()            # fp64 build then .float(), as dct.py:60


def band_masks(n: int = 224) -> torch.Tensor:
    """(3,n,n) 0/1 masks on i+j: low [0, n//2.82], mid [n//2.82, n//2], high [n, 2n] (dct:46-47, 66-68)."""
    s = (torch.arange(n).view(n, 1) + torch.arange(n).view(1, n)).double()
    bands = [(0.0, n // 2.82), (n // 2.82, n // 2), (float(n), float(2 * n))]
    return torch.stack([((s >= lo) & (s <= hi)).float() for lo, hi in bands])


def faf_frame1(x: torch.Tensor) -> torch.Tensor:
    """x (B,T,3,224,224) -> (B,9,224,224): band-passed reconstructions of frame index 1."""
    d = dct_matrix(x.shape[-1])
    dt = d.t().contiguous()
    f = x[:, 1]                                               # (B,3,H,W)
    xf = d @ f @ dt                                           # dct:72
    ys = [dt @ (xf * m) @ d for m in band_masks(x.shape[-1])]  # dct:74-76
    return torch.cat(ys, dim=1)                               # band-major channels: [low rgb, mid rgb, high rgb]


# ----------------------------------------------------------------------------------------------
# row 2-3: tokenizer (mTVE:574-618) + temporal alignment (mTVE:701-708)
# ----------------------------------------------------------------------------------------------
def tokenize(x: torch.Tensor, sd: SD, cfg: MumpyConfig, p: str = "base.tokenize") -> List[torch.Tensor]:
    """x (B,T,3,H,W) -> three views (B, t_v*3136, C_v), frames stacked on the token axis."""
    xc = x.permute(0, 2, 1, 3, 4)                             # b c t h w
    outs = []
    for v in range(3):
        t = cfg.tubelets[v]
        y = F.conv3d(xc, sd[f"{p}.project{v + 1}.weight"], sd[f"{p}.project{v + 1}.bias"], stride=(t, 4, 4))
        b, c, tt, h, w = y.shape
        y = y.permute(0, 2, 3, 4, 1).reshape(b, tt * h * w, c)
        outs.append(_ln(y, sd, f"{p}.norm{v + 1}"))
    return outs


# ----------------------------------------------------------------------------------------------
# rows 4-5: window attention on a raster-ordered token tensor
# ----------------------------------------------------------------------------------------------
def window_attention_core(qkv: torch.Tensor, table: torch.Tensor, rel_index: torch.Tensor, hs: int, w: int, shift: int,
                          mask: Optional[torch.Tensor]) -> torch.Tensor:
    """The part of WindowAttention.forward between the two Linears (swin:143-163) on raster-ordered qkv (B, hs*w, 3C):
    window partition + roll folded into `idx`, q*scale, q k^T + relative position bias (+ shift mask), softmax, @ v,
    window reverse.  Returns (B, hs*w, C) raster.  Differentiable in qkv and table (used as the backward oracle)."""
    b, l, c3 = qkv.shape
    c = c3 // 3
    nh = c // HEAD_DIM
    idx = window_token_index(hs, w, shift)
    qw = qkv[:, idx].reshape(-1, WS * WS, 3, nh, HEAD_DIM).permute(2, 0, 3, 1, 4)     # (3, B*nW, nH, 49, 32)
    q, k, v = qw[0] * (HEAD_DIM ** -0.5), qw[1], qw[2]         # scale on q first (swin:145)
    attn = q @ k.transpose(-2, -1)
    bias = table[rel_index.reshape(-1)].view(49, 49, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(-1, nw, nh, 49, 49) + mask.view(1, nw, 1, 49, 49)).view(-1, nh, 49, 49)
    attn = attn.softmax(-1)
    y = (attn @ v).transpose(1, 2).reshape(b, l, c)
    out = torch.empty_like(y)
    out[:, idx] = y                                           # window_reverse + roll back
    return out


def window_attention(xn: torch.Tensor, sd: SD, p: str, hs: int, w: int, shift: int,
                     mask: Optional[torch.Tensor]) -> torch.Tensor:
    """xn (B, hs*w, C) already LayerNorm-ed; returns W-MSA output in raster order (swin:134-166, 266-301).  The two
    Linears are per-token, so applying them before the window gather / after the scatter is the same computation."""
    a = window_attention_core(_lin(xn, sd, p + ".qkv"), sd[p + ".relative_position_bias_table"],
                              sd[p + ".relative_position_index"], hs, w, shift, mask)
    return _lin(a, sd, p + ".proj")


def mlp(x, sd: SD, p: str):
    return _lin(F.gelu(_lin(x, sd, p + ".fc1")), sd, p + ".fc2")     # swin:45-51 (exact-erf GELU)


def swin_block(x: torch.Tensor, sd: SD, p: str, hs: int, w: int, shift: int) -> torch.Tensor:
    """SwinTransformerBlock.forward (swin:259-307); frames stacked on rows: hs = t*H."""
    if min(hs, w) <= WS or w <= WS:                           # swin:217-220 uses input_resolution (H,W)
        shift = 0
    mask = sd.get(p + ".attn_mask") if shift > 0 else None
    if shift > 0 and mask is None:
        mask = shift_attn_mask(hs, w, shift)
    x = x + window_attention(_ln(x, sd, p + ".norm1"), sd, p + ".attn", hs, w, shift, mask)
    return x + mlp(_ln(x, sd, p + ".norm2"), sd, p + ".mlp")


# ----------------------------------------------------------------------------------------------
# row 10: SwinDAttention (deform:324-405) restated per window, with its index quirks kept
# ----------------------------------------------------------------------------------------------
def _conv1x1(x, sd: SD, p: str):
    """x (..., C) token-major; 1x1 conv == per-token Linear with weight (Cout,Cin,1,1)."""
    wgt = sd[p + ".weight"]
    return F.linear(x, wgt.reshape(wgt.shape[0], -1), sd.get(p + ".bias"))


def deform_offsets(q: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """q (Bq,49,C) -> sampling positions (Bq, 3, 49, 2) as (y,x) in [-1,1] units (deform:334-349)."""
    bq, _, c = q.shape
    g, cg = 3, c // 3
    qg = q.reshape(bq, 7, 7, g, cg).permute(0, 3, 4, 1, 2).reshape(bq * g, cg, 7, 7)
    o = F.conv2d(qg, sd[p + ".conv_offset.0.weight"], sd[p + ".conv_offset.0.bias"], padding=2, groups=cg)
    o = o.permute(0, 2, 3, 1)                                 # LayerNormProxy: norm over channels
    o = F.gelu(_ln(o, sd, p + ".conv_offset.1.norm"))
    o = F.linear(o, sd[p + ".conv_offset.3.weight"].reshape(2, cg))         # (Bq*g,7,7,2)  (y,x)
    o = torch.tanh(o) * (1.0 / 7.0) * 2.0                     # deform:339-340, range factor 2
    ref = (torch.linspace(0.5, 6.5, 7) / 7.0) * 2.0 - 1.0     # deform:313-319
    ref = torch.stack(torch.meshgrid(ref, ref, indexing="ij"), -1)          # (7,7,2) (y,x)
    return (o + ref).reshape(bq, g, 49, 2)


def bilinear_sample_window(x2: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """x2 (B2,49,C) token-major windows, pos (B2,3,49,2) (y,x) normalised.  grid_sample semantics:
    bilinear, align_corners=True, zeros padding (deform:353-356).  Returns (B2,49,C) token-major."""
    b2, _, c = x2.shape
    g, cg = 3, c // 3
    img = x2.reshape(b2, 7, 7, g, cg).permute(0, 3, 4, 1, 2).reshape(b2 * g, cg, 7, 7)
    grid = pos.reshape(b2 * g, 7, 7, 2)[..., (1, 0)]          # (x,y)
    s = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return s.reshape(b2, g, cg, 49).permute(0, 3, 1, 2).reshape(b2, 49, c)


def swin_dattention(x1: torch.Tensor, x2: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """x1 (B1,49,C) q-windows, x2 (B2,49,C) kv-windows, B2 = r*B1.  Returns y (B1,49,C) exactly as
    SwinDAttention.forward does, including: q window = kv window mod B1 (x1.repeat, deform:330), sum over
    ADJACENT kv-window triples (deform:394-395) and the un-permuted (B,C,49)->(B,49,C) reshape (deform:403)."""
    b1, _, c = x1.shape
    b2 = x2.shape[0]
    r = b2 // b1
    nh = c // HEAD_DIM
    q = _conv1x1(x1, sd, p + ".proj_q")                       # only B1 distinct q's
    pos = deform_offsets(q, sd, p)                            # (B1,3,49,2)
    sel = torch.arange(b2) % b1
    samp = bilinear_sample_window(x2, pos[sel])               # (B2,49,C)
    k = _conv1x1(samp, sd, p + ".proj_k").reshape(b2, 49, nh, HEAD_DIM).transpose(1, 2)
    v = _conv1x1(samp, sd, p + ".proj_v").reshape(b2, 49, nh, HEAD_DIM).transpose(1, 2)
    qh = q[sel].reshape(b2, 49, nh, HEAD_DIM).transpose(1, 2)
    attn = ((qh @ k.transpose(-2, -1)) * (HEAD_DIM ** -0.5)).softmax(-1)     # scale on the product (deform:364)
    o = (attn @ v).transpose(1, 2).reshape(b2, 49, c)         # token-major (b2, p, head*32+d)
    o = o.reshape(b1, r, 49, c).sum(1)                        # adjacent r-tuples of kv windows
    y = _conv1x1(o, sd, p + ".proj_out")                      # (B1,49,C) token-major == (B1,C,7,7) channel-major
    return y.transpose(1, 2).reshape(b1, 49, c)               # flat (C,49) memory re-read as (49,C)


# ----------------------------------------------------------------------------------------------
# row 9: CrossSwinBlock (mTVE:228-291)
# ----------------------------------------------------------------------------------------------
def cross_swin_block(x1: torch.Tensor, x2: Optional[torch.Tensor], sd: SD, p: str, res: int,
                     last_view: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    b, l1, c1 = x1.shape
    hs1 = l1 // res
    out = window_attention(_ln(x1, sd, p + ".norm1"), sd, p + ".attn", hs1, res, 0, None)   # mTVE:238-275
    x1 = x1 + out
    if not last_view:
        hs2 = x2.shape[1] // res
        i1 = window_token_index(hs1, res, 0)
        i2 = window_token_index(hs2, res, 0)
        x1w = x1[:, i1].reshape(-1, 49, c1)                                                 # mTVE:280-281
        x2w = _lin(x2[:, i2].reshape(-1, 49, x2.shape[-1]), sd, p + ".pre")                 # mTVE:282-283
        y = x1w + swin_dattention(x1w, x2w, sd, p + ".cva.crossattn")                       # mTVE:138
        x1 = x1 + y.reshape(b, l1, c1)                # window-major y added to raster x1 (mTVE:285-286)
    x1 = x1 + mlp(_ln(x1, sd, p + ".norm2"), sd, p + ".mlp")
    return x1, out


# ----------------------------------------------------------------------------------------------
# row 8: patch merging on the stacked grid (swin:344-367, 637-657)
# ----------------------------------------------------------------------------------------------
def patch_merging(x: torch.Tensor, sd: SD, p: str, hs: int, w: int) -> torch.Tensor:
    b, l, c = x.shape
    x = x.view(b, hs, w, c)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.reshape(b, -1, 4 * c)
    return F.linear(_ln(x, sd, p + ".norm"), sd[p + ".reduction.weight"])


# ----------------------------------------------------------------------------------------------
# row 13: global temporal ViT blocks (blocks:37-92; vmapped over the 49 sites, mTVE:741)
# ----------------------------------------------------------------------------------------------
def global_block(x: torch.Tensor, sd: SD, p: str, heads: int) -> torch.Tensor:
    """x (S, T, 768): S = B*49 independent sequences of T temporal tokens."""
    s, t, c = x.shape
    hd = c // heads
    qkv = _lin(_ln(x, sd, p + ".norm1"), sd, p + ".attn.qkv").reshape(s, t, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(-1)   # scale on product (blocks:66)
    y = (attn @ qkv[2]).transpose(1, 2).reshape(s, t, c)
    x = x + _lin(y, sd, p + ".attn.proj")
    return x + _lin(F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")


# ----------------------------------------------------------------------------------------------
# rows 11-14: encoder graph (mTVE:732-746, encoder.py:11-18)
# ----------------------------------------------------------------------------------------------
def merge_views_along_channel(views: List[torch.Tensor], token_t: List[int]) -> torch.Tensor:
    """views[v] (B, t_v*n, C_v) -> (B, Tmax, n, sum C) (mTVE:710-718 / decoder:43-51)."""
    tmax = max(token_t)
    xs = []
    for v, x in enumerate(views):
        b, l, c = x.shape
        x = x.reshape(b, token_t[v], l // token_t[v], c)
        xs.append(x.repeat(1, tmax // token_t[v], 1, 1))
    return torch.cat(xs, -1)


def encoder_forward(sd: SD, x: torch.Tensor, cfg: Optional[MumpyConfig] = None):
    """Encoder.forward: x (B,T,3,224,224) -> (final_x (B,2304,7,7), view_x[4][3] (B,1,L,C), dct_x (B,9,224,224))."""
    cfg = cfg or MumpyConfig(frames=x.shape[1])
    b = x.shape[0]
    dct_x = faf_frame1(x)
    xs = tokenize(x, sd, cfg)
    view_x = []
    for s in range(4):
        res = cfg.res[s]
        hs = [cfg.token_t[v] * res for v in range(3)]
        base = f"base.layers.layers.{s}"
        for i in range(cfg.depths[s]):
            p = f"{base}.blocks.{i}"
            if i == 0:                                                       # mTVE:345-350
                xs[2], out2 = cross_swin_block(xs[2], None, sd, p + ".block3", res, True)
                xs[1], out1 = cross_swin_block(xs[1], out2, sd, p + ".block2", res, False)
                xs[0], _ = cross_swin_block(xs[0], out1, sd, p + ".block1", res, False)
            else:                                                            # mTVE:445-450
                for v in range(3):
                    if i < cfg.view_depths[v][s]:
                        xs[v] = swin_block(xs[v], sd, f"{p}.block{v + 1}", hs[v], res, 3 if i % 2 else 0)
        view_x.append([t.unsqueeze(1) for t in xs])                          # mTVE:535 (pre-downsample)
        if s < 3:
            xs = [patch_merging(xs[v], sd, f"{base}.downsample.downsample{v + 1}", hs[v], res) for v in range(3)]
    g = merge_views_along_channel(xs, cfg.token_t)                           # (B,T,49,2560)
    g = _lin(g, sd, "base.globalembedding")
    t = g.shape[1]
    g = g.permute(0, 2, 1, 3).reshape(b * 49, t, 768)                        # site-major sequences of T tokens
    for i in range(cfg.global_layers):
        g = global_block(g, sd, f"base.globalblocks.blocks.{i}", cfg.global_heads)
    g = g.reshape(b, 49, t, 768)
    final = torch.cat([g[:, :, 0], g[:, :, 1], g[:, :, 2]], -1)              # frames 0,1,2 only (mTVE:745)
    final = final.reshape(b, 7, 7, 2304).permute(0, 3, 1, 2).contiguous()    # encoder.py:16-17
    return final, view_x, dct_x


# ----------------------------------------------------------------------------------------------
# row 15: decoder (decoder:183-225)
# ----------------------------------------------------------------------------------------------
def _gn(x, sd: SD, p: str, groups: int):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _conv(x, sd: SD, p: str, padding):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def _gcm(x, sd: SD, p: str):
    l = _conv(_conv(x, sd, p + ".conv_l1", (3, 0)), sd, p + ".conv_l2", (0, 3))
    r = _conv(_conv(x, sd, p + ".conv_r1", (0, 3)), sd, p + ".conv_r2", (3, 0))
    return l + r


def _up(x, scale, align):
    return F.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=align)


def decoder_forward(sd: SD, x: torch.Tensor, view_x, ffinfo: torch.Tensor, token_t: List[int]):
    """Decoder.forward(x, view_x, ffinfo) -> (logits (B,1,224,224), x_feats (B,32,224,224))."""
    shape = [56, 28, 14, 7]
    rgb = []
    for s in range(4):
        m = merge_views_along_channel([v.squeeze(1) for v in view_x[s]], token_t)        # (B,T,n,C')
        b, t, n, c = m.shape
        m = m.permute(0, 3, 1, 2).reshape(b, c, t, shape[s], shape[s])
        y = F.conv3d(m, sd[f"rgb_decoder_{s + 1}.0.weight"], sd[f"rgb_decoder_{s + 1}.0.bias"],
                     stride=(t, 1, 1)).squeeze(2)
        rgb.append(F.relu(_gn(y, sd, f"rgb_decoder_{s + 1}.1", 16)))
    rgb1, rgb2, rgb3, rgb4 = rgb
    freq = []
    f = ffinfo
    for i, gsz in enumerate([8, 8, 8, 4, 8]):
        p = f"decoder_frequency_{i}"
        f = torch.sigmoid(_gn(_conv(F.avg_pool2d(f, 2), sd, p + ".1", 1), sd, p + ".2", gsz))
        freq.append(f)
    gcn0 = _gcm(torch.cat([rgb4, x], 1), sd, "gcm1")
    out1 = F.pixel_shuffle(gcn0 * freq[4], 2)
    seb1 = rgb3 * _up(_conv(rgb4, sd, "seb1.conv", 1), 2, False)
    gcn1 = _gcm(seb1, sd, "gcm2")
    seb2 = rgb2 * _up(_conv(torch.cat([rgb3, _up(rgb4, 2, False)], 1), sd, "seb2.conv", 1), 2, False)
    gcn2 = _gcm(seb2, sd, "gcm3")
    seb3 = rgb1 * _up(_conv(torch.cat([rgb2, _up(rgb3, 2, False), _up(rgb4, 4, False)], 1), sd, "seb3.conv", 1),
                      2, False)
    gcn3 = _gcm(seb3, sd, "gcm4")

    def dec(z, p):
        return _up(F.relu(_gn(_conv(z, sd, p + ".0", 1), sd, p + ".1", 8)), 2, True)

    z = dec(gcn1 * freq[3] + out1, "decoder_2")
    z = dec(z + gcn2 * freq[2], "decoder_3")
    z = dec(z + gcn3 * freq[1], "decoder_4")
    z = dec(z * freq[0], "decoder_5")
    feats = F.avg_pool2d(F.pixel_shuffle(z, 2), 2)
    return _conv(feats, sd, "final_out", 1), feats


def full_forward(sd_enc: SD, sd_dec: SD, x: torch.Tensor):
    cfg = MumpyConfig(frames=x.shape[1])
    fx, vx, dx = encoder_forward(sd_enc, x, cfg)
    logits, feats = decoder_forward(sd_dec, fx, vx, dx, cfg.token_t)
    return logits, feats, fx, vx, dx


# ----------------------------------------------------------------------------------------------
# row 17: single-scale baseline encoder (swin:502-634, encoder.py:22-30) — config 1
# ----------------------------------------------------------------------------------------------
def baseline_encoder_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    xc = x.permute(0, 2, 1, 3, 4)
    y = F.conv3d(xc, sd["base.patch_embed.proj.weight"], sd["base.patch_embed.proj.bias"], stride=(3, 4, 4))
    b, c = y.shape[:2]
    y = _ln(y.squeeze(2).flatten(2).transpose(1, 2), sd, "base.patch_embed.norm")
    res = 56
    for s, depth in enumerate([2, 2, 18, 2]):
        for i in range(depth):
            y = swin_block(y, sd, f"base.layers.{s}.blocks.{i}", res, res, 3 if i % 2 else 0)
        if s < 3:
            y = patch_merging(y, sd, f"base.layers.{s}.downsample", res, res)
            res //= 2
    y = _ln(y, sd, "base.norm")
    return y.reshape(b, 7, 7, -1).permute(0, 3, 1, 2).contiguous()


def baseline_decoder_forward(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """BaselineDecoder.forward (decoder.py:277-284): 5 x [conv3x3 -> GroupNorm(32) -> ReLU -> bilinear x2 with
    align_corners=True] (decoder.py:233-271) then final_out conv3x3 (decoder.py:273).  sd: un-prefixed decoder state_dict."""
    for i in range(1, 6):
        x = F.conv2d(x, sd[f"decoder_{i}.0.weight"], sd[f"decoder_{i}.0.bias"], padding=1)
        x = F.relu(F.group_norm(x, 32, sd[f"decoder_{i}.1.weight"], sd[f"decoder_{i}.1.bias"], eps=1e-5))
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(x, sd["final_out.weight"], sd["final_out.bias"], padding=1)


# ----------------------------------------------------------------------------------------------
# eval-harness tail (test.py:100-111) and F1/IoU (measure.py:57-62, 86-89)
# ----------------------------------------------------------------------------------------------
def stage_frames(frames: torch.Tensor, size=None, mean=(0.4776, 0.479, 0.4465), std=(0.230, 0.2085, 0.2324)) -> torch.Tensor:
    """Input staging of the loader + eval transforms (universaldataset.py:75-79, test.py:22-25): frames (..., Hs, Ws, 3) uint8
    -> optional PIL `img.resize(inputRes)` with the default filter of the pinned pillow==4.0.0 (NEAREST: Pillow's Geometry.c
    ImagingScaleAffine walks xo = 0.5 a, xin = int(xo), xo += a with a = src / dst in double -- the accumulated value
    decides exact ties, so the walk is restated as a sequential float64 cumulative sum) -> ToTensor
    (/255) -> Normalize(mean, std) -> (..., 3, H, W) float32.  Pinned against PIL's own NEAREST resize in
    tests/test_oracle_golden.py::test_stage_frames_matches_pil_nearest."""
    hs, ws = frames.shape[-3], frames.shape[-2]
    if size is not None and tuple(size) != (hs, ws):
        h, w = size
        def walk(src, dst):
            a = src / dst
            steps = torch.full((dst,), a, dtype=torch.float64)
            steps[0] = a * 0.5
            return torch.clamp(torch.cumsum(steps, 0).floor().long(), max=src - 1)      # cumsum adds left to right
        yi, xi = walk(hs, h), walk(ws, w)
        frames = frames.index_select(-3, yi).index_select(-2, xi)
    x = frames.movedim(-1, -3).float() / 255.0
    m = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
    sd = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return (x - m) / sd


def mask_from_logits(logits: torch.Tensor) -> torch.Tensor:
    return (torch.sigmoid(logits) > 0.5).to(torch.uint8)


def f1_iou_per_clip(pred: torch.Tensor, gt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-clip F1 / IoU exactly as measure.py:57-62 (iou_score) and :86-89 (evaluate_image):
    note the recall denominator is sum(gt + 1e-6) over ALL pixels.  pred, gt: (B,1,H,W) 0/1."""
    p = pred.reshape(pred.shape[0], -1).bool()
    g = gt.reshape(gt.shape[0], -1).bool()
    inter = (p & g).sum(1).double()
    union = (p | g).sum(1).double()
    recall = inter / (g.sum(1).double() + 1e-6 * p.shape[1])
    precision = inter / (p.sum(1).double() + 1e-6)
    f1 = 2 * (precision * recall) / (precision + recall + 1e-6)
    iou = (inter + 1e-5) / (union + 1e-5)
    return f1, iou


def metric_vector(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """float64[3] = [sum f1, sum iou, n_clips]: the vector that is all-reduced(sum) across ranks; the
    reported metrics are the means (measure.py:128-130)."""
    f1, iou = f1_iou_per_clip(pred, gt)
    return torch.stack([f1.sum(), iou.sum(), torch.tensor(float(pred.shape[0]), dtype=torch.float64)])


# ----------------------------------------------------------------------------------------------
# training tail — mask loss (utils/loss.py:6-55 as called at train.py:107-113), PolynomialLR
# (utils/optimizer/scheduler.py:24-41).  AdamW's checker is torch.optim.AdamW itself (utils/utils.py:258).
# ----------------------------------------------------------------------------------------------
def mask_loss(logits: torch.Tensor, target: torch.Tensor, eps: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (iou + focal, iou, focal); differentiable in `logits`.  logits (B,...), target 0/1 with the same per-sample size.
    softIoU: loss.py:27-42 with e = eps (the call site loss.py:49 passes recall=False into `e`, i.e. 0) averaged over the
    batch (loss.py:54, train.py:108); focal: loss.py:15-24 with alpha=[1,1] (loss.py:12), gamma=2, mean over all elements."""
    b = logits.shape[0]
    z = logits.reshape(b, -1)
    t = target.reshape(b, -1).to(z.dtype)
    p = torch.sigmoid(z)
    iou = (1 - (p * t).sum(1) / ((p + t - p * t).sum(1) + eps)).mean()
    bce = F.binary_cross_entropy_with_logits(z, t, reduction="none")
    focal = ((1 - torch.exp(-bce)) ** 2 * bce).mean()
    return iou + focal, iou, focal


def polynomial_lr_sequence(base_lr: float, iter_max: int, steps: int, power: float = 0.9, min_lr: float = 1e-5) -> List[float]:
    """Learning rates [before any step, after step 1, ...] of PolynomialLR(step_size=1, iter_warmup=0) (scheduler.py:24-41):
    unchanged at last_epoch 0 and past iter_max, else (base - min)(1 - it/iter_max)^power + min."""
    lrs, lr = [base_lr], base_lr
    for it in range(1, steps + 1):
        if it <= iter_max:
            lr = (base_lr - min_lr) * (1 - it / iter_max) ** power + min_lr
        lrs.append(lr)
    return lrs
