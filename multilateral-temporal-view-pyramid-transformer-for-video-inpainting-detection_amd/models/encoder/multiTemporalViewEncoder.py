"""Multi-temporal-view pyramid encoder (ThreeViewSwinTransformer) on the MI355X HIP kernels.

Module tree, constructor arguments and state_dict keys follow the reference's
models/encoder/multiTemporalViewEncoder.py (so `Encoder().load_state_dict(ckpt)` is strict-compatible); the forward is
re-designed: three raster-ordered token tensors (B, t_v*H*W, C_v) flow through the stages, there is no vmap, no
window-partitioned tensor and no python list mutated in place.  Per cross block (block 0 of each stage):
    view3: W-MSA + MLP                                        -> out3 (pre-residual W-MSA output)
    view2: W-MSA, pre(out3) GEMM, deformable CVA, combine, MLP -> out2
    view1: W-MSA, pre(out2) GEMM, deformable CVA, combine, MLP
"""
import torch
import torch.nn as nn

from models.modules.blocks import Block
from models.modules.dct import FAF
from models.modules.deformableAttention import SwinDAttention
from models.modules.layers import Derived, DropPath, refuse_stochastic_depth, to_2tuple, trunc_normal_
from models.modules.swinTransformer import Mlp, SwinTransformerBlock, ThreeViewPatchMerging, WindowAttention
from mumpy_hip import ops
from mumpy_hip.streams import run_parallel


class CVAModule(nn.Module):
    def __init__(self, dim1, num_heads, window_size=7, temporal_dims=[], qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, cur_stage=0):
        super().__init__()
        self.crossattn = SwinDAttention(dim1, num_heads, attn_drop, n_groups=3)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x1, x2, mask=None, return_attention=False):
        """Reference signature (mTVE:134-139): windows in, (x1 + y, attn-or-None) out; return_attention=True returns the maps
        (B1, r*nH, 49, 49) alone, as the reference does."""
        if return_attention:
            return self.crossattn(x1, x2, return_attention=True)[1]
        y, _ = self.crossattn(x1, x2)
        return ops.add(x1, y), None


class CrossSwinBlock(nn.Module):
    def __init__(self, dim1, dim2, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, qkv_bias=True,
                 qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 fused_window_process=False, last_view=False, temporal_dims=1, cur_stage=0):
        super().__init__()
        if shift_size != 0:
            raise NotImplementedError("cross blocks are always un-shifted (mTVE:310,323,336)")
        self.dim, self.input_resolution, self.num_heads = dim1, tuple(input_resolution), num_heads
        self.window_size = min(window_size, min(self.input_resolution))
        self.shift_size, self.mlp_ratio = 0, mlp_ratio
        self.last_view, self.temporal_dims, self.cur_stage = last_view, temporal_dims, cur_stage
        self.norm1 = norm_layer(dim1)
        self.attn = WindowAttention(dim1, to_2tuple(self.window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim1)
        self.mlp = Mlp(dim1, int(dim1 * mlp_ratio), act_layer=act_layer, drop=drop)
        if last_view:
            self.pre, self.cva = nn.Identity(), nn.Identity()
        else:
            self.pre = nn.Linear(dim2, dim1)
            trunc_normal_(self.pre.weight, std=0.02)
            nn.init.zeros_(self.pre.bias)
            self.cva = CVAModule(dim1, num_heads, to_2tuple(self.window_size), temporal_dims, qkv_bias, qk_scale, drop,
                                 attn_drop, drop_path, cur_stage)
        self.register_buffer("attn_mask", None)

    def msa(self, x1, need_out=True):
        """First half: x1 + W-MSA(LN(x1)).  Returns (x1 after the residual, out = W-MSA output BEFORE the residual, which is
        what the next view's cross attention consumes: mTVE:275, 347-349).  need_out=False (view 1: nobody consumes its `out`,
        mTVE:349): the residual add rides in the projection's epilogue and `out` is None."""
        h, w = self.input_resolution
        b, l1, c1 = x1.shape
        if self.training:
            refuse_stochastic_depth(self)
        if ops.storage() == "bf16":                              # config 3: bf16 LN output / qkv / attention output
            from models.modules.swinTransformer import _w16
            a = self.attn.attend(ops.layernorm_bf16(x1, self.norm1.weight, self.norm1.bias, self.norm1.eps), b, l1 // w, w, 0, None)
            out = ops.linear_bf16s(a, _w16(self.attn, "proj"), self.attn.proj.bias, out_bf16=False)
            return ops.add(x1, out), out
        a = self.attn.attend(ops.layernorm(x1, self.norm1.weight, self.norm1.bias, self.norm1.eps), b, l1 // w, w, 0, None)
        if not need_out:
            return ops.linear(a, self.attn.proj.weight, self.attn.proj.bias, residual=x1), None
        out = ops.linear(a, self.attn.proj.weight, self.attn.proj.bias)
        return ops.add(x1, out), out

    def prep(self, x1):
        """The part of the cross attention that needs only this view's tokens (q projection + sampling positions): off the
        dependency chain msa(next view) -> pre -> sampling -> ... when it is issued before the wait for the next view."""
        if self.last_view:
            return None
        h, w = self.input_resolution
        b, l1, _ = x1.shape
        return self.cva.crossattn._prep(x1, (b, l1 // w, w))

    def tail(self, x1, x2, prep=None):
        """Second half: deformable cross-view attention against x2 (skipped for the last view) and the MLP."""
        h, w = self.input_resolution
        b, l1, c1 = x1.shape
        hs1 = l1 // w
        if not self.last_view:
            hs2 = x2.shape[1] // w
            x2p = ops.linear(x2, self.pre.weight, self.pre.bias)                    # per-token, raster (mTVE:283)
            # x1 + [x1 in window order] + [scrambled proj_out]  (mTVE:138, 285-286; deform:403)
            x1 = self.cva.crossattn.attend_combine(x1, x2p, b, hs1, w, hs2, prep)
        if ops.storage() == "bf16":
            return self.mlp.forward_bf16(ops.layernorm_bf16(x1, self.norm2.weight, self.norm2.bias, self.norm2.eps), x1)
        # (fc2 leaves row statistics for the first plain block's norm1 when both GEMMs run on the persistent kernel)
        m = b * l1
        emit = ops.linear_ln_tiles(m, c1, self.mlp.fc2.in_features) > 0 and ops.linear_ln_tiles(m, 3 * c1, c1) > 0
        return self.mlp(ops.layernorm(x1, self.norm2.weight, self.norm2.bias, self.norm2.eps), residual=x1, emit_stats=emit)

    def forward(self, x1, x2):
        """x1 (B, t1*H*W, C1), x2 (B, t2*H*W, C2) raster.  Returns (x1_new, out)."""
        x1, out = self.msa(x1)
        return self.tail(x1, x2), out


class CrossThreeViewSwinBlock(nn.Module):
    def __init__(self, view_configs, input_resolution, cur_stage, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, fused_window_process=False):
        super().__init__()
        kw = dict(shift_size=0, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                  drop_path=drop_path, cur_stage=cur_stage)
        hid = [view_configs[v]["hidden_size"][cur_stage] for v in range(3)]
        heads = [view_configs[v]["num_heads"][cur_stage] for v in range(3)]
        ws = [view_configs[v]["window_size"] for v in range(3)]
        self.block1 = CrossSwinBlock(hid[0], hid[1], input_resolution[0], heads[0], window_size=ws[0],
                                     temporal_dims=view_configs[0]["temporal_ratio"], **kw)
        self.block2 = CrossSwinBlock(hid[1], hid[2], input_resolution[1], heads[1], window_size=ws[1],
                                     temporal_dims=view_configs[0]["temporal_ratio"], **kw)
        self.block3 = CrossSwinBlock(hid[2], hid[2], input_resolution[2], heads[2], window_size=ws[2], last_view=True,
                                     temporal_dims=3, **kw)

    def forward(self, x):
        """Reference order (mTVE:345-350): block3(x3) -> block2(x2, out3) -> block1(x1, out2).  The three W-MSA halves are
        independent and a view's cross attention needs only the NEXT view's W-MSA output, so the block runs as three
        branches with two cross-branch events:
            current stream:  msa3 --(out3)--> mlp3
            side A:          msa2 --(out2)--> q2/offsets2 [wait out3] cva2 + mlp2
            side B:          msa1 ----------> q1/offsets1 [wait out2] cva1 + mlp1
        (the q projection and the offset network of a view need only its own tokens: they are issued before the wait)"""
        from mumpy_hip import streams
        if streams.SERIAL:
            x3, out3 = self.block3(x[2], x[2])
            x2, out2 = self.block2(x[1], out3)
            x1, _ = self.block1(x[0], out2)
            return [x1, x2, x3]
        main = torch.cuda.current_stream()
        sa, sb = streams._side_stream(main.device, 0), streams._side_stream(main.device, 1)
        fork, e3, e2 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()
        fork.record(main)
        streams._DEPTH[0] += 1
        try:
            x3a, out3 = self.block3.msa(x[2])
            e3.record(main)
            sa.wait_event(fork)
            x[1].record_stream(sa)
            with torch.cuda.stream(sa):
                x2a, out2 = self.block2.msa(x[1])
                e2.record(sa)
                prep2 = self.block2.prep(x2a)
                sa.wait_event(e3)
                out3.record_stream(sa)
                x2 = self.block2.tail(x2a, out3, prep2)
            sb.wait_event(fork)
            x[0].record_stream(sb)
            with torch.cuda.stream(sb):
                x1a, _ = self.block1.msa(x[0], need_out=False)
                prep1 = self.block1.prep(x1a)
                sb.wait_event(e2)
                out2.record_stream(sb)
                x1 = self.block1.tail(x1a, out2, prep1)
            x3 = self.block3.tail(x3a, None)
        finally:
            streams._DEPTH[0] -= 1
        main.wait_stream(sa)
        main.wait_stream(sb)
        x2.record_stream(main)
        x1.record_stream(main)
        return [x1, x2, x3]


class OriginalThreeViewSwinBlock(nn.Module):
    def __init__(self, view_configs, input_resolution, cur_stage, cur_lyr, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, fused_window_process=False):
        super().__init__()
        shift = 0 if cur_lyr % 2 == 0 else view_configs[0]["window_size"] // 2
        for v in range(3):
            if cur_lyr < view_configs[v]["depths"][cur_stage]:
                blk = SwinTransformerBlock(view_configs[v]["hidden_size"][cur_stage], input_resolution[v],
                                           view_configs[v]["num_heads"][cur_stage], view_configs[v]["window_size"], shift,
                                           mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path,
                                           norm_layer=norm_layer, temporal_dim=view_configs[v]["temporal_dim"])
            else:
                blk = nn.Identity()                         # shallower views idle in the tail of a stage (mTVE:415)
            setattr(self, f"block{v + 1}", blk)

    def forward(self, x):
        return [self.block1(x[0]), self.block2(x[1]), self.block3(x[2])]


class MultiViewBasicLayer(nn.Module):
    def __init__(self, view_configs, cur_stage, depth, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None, fused_window_process=False):
        super().__init__()
        res = [view_configs[k]["input_resolution"][cur_stage] for k in range(3)]
        blocks = []
        for i in range(depth):
            dp = drop_path[i] if isinstance(drop_path, list) else drop_path
            kw = dict(mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                      drop_path=dp, norm_layer=norm_layer)
            blocks.append(CrossThreeViewSwinBlock(view_configs, res, cur_stage, **kw) if i == 0 else
                          OriginalThreeViewSwinBlock(view_configs, res, cur_stage, i, **kw))
        self.blocks = nn.ModuleList(blocks)
        self.downsample = downsample(view_configs, cur_stage) if downsample is not None else None

    def _view_chain(self, v, x):
        """Blocks 1..d-1 of view v (independent of the other views, mTVE:445-450) and its patch merging."""
        for blk in self.blocks[1:]:
            x = getattr(blk, f"block{v + 1}")(x)
        out = x                                             # features BEFORE the downsample feed the decoder (mTVE:535)
        if self.downsample is not None:
            x = getattr(self.downsample, f"downsample{v + 1}")(x)
        return x, out

    def forward(self, x):
        x = self.blocks[0](x)                               # cross-view block: view 3 -> 2 -> 1 dependency chain
        # the three views are independent from here to the end of the stage: fork them (view 3, the heaviest, stays on
        # the current stream)
        # (views 1 / 2 run beside view 3's persistent GEMMs, which own every CU's LDS: they take the LDS-free "background" kernels)
        def bg(v):
            from mumpy_hip import streams
            if streams.SERIAL:
                return self._view_chain(v, x[v])
            with ops.background():
                return self._view_chain(v, x[v])
        res = run_parallel([lambda: bg(0), lambda: bg(1), lambda: self._view_chain(2, x[2])], [(x[0],), (x[1],), (x[2],)])
        return [r[0] for r in res], [r[1] for r in res]


class CreateStages(nn.Module):
    def __init__(self, view_configs, depths=[2, 2, 18, 2], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, stages=4,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=nn.LayerNorm, ape=False,
                 patch_norm=True, use_checkpoint=False, fused_window_process=False):
        super().__init__()
        dpr = torch.linspace(0, drop_path_rate, sum(depths)).tolist()
        self.layers = nn.ModuleList([
            MultiViewBasicLayer(view_configs, i, depths[i], mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate,
                                dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer,
                                ThreeViewPatchMerging if i < stages - 1 else None)
            for i in range(stages)])

    def forward(self, x):
        outs = []
        for layer in self.layers:
            x, o = layer(x)
            outs.append(o)
        return x, outs


class CrossThreeViewTokenize(nn.Module):
    """Three Conv3d(k=s=(t_v,4,4)) + LayerNorm tokenizers (mTVE:574-618) on the implicit-GEMM kernel; output of
    view v is (B, t_out*H/4*W/4, C_v) with frames already stacked on the token axis (mTVE:701-708)."""

    def __init__(self, view_configs):
        super().__init__()
        for v in range(3):
            size = view_configs[v]["patches"].size
            k = (size[-1], size[0], size[1])
            c = view_configs[v]["hidden_size"][0]
            setattr(self, f"project{v + 1}", nn.Conv3d(3, c, kernel_size=k, stride=k, padding=0))
        for v in range(3):
            setattr(self, f"norm{v + 1}", nn.LayerNorm(view_configs[v]["hidden_size"][0]))
        self._wt = [Derived(), Derived(), Derived()]

    def _view(self, v, x):
        proj, norm = getattr(self, f"project{v + 1}"), getattr(self, f"norm{v + 1}")
        w = proj.weight
        if tuple(w.shape[3:]) != (4, 4):
            raise NotImplementedError("tokenizer kernel is built for 4x4 spatial patches")
        wt = self._wt[v].get((w,), lambda: w.reshape(w.shape[0], -1).t().contiguous())
        return ops.patch_embed(x, wt, proj.bias, norm.weight, norm.bias, w.shape[2], norm.eps)

    def forward(self, x):
        # the three tokenizers read the same clip and are independent: views 1/2 (392 blocks each, latency-bound) run
        # on side streams beside view 3's launch instead of in front of it
        return run_parallel([lambda: self._view(0, x), lambda: self._view(1, x), lambda: self._view(2, x)], [(x,), (x,), (x,)])


class CreateGlobalBlocks(nn.Module):
    def __init__(self, global_encoder_config, dpr, dropout_rate):
        super().__init__()
        g = global_encoder_config
        self.blocks = nn.ModuleList([Block(g["hidden_size"], g["num_heads"], g["mlp_dim"], dropout_rate, dpr[i])
                                     for i in range(g["num_layers"])])

    def forward(self, x, keep_t=None):
        """keep_t: the caller only uses temporal tokens 0 .. keep_t-1 of the result (the encoder tail, mTVE:745): the last block
        then skips the others' projection / MLP work and returns (S, keep_t, C)."""
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            x = blk(x, keep_t=keep_t) if (i == last and keep_t is not None) else blk(x)
        return x


class ThreeViewSwinTransformer(nn.Module):
    def __init__(self, view_configs, input_token_temporal_dims, global_encoder_config, depths=[2, 2, 18, 2], mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, stages=4, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                 norm_layer=nn.LayerNorm, ape=False, patch_norm=True, use_checkpoint=False, fused_window_process=False):
        super().__init__()
        self.faf = FAF()
        self.tokenize = CrossThreeViewTokenize(view_configs)
        self.input_token_temporal_dims = list(input_token_temporal_dims)
        self.layers = CreateStages(view_configs, depths, mlp_ratio, qkv_bias, qk_scale, stages, drop_rate, attn_drop_rate,
                                   drop_path_rate, norm_layer)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.globalembedding = nn.Linear(2560, 768)
        self.global_dpr = torch.linspace(0, drop_path_rate, global_encoder_config["num_layers"]).tolist()
        self.globalblocks = CreateGlobalBlocks(global_encoder_config, self.global_dpr, drop_rate)

    def merge_views_along_channel_axis(self, tokens):
        """[(B, t_v*n, C_v)] -> (B, Tmax, n, sum C_v): shallower views repeated over time (mTVE:710-718)."""
        tmax = max(self.input_token_temporal_dims)
        parts = []
        for v, x in enumerate(tokens):
            b, l, c = x.shape
            t = self.input_token_temporal_dims[v]
            x = x.reshape(b, t, l // t, c)
            parts.append(x.expand(b, tmax, l // t, c) if t == 1 else x.repeat(1, tmax // t, 1, 1))
        return torch.cat(parts, dim=-1)

    def forward_stages(self, x):
        """DCT branch + tokenizer + the four pyramid stages -> (views after stage 3, per-stage features, dct)."""
        # the DCT branch is independent of the token path until the decoder: fork it (frame index 1 only, mTVE:734)
        (ffinfo,), (views, stage_out) = run_parallel(
            [lambda: (self.faf.forward_frame(x, 1),), lambda: self.layers(self.tokenize(x))], [(x,), (x,)])
        return views, [[v.unsqueeze(1) for v in stage] for stage in stage_out], ffinfo

    def forward_global(self, views, dense=True):
        """Channel-merge of the views + the 12 temporal ViT blocks -> tokens (B,49,2304).
        dense=False (the fused pipeline): the result is the strided view of the blocks' output -- the first three temporal
        slices of a site are 2304 contiguous floats of its T * 768 -- and the decoder copies it straight into its
        concatenated map; dense=True materialises it (one strided row copy)."""
        b = views[0].shape[0]
        t = max(self.input_token_temporal_dims)
        g = ops.merge_views(views, self.input_token_temporal_dims).reshape(b * 49, t, -1)     # one T-token sequence per site
        g = ops.linear(g, self.globalembedding.weight, self.globalembedding.bias)
        if t < 3:
            raise RuntimeError("ThreeViewSwinTransformer: the encoder tail takes temporal slices 0,1,2 (mTVE:745): T >= 3")
        g = self.globalblocks(g, keep_t=3)          # frames 0,1,2 only leave the last block (mTVE:745): (B*49, 3 or T, 768)
        tk = g.shape[1]
        g = g.reshape(b, 49, tk * 768)
        if tk == 3:
            return g                                                           # already frames 0,1,2 on channels, dense
        if not dense:
            return g[:, :, :3 * 768]
        out = torch.empty(b, 49, 3 * 768, device=g.device, dtype=torch.float32)
        ops.copy_rows(g, tk * 768, out, 3 * 768, b * 49, 3 * 768)
        return out

    def forward(self, x):
        """x (B,T,3,224,224) -> (tokens (B,49,2304), view_x[4][3] of (B,1,L,C), dct (B,9,224,224))."""
        views, out_x, ffinfo = self.forward_stages(x)
        return self.forward_global(views), out_x, ffinfo
