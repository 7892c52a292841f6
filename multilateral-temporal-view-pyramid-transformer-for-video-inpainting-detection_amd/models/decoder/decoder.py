"""Multi-pyramid decoder — drop-in for the reference's models/decoder/decoder.py (`Decoder`), same constructor
defaults and the same 92 state_dict keys (nn.Sequential indices included).

Execution: the decoder is NHWC (torch.channels_last) end to end.
  * The four temporal heads Conv3d(C', 256, k=s=(T,1,1)) are not convolutions (kernel = stride = full extent) but
    per-pixel Linears over (C', T).  They run on the fp32 MFMA GEMM straight from the encoder's token-major stage
    outputs: views 1/2 are repeated over time by the reference (decoder.py:50), so their T weight slices are summed once
    and applied in one GEMM each; view 3 contributes T GEMMs over strided time slices chained through the residual
    input.  The (B,C',T,h,h) merged tensor of decoder.py:43-53 and its permute are never built; FLOPs drop ~2x.
  * Everything between two convolutions — GroupNorm, ReLU/Sigmoid, DAP (PixelShuffle+AvgPool == 4-channel mean),
    bilinear x2/x4 (both align_corners modes), the "+ gcn*freq" / "* freq" / SEB multiply — is two hand-written
    kernels (mumpy_gn_stats_nhwc_fwd, mumpy_gn_apply_resample_nhwc_fwd).  decoder_5's (B,128,224,224) output and
    DAP's (B,32,448,448) intermediate are never formed.
  * The spatial convolutions (3x3, 7x1, 1x7; 8 % of the forward's FLOPs) are the hand-written implicit GEMM
    mumpy_conv2d_nhwc_fwd (same MFMA tile machinery as the Linears; GCM's x_l + x_r rides in an epilogue), and
    final_out (32 -> 1) is a fused streaming kernel that can also emit the thresholded mask.  No MIOpen / library
    kernel is left on the path.  Returned tensors are logical NCHW with NHWC strides.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from models.modules.layers import Derived
from mumpy_hip import ops
from mumpy_hip.streams import run_parallel


class SEB(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear")

    def forward(self, x):
        x1, x2 = x
        return x1 * self.upsample(self.conv(x2))


class _GlobalConvModule(nn.Module):
    def __init__(self, in_dim, out_dim, kernel_size):
        super().__init__()
        p0, p1 = (kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2
        self.conv_l1 = nn.Conv2d(in_dim, out_dim, kernel_size=(kernel_size[0], 1), padding=(p0, 0))
        self.conv_l2 = nn.Conv2d(out_dim, out_dim, kernel_size=(1, kernel_size[1]), padding=(0, p1))
        self.conv_r1 = nn.Conv2d(in_dim, out_dim, kernel_size=(1, kernel_size[1]), padding=(0, p1))
        self.conv_r2 = nn.Conv2d(out_dim, out_dim, kernel_size=(kernel_size[0], 1), padding=(p0, 0))

    def forward(self, x):
        return self.conv_l2(self.conv_l1(x)) + self.conv_r2(self.conv_r1(x))


def _up_block(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.GroupNorm(8, cout), nn.ReLU(inplace=True),
                         nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))


def _freq_block(cin, cout, groups):
    return nn.Sequential(nn.AvgPool2d(2, stride=2), nn.Conv2d(cin, cout, 3, padding=1), nn.GroupNorm(groups, cout),
                         nn.Sigmoid())


class Decoder(nn.Module):
    def __init__(self, in_channels=2304, out_channels=1, kernel_size=7, num_classes=32, dap_k=2,
                 features=[256, 256, 256, 256, 256], input_token_temporal_dims=[1, 1, 3],
                 rgb_features=[320, 640, 1280, 2560], shape=[56, 28, 14, 7]):
        super().__init__()
        self.input_token_temporal_dims = list(input_token_temporal_dims)
        tmax = max(self.input_token_temporal_dims)
        self.shape = list(shape)
        wide = num_classes * dap_k ** 2
        self.decoder_2 = _up_block(num_classes, wide)
        self.decoder_3 = _up_block(wide, wide)
        self.decoder_4 = _up_block(wide, wide)
        self.decoder_5 = _up_block(wide, wide)
        self.final_out = nn.Conv2d(num_classes, out_channels, 3, padding=1)
        for i in range(4):
            setattr(self, f"rgb_decoder_{i + 1}", nn.Sequential(
                nn.Conv3d(rgb_features[i], features[i], kernel_size=(tmax, 1, 1), padding=0, stride=(tmax, 1, 1)),
                nn.GroupNorm(16, features[i]), nn.ReLU(inplace=True)))
        k = (kernel_size, kernel_size)
        self.gcm1 = _GlobalConvModule(features[-1] + in_channels, wide, k)
        self.gcm2 = _GlobalConvModule(features[-2], num_classes, k)
        self.gcm3 = _GlobalConvModule(features[-3], wide, k)
        self.gcm4 = _GlobalConvModule(features[-4], wide, k)
        self.ecre = nn.PixelShuffle(2)
        self.seb1 = SEB(features[-1], features[-2])
        self.seb2 = SEB(features[-2] + features[-1], features[-3])
        self.seb3 = SEB(features[-3] + features[-2] + features[-1], features[-4])
        self.upsample2 = nn.Upsample(scale_factor=2, mode="bilinear")
        self.upsample4 = nn.Upsample(scale_factor=4, mode="bilinear")
        self.DAP = nn.Sequential(nn.PixelShuffle(dap_k), nn.AvgPool2d((dap_k, dap_k)))
        self.decoder_frequency_0 = _freq_block(9, wide, 8)
        self.decoder_frequency_1 = _freq_block(wide, wide, 8)
        self.decoder_frequency_2 = _freq_block(wide, wide, 8)
        self.decoder_frequency_3 = _freq_block(wide, num_classes, 4)
        self.decoder_frequency_4 = _freq_block(num_classes, wide, 8)
        self._derived = {}

    def merge_views_along_channel_axis(self, tokens, height):
        """API parity with decoder.py:43-53: [(B,t,n,C_v)] -> (B, sum C, Tmax, h, h).  Not used by forward()."""
        tmax = max(self.input_token_temporal_dims)
        parts = []
        for v, x in enumerate(tokens):
            b, t, n, c = x.shape
            tv = self.input_token_temporal_dims[v]
            x = x.reshape(b, tv, (t * n) // tv, c)
            parts.append(x.repeat(1, tmax // tv, 1, 1))
        m = torch.cat(parts, dim=-1)
        b, t, n, c = m.shape
        return m.reshape(b, t, height, n // height, c).permute(0, 4, 1, 2, 3)

    # ------------------------------------------------------------------------------------------------ helpers
    def _cached(self, key, sources, fn):
        d = self._derived.get(key)
        if d is None:
            d = self._derived[key] = Derived()
        return d.get(sources, fn)

    def _conv(self, x, conv, residual=None):
        """nn.Conv2d (stride 1, same padding) as the hand-written implicit GEMM, NHWC in / NHWC out.  The (Cout,kh,kw,Cin)
        weight image is cached per weight version; input channels are zero-padded to a multiple of 32 (only the 9-channel
        frequency input needs it)."""
        cin = conv.weight.shape[1]
        pad = (-cin) % 32
        w = self._cached(("w", id(conv)), (conv.weight,),
                         lambda: F.pad(conv.weight.permute(0, 2, 3, 1), (0, pad)).contiguous())
        if pad and x.shape[1] != cin + pad:
            raise RuntimeError(f"Decoder._conv: a {cin}-channel convolution takes its input zero-padded to {cin + pad} channels "
                               "(ops.avgpool2_pad does it for the frequency branch)")
        return ops.conv2d_nhwc(x, w, conv.bias, residual=residual)

    def _gcm(self, m, x):                                              # x_l + x_r fused into the last conv's epilogue
        left = self._conv(self._conv(x, m.conv_l1), m.conv_l2)
        return self._conv(self._conv(x, m.conv_r1), m.conv_r2, residual=left)

    @staticmethod
    def _gn(x, gn):
        x, partial, nsplit = ops.gn_stats(x, gn.num_groups)
        return x, (partial, nsplit, gn.weight, gn.bias, gn.num_groups, gn.eps)

    def _rgb_head(self, idx, stage_views, side):
        """Conv3d(C', 256, k=s=(T,1,1)) + GroupNorm(16) + ReLU on the channel-merged, time-repeated views."""
        tmax = max(self.input_token_temporal_dims)
        seq = getattr(self, f"rgb_decoder_{idx + 1}")
        conv, gn = seq[0], seq[1]
        w = conv.weight                                               # (256, C1+C2+C3, T, 1, 1)
        y, c0 = None, 0
        for v, x in enumerate(stage_views):
            b, t, l, c = x.shape
            tv = self.input_token_temporal_dims[v]
            n = (t * l) // tv
            if tv == 1:       # same tokens at every time step -> one GEMM with the T weight slices summed
                wv = self._cached(("h", idx, v), (w,), lambda: w[:, c0:c0 + c, :, 0, 0].sum(2).contiguous())
                y = ops.linear(x.reshape(b * n, c), wv, conv.bias if y is None else None, residual=y)
            elif tv == tmax:  # ONE GEMM over all time slices: K index (t, c), the tokens' t-slices are K segments of the A rows
                wt = self._cached(("h", idx, v, "t"), (w,),
                                  lambda: w[:, c0:c0 + c, :, 0, 0].permute(0, 2, 1).reshape(w.shape[0], tv * c).contiguous())
                y = ops.linear_time_slices(x.reshape(b, tv, n, c), wt, conv.bias if y is None else None, residual=y)
            else:
                raise NotImplementedError("views must have temporal dim 1 or max (true for every Mumpy config)")
            c0 += c
        y = y.reshape(b, side, side, -1).permute(0, 3, 1, 2)          # logical NCHW over NHWC memory
        y, g = self._gn(y, gn)
        return ops.gn_apply_resample(y, g, act=ops.ACT_RELU)

    def _freq(self, seq, x, nchw_in=False):
        """AvgPool2d(2) -> conv3x3 -> GroupNorm -> Sigmoid (decoder.py:149-178).  The pooling kernel also pads the channels to
        the multiple of 32 the implicit-GEMM convolution wants and reads the FAF output in its NCHW layout."""
        cin = seq[1].weight.shape[1]
        x = self._conv(ops.avgpool2_pad(x, cin + (-cin) % 32, nchw_in=nchw_in), seq[1])
        x, g = self._gn(x, seq[2])
        return ops.gn_apply_resample(x, g, act=ops.ACT_SIGMOID)

    def _up(self, x, scale, out=None, out_coff=0):                     # nn.Upsample(bilinear, align_corners=False)
        return ops.gn_apply_resample(x, None, scale=scale, align_corners=False, out=out, out_coff=out_coff)

    def _seb(self, m, x1, x2):                                         # x1 * upsample(conv(x2))  (decoder.py:12-14)
        return ops.gn_apply_resample(self._conv(x2, m.conv), None, scale=2, align_corners=False, ep_mode=ops.EP_MUL, ep_a=x1)

    def _dec(self, seq, x, ep_mode=0, ep_a=None, ep_b=None, mean4=False):
        """conv3x3 -> GroupNorm(8) -> ReLU -> bilinear x2 (align_corners=True) [-> DAP] [-> epilogue]."""
        x, g = self._gn(self._conv(x, seq[0]), seq[1])
        return ops.gn_apply_resample(x, g, act=ops.ACT_RELU, mean4=mean4, scale=2, align_corners=True, ep_mode=ep_mode,
                                     ep_a=ep_a, ep_b=ep_b)

    def _branches(self, view_x, ffinfo):
        """Everything that does not need the encoder's final tokens: frequency pyramid, the four temporal heads and the
        SEB/GCM pyramids 2-4.  Independent chains of small kernels are forked onto side streams."""
        b, dev = ffinfo.shape[0], ffinfo.device

        def freq_chain():
            f0 = self._freq(self.decoder_frequency_0, ffinfo, nchw_in=ffinfo.is_contiguous())
            f1 = self._freq(self.decoder_frequency_1, f0)
            f2 = self._freq(self.decoder_frequency_2, f1)
            f3 = self._freq(self.decoder_frequency_3, f2)
            return [f0, f1, f2, f3, self._freq(self.decoder_frequency_4, f3)]

        flat = [t for stage in view_x for t in stage]
        (rgb3, rgb4), (rgb2,), freq, (rgb1,) = run_parallel(
            [lambda: [self._rgb_head(2, view_x[2], self.shape[2]), self._rgb_head(3, view_x[3], self.shape[3])],
             lambda: [self._rgb_head(1, view_x[1], self.shape[1])],
             freq_chain,
             lambda: [self._rgb_head(0, view_x[0], self.shape[0])]],
            [flat, flat, (ffinfo,), flat])

        def pyr1():
            return [self._gcm(self.gcm2, self._seb(self.seb1, rgb3, rgb4))]

        def pyr2():
            cat2 = ops.empty_nhwc(b, 512, 14, 14, dev)                 # [rgb3 | up2(rgb4)]          (decoder.py:210)
            ops.set_channels(cat2, 0, rgb3)
            self._up(rgb4, 2, out=cat2, out_coff=256)
            return [self._gcm(self.gcm3, self._seb(self.seb2, rgb2, cat2))]

        def pyr3():
            cat3 = ops.empty_nhwc(b, 768, 28, 28, dev)                 # [rgb2 | up2(rgb3) | up4(rgb4)] (decoder.py:213)
            ops.set_channels(cat3, 0, rgb2)
            self._up(rgb3, 2, out=cat3, out_coff=256)
            self._up(rgb4, 4, out=cat3, out_coff=512)
            return [self._gcm(self.gcm4, self._seb(self.seb3, rgb1, cat3))]

        deps = (rgb1, rgb2, rgb3, rgb4)
        (gcn1,), (gcn2,), (gcn3,) = run_parallel([pyr1, pyr2, pyr3], [deps, deps, deps])
        return {"rgb4": rgb4, "freq": freq, "gcn": (gcn1, gcn2, gcn3)}

    def _trunk(self, x, br):
        """gcm1 on [rgb4 | final tokens] and the sequential decoder_2..5 trunk -> x_feats (B,32,224,224), NHWC memory."""
        freq0, freq1, freq2, freq3, freq4 = br["freq"]
        gcn1, gcn2, gcn3 = br["gcn"]
        rgb4 = br["rgb4"]
        b, c4, h, w = rgb4.shape
        cat1 = ops.empty_nhwc(b, c4 + x.shape[1], h, w, rgb4.device)                        # [rgb4 | final tokens] (decoder.py:197)
        ops.set_channels(cat1, 0, rgb4)
        ops.set_channels(cat1, c4, x)                   # x: dense NHWC or the strided 3-of-T slice view of the global tokens
        # PixelShuffle(2)(gcm1(...) * freq4), then gcn1 * freq3 + that: one kernel (decoder.py:198-205)
        z = ops.trunk_head(self._gcm(self.gcm1, cat1), freq4, gcn1, freq3)
        z = self._dec(self.decoder_2, z, ops.EP_ADD_MUL, gcn2, freq2)                       # = decoder_3's input
        z = self._dec(self.decoder_3, z, ops.EP_ADD_MUL, gcn3, freq1)                       # = decoder_4's input
        z = self._dec(self.decoder_4, z, ops.EP_MUL, freq0)                                 # = decoder_5's input
        return self._dec(self.decoder_5, z, mean4=True)                                     # DAP folded in

    def _features(self, x, view_x, ffinfo):
        return self._trunk(x, self._branches(view_x, ffinfo))

    def _final_weight(self):
        return self._cached(("wf",), (self.final_out.weight,), lambda: self.final_out.weight.permute(0, 2, 3, 1).contiguous())

    def forward(self, x, view_x, ffinfo):
        """x (B,2304,7,7), view_x[4][3] of (B,1,L,C), ffinfo (B,9,224,224) -> (logits (B,1,224,224), feats (B,32,224,224))."""
        x_feats = self._features(x, view_x, ffinfo)
        return ops.final_conv(x_feats, self._final_weight(), self.final_out.bias), x_feats

    def predict_mask(self, x, view_x, ffinfo, thr=0.5):
        """forward() plus the eval tail of test.py:100-108 (sigmoid -> > thr -> uint8) emitted by the same last kernel:
        -> (logits (B,1,224,224), mask uint8 (B,1,224,224), feats)."""
        x_feats = self._features(x, view_x, ffinfo)
        logits, mask = ops.final_conv(x_feats, self._final_weight(), self.final_out.bias, with_mask=True, thr=thr)
        return logits, mask, x_feats


class BaselineDecoder(nn.Module):
    """Single-scale decoder of config 1 (decoder.py:228-284): five x (conv3x3 -> GroupNorm(32) -> ReLU -> bilinear x2,
    align_corners=True) and a 3x3 conv to the logits.  Same constructor / state_dict as the reference; note the reference
    builds every GroupNorm with `features[1]` channels (decoder.py:236-265), which is kept.  With `BaselineEncoder` the
    caller must pass `in_channels=1024` (the 2304 default only fits the three-view encoder).
    NHWC throughout: implicit-GEMM convs, GroupNorm + ReLU + upsample as one streaming kernel per block."""

    def __init__(self, in_channels=2304, out_channels=1, features=[256, 256, 256, 256, 256]):
        super().__init__()
        cin = in_channels
        for i in range(5):
            setattr(self, f"decoder_{i + 1}", nn.Sequential(
                nn.Conv2d(cin, features[i], 3, padding=1), nn.GroupNorm(32, features[1]), nn.ReLU(inplace=True),
                nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)))
            cin = features[i]
        self.final_out = nn.Conv2d(features[-1], out_channels, 3, padding=1)
        self._derived = {}

    def _cached(self, key, sources, fn):
        d = self._derived.get(key)
        if d is None:
            d = self._derived[key] = Derived()
        return d.get(sources, fn)

    def _block(self, seq, x):
        conv, gn = seq[0], seq[1]
        w = self._cached(("w", id(conv)), (conv.weight,), lambda: conv.weight.permute(0, 2, 3, 1).contiguous())
        y, partial, nsplit = ops.gn_stats(ops.conv2d_nhwc(x, w, conv.bias), gn.num_groups)
        return ops.gn_apply_resample(y, (partial, nsplit, gn.weight, gn.bias, gn.num_groups, gn.eps), act=ops.ACT_RELU,
                                     scale=2, align_corners=True)

    def forward(self, x):
        """x (B,in_channels,7,7) -> logits (B,out_channels=1,224,224)."""
        if self.final_out.out_channels != 1:
            raise NotImplementedError("BaselineDecoder: the HIP final conv emits one logit channel (the reference default)")
        x = x.contiguous(memory_format=torch.channels_last)
        for i in range(5):
            x = self._block(getattr(self, f"decoder_{i + 1}"), x)
        wf = self._cached(("wf",), (self.final_out.weight,), lambda: self.final_out.weight.permute(0, 2, 3, 1).contiguous())
        return ops.final_conv(x, wf, self.final_out.bias)
