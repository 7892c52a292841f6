"""Multi-pyramid decoder — drop-in for the reference's models/decoder/decoder.py (`Decoder`), same constructor
defaults and the same 92 state_dict keys (nn.Sequential indices included).

Round-1 status: the decoder is 8 % of the forward's FLOPs, all of it dense convolution.  Its graph is
restated here without the reference's intermediate lists; convolutions / GroupNorm / resampling run as PyTorch-ROCm
GPU ops (MIOpen) on the same stream — they are captured in the same hipGraph as the encoder's HIP kernels.  The
temporal Conv3d(k=(T,1,1)) heads are NOT convolutions at all (kernel = stride = full extent): they are per-pixel
Linears over the (C,T) axis and run on mumpy_linear_fwd straight from the encoder's token-major stage outputs, so the
(B,C',T,h,h) merged tensor of decoder.py:43-53 is never built.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from models.modules.layers import Derived
from mumpy_hip import ops


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
        self._rgbw = [Derived() for _ in range(4)]

    def merge_views_along_channel_axis(self, tokens, height):
        """API parity with decoder.py:43-53: [(B,t,n,C_v)] -> (B, sum C, Tmax, h, h)."""
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

    def _rgb_head(self, idx, stage_views, side):
        """Conv3d(C', 256, k=s=(T,1,1)) over the channel-merged views == per-pixel Linear with K = C'*T.
        Token-major operand X[b, n, (t, c)] is assembled directly from the three stage outputs."""
        tmax = max(self.input_token_temporal_dims)
        conv, gn = getattr(self, f"rgb_decoder_{idx + 1}")[0], getattr(self, f"rgb_decoder_{idx + 1}")[1]
        cols = []
        for v, x in enumerate(stage_views):
            b, t, n, c = x.shape
            tv = self.input_token_temporal_dims[v]
            x = x.reshape(b, tv, (t * n) // tv, c)
            cols.append(x.expand(b, tmax, x.shape[2], c) if tv == 1 else x.repeat(1, tmax // tv, 1, 1))
        m = torch.cat(cols, dim=-1)                                   # (B,T,n,C')
        b, t, n, c = m.shape
        xm = m.permute(0, 2, 1, 3).reshape(b * n, t * c)              # k = t*C' + c
        w = conv.weight                                               # (256, C', T, 1, 1)
        wk = self._rgbw[idx].get((w,), lambda: w.reshape(w.shape[0], c, t).permute(0, 2, 1).reshape(w.shape[0], t * c).contiguous())
        y = ops.linear(xm, wk, conv.bias)                             # (B*n, 256)
        y = y.reshape(b, side, side, -1).permute(0, 3, 1, 2)
        return F.relu(F.group_norm(y, gn.num_groups, gn.weight, gn.bias, gn.eps))

    def forward(self, x, view_x, ffinfo):
        """x (B,2304,7,7), view_x[4][3] of (B,1,L,C), ffinfo (B,9,224,224) -> (logits (B,1,224,224), feats (B,32,224,224))."""
        rgb1, rgb2, rgb3, rgb4 = [self._rgb_head(i, view_x[i], self.shape[i]) for i in range(4)]
        freq0 = self.decoder_frequency_0(ffinfo)
        freq1 = self.decoder_frequency_1(freq0)
        freq2 = self.decoder_frequency_2(freq1)
        freq3 = self.decoder_frequency_3(freq2)
        freq4 = self.decoder_frequency_4(freq3)
        out1 = self.ecre(self.gcm1(torch.cat([rgb4, x], dim=1)) * freq4)
        gcn1 = self.gcm2(self.seb1([rgb3, rgb4]))
        gcn2 = self.gcm3(self.seb2([rgb2, torch.cat([rgb3, self.upsample2(rgb4)], dim=1)]))
        gcn3 = self.gcm4(self.seb3([rgb1, torch.cat([rgb2, self.upsample2(rgb3), self.upsample4(rgb4)], dim=1)]))
        z = self.decoder_2(gcn1 * freq3 + out1)
        z = self.decoder_3(z + gcn2 * freq2)
        z = self.decoder_4(z + gcn3 * freq1)
        z = self.decoder_5(z * freq0)
        x_feats = self.DAP(z)
        return self.final_out(x_feats), x_feats
