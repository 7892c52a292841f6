#!/usr/bin/env python3
"""Every convolution launch of one B=8,T=5 forward (mumpy_conv2d_nhwc_fwd: the pyramid decoder), timed on its own:
shape, launch count, microseconds and TFLOP/s.  Run on the GPU box: python tools/conv_shapes.py"""
import collections, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]


def main():
    dev = torch.device("cuda:0")
    from mumpy_hip import ops
    from models.decoder.decoder import Decoder
    from models.encoder.encoder import Encoder
    from weight_fill import fill_module_, seeded_randn
    B, T = int(os.environ.get("B", "8")), int(os.environ.get("T", "5"))
    enc = fill_module_(Encoder(num_frames=T)).eval().to(dev)
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, T])).eval().to(dev)
    x = seeded_randn(1234, B, T, 3, 224, 224).to(dev)
    seen = collections.OrderedDict()
    real = ops.conv2d_nhwc

    def spy(xx, w, bias=None, act=ops.ACT_NONE, residual=None):
        key = (tuple(xx.shape), tuple(w.shape), bias is not None, int(act), residual is not None)
        seen[key] = seen.get(key, 0) + 1
        return real(xx, w, bias, act, residual)

    ops.conv2d_nhwc = spy
    with torch.no_grad():
        fx, vx, dx = enc(x)
        dec(fx, vx, dx)
    ops.conv2d_nhwc = real
    print(f"{'B':>3s} {'Cin':>5s} {'H':>4s} {'W':>4s} {'Cout':>5s} {'k':>2s} {'res':>3s} {'cnt':>3s} {'rows':>7s} {'K':>6s} {'us':>8s} {'TF':>6s} {'ms tot':>7s}")
    tot = tot_f = 0.0
    for (xs, ws, hb, act, hr), cnt in seen.items():
        b, cin, h, w = xs
        cout, kh, kw, _ = ws
        xx = ops.empty_nhwc(b, cin, h, w, dev).normal_()
        wt = torch.randn(cout, kh, kw, cin, device=dev) / (kh * kw * cin) ** 0.5
        bias = torch.randn(cout, device=dev) if hb else None
        res = ops.empty_nhwc(b, cout, h, w, dev).normal_() if hr else None
        for _ in range(3):
            real(xx, wt, bias, act, res)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            real(xx, wt, bias, act, res)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        fl = 2.0 * b * h * w * cout * kh * kw * cin
        print(f"{b:3d} {cin:5d} {h:4d} {w:4d} {cout:5d} {kh:2d} {int(hr):3d} {cnt:3d} {b*h*w:7d} {kh*kw*cin:6d} {us:8.1f} {fl/us/1e6:6.1f} {cnt*us/1e3:7.2f}")
        tot += cnt * us / 1e3; tot_f += cnt * fl
    print(f"TOTAL {tot:.2f} ms, {tot_f/1e9:.1f} GFLOP, {tot_f/tot/1e9:.1f} TF average")


if __name__ == "__main__":
    main()
