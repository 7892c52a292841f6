#!/usr/bin/env python3
"""Phase-level timeline of the fused forward (events on the main stream; side streams join at phase ends)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
from mumpy_hip import ops
from mumpy_hip.streams import run_parallel
dev = torch.device("cuda:0")
T = 5
enc = fill_module_(Encoder(num_frames=T).eval()).to(dev)
dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, T]).eval()).to(dev)
x = seeded_randn(1, 8, T, 3, 224, 224).to(dev)
base = enc.base
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
@torch.no_grad()
def run():
    marks.clear()
    mark("start")
    toks = base.tokenize(x); mark("tokenize")
    ff = base.faf.forward_frame(x, 1); mark("faf")
    xs = toks
    outs = []
    for s, layer in enumerate(base.layers.layers):
        xs = layer.blocks[0](xs); mark(f"stage{s} cross")
        res = run_parallel([lambda: layer._view_chain(0, xs[0]), lambda: layer._view_chain(1, xs[1]), lambda: layer._view_chain(2, xs[2])],
                           [(xs[0],), (xs[1],), (xs[2],)])
        xs = [r[0] for r in res]; outs.append([r[1].unsqueeze(1) for r in res]); mark(f"stage{s} chains")
    flat = [t for st in outs for t in st]
    (tokens,), br = run_parallel([lambda: (base.forward_global(xs),), lambda: dec._branches(outs, ff)], [xs, flat + [ff]])
    mark("global || decoder branches")
    b, _, c = tokens.shape
    feats = dec._trunk(tokens.reshape(b, 7, 7, c).permute(0, 3, 1, 2), br); mark("decoder trunk")
    ops.final_conv(feats, dec._final_weight(), dec.final_out.bias, with_mask=True); mark("final conv + mask")
for _ in range(3):
    run()
torch.cuda.synchronize()
run(); torch.cuda.synchronize()
tot = marks[0][1].elapsed_time(marks[-1][1])
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    print(f"{n1:32s} {e0.elapsed_time(e1):7.3f} ms")
print(f"{'TOTAL':32s} {tot:7.3f} ms")
