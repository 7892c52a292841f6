import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from weight_fill import fill_module_, seeded_randn
from models.encoder.encoder import Encoder
from models.decoder.decoder import Decoder
dev = torch.device("cuda:0")
enc, dec = fill_module_(Encoder().eval()).to(dev), fill_module_(Decoder().eval()).to(dev)
x = seeded_randn(78, 1, 3, 3, 224, 224).to(dev)
def run():
    with torch.no_grad():
        fx, vx, dx = enc(x)
        lg, ft = dec(fx, vx, dx)
    return fx.clone(), [v.clone() for s in vx for v in s], dx.clone(), lg.clone()
a, b = run(), run()
print("eager determinism: fx", torch.equal(a[0], b[0]), "views", all(torch.equal(p, q) for p, q in zip(a[1], b[1])),
      "dct", torch.equal(a[2], b[2]), "logits", torch.equal(a[3], b[3]))
# graph over encoder only
static_x = x.clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    fx, vx, dx = enc(static_x)
    lg, ft = dec(fx, vx, dx)
g.replay(); torch.cuda.synchronize()
vl = [v for s_ in vx for v in s_]
print("graph vs eager: fx", torch.equal(fx, a[0]), "views", [torch.equal(p, q) for p, q in zip(vl, a[1])], "dct", torch.equal(dx, a[2]),
      "logits", torch.equal(lg, a[3]), float((lg - a[3]).abs().max()))
g.replay(); torch.cuda.synchronize()
print("replay2 logits", torch.equal(lg, a[3]), float((lg - a[3]).abs().max()), "fx", float((fx - a[0]).abs().max()))
