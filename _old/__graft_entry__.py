"""Driver entry points: build() compiles every HIP source for gfx950 in-tree; smoke() runs one small forward of the
hot path on cuda:0 through the C ABI and checks it against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build() -> None:
    """hipcc --offload-arch=gfx950 for every kernel -> <pkg>/lib/libmumpy_hip.so (cross-compiles without a GPU).
    The oracle is pure torch/numpy (the reference is pure Python: nothing to compile, no oracle/_ref)."""
    from mumpy_hip.lib import build_library, load_library
    path = build_library()
    load_library()
    import models.decoder.decoder  # noqa: F401
    import models.encoder.encoder  # noqa: F401
    print(f"built and loaded {path}")


def smoke() -> None:
    import torch
    from models.decoder.decoder import Decoder
    from models.encoder.encoder import Encoder
    from oracle import mumpy_oracle as O
    from weight_fill import fill_module_, seeded_randn
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    dev = torch.device("cuda:0")
    enc, dec = fill_module_(Encoder().eval()), fill_module_(Decoder().eval())
    x = seeded_randn(4321, 1, 3, 3, 224, 224)
    sde = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    sdd = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    enc, dec = enc.to(dev), dec.to(dev)
    with torch.no_grad():
        fx, vx, dx = enc(x.to(dev))
        logits, _ = dec(fx, vx, dx)
        ref = O.full_forward(sde, sdd, x)[0]
    err = float((logits.cpu().double() - ref.double()).abs().max() / ref.double().abs().max())
    print(f"smoke: B=1,T=3 full forward on {torch.cuda.get_device_name(0)}: rel err vs oracle = {err:.3e}")
    assert err < 1e-3, "HIP forward disagrees with the oracle"


if __name__ == "__main__":
    build()
    if len(sys.argv) > 1 and sys.argv[1] == "smoke":
        smoke()
