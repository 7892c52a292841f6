#!/usr/bin/env python3
"""Build container only (needs /root/reference): time the REFERENCE's own CPU forward and the oracle (its CPU restatement)
on the same B=8, T=5 batch, and write tests/golden/cpu_ref_ratio.json -- the factor by which bench.py's `cpu_baseline`
(kind "port": the oracle, because the reference cannot travel to the GPU box) over- or under-states the reference's CPU path
(SURVEY 8d asks for +-20 %).  The reference is imported exactly as tests/golden/gen_goldens.py does.

    PYTHONPYCACHEPREFIX=/tmp/pycache python tools/cpu_ref_ratio.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen_goldens as G                                                    # noqa: E402
from weight_fill import fill_module_, seeded_randn                        # noqa: E402


def main(ref="/root/reference"):
    G.install_stubs()
    sys.path.insert(0, ref)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.set_grad_enabled(False)
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    import ml_collections
    import models.factory.modelFactory as factory
    factory.load_model_weights = lambda model, path, strict=False: model
    from models.decoder.decoder import Decoder
    from models.encoder.multiTemporalViewEncoder import ThreeViewSwinTransformer
    cvc = factory.create_view_config
    res = [(56, 56), (28, 28), (14, 14), (7, 7)]
    vcs = [cvc([96, 192, 384, 768], (4, 4, 5), [2, 2, 6, 2], [3, 6, 12, 24], 768, 1, res, 1, [1, 1]),
           cvc([96, 192, 384, 768], (4, 4, 4), [2, 2, 18, 2], [3, 6, 12, 24], 1536, 1, res, 1, [1, 5]),
           cvc([128, 256, 512, 1024], (4, 4, 1), [2, 2, 18, 2], [4, 8, 16, 32], 3072, 5, res, 5)]
    gcfg = ml_collections.ConfigDict({'num_heads': 12, 'mlp_dim': 3072, 'num_layers': 12, 'hidden_size': 768,
                                      'merge_axis': 'channel', 'num_frames': 5})
    model5 = ThreeViewSwinTransformer(view_configs=vcs, input_token_temporal_dims=[1, 1, 5], global_encoder_config=gcfg).eval()

    class Enc5(torch.nn.Module):                      # same wrapper arithmetic as encoder.py:11-18
        def __init__(self):
            super().__init__()
            self.base = model5

        def forward(self, x):
            fx, vx, dx = self.base(x)
            return fx.reshape(fx.shape[0], 7, 7, 2304).permute(0, 3, 1, 2), vx, dx

    enc5 = fill_module_(Enc5().eval())
    dec5 = fill_module_(Decoder(input_token_temporal_dims=[1, 1, 5]).eval())
    x = seeded_randn(1234, 8, 5, 3, 224, 224)

    def ref_fwd(xx):
        fx, vx, dx = enc5(xx)
        return dec5(fx, vx, dx)[0]
    ref_fwd(x[:1])
    tr = []
    for _ in range(3):
        t0 = time.perf_counter()
        yr = ref_fwd(x)
        tr.append(time.perf_counter() - t0)
    sde = {k: v.detach().clone() for k, v in enc5.state_dict().items()}
    sdd = {k: v.detach().clone() for k, v in dec5.state_dict().items()}
    for m in [k for k in sys.modules if k == 'models' or k.startswith('models.')]:
        del sys.modules[m]
    sys.path.remove(ref)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
    from oracle import mumpy_oracle as O
    O.full_forward(sde, sdd, x[:1])
    to = []
    for _ in range(3):
        t0 = time.perf_counter()
        yo = O.full_forward(sde, sdd, x)[0]
        to.append(time.perf_counter() - t0)
    out = {"what": f"B=8,T=5,224x224 fp32 forward on the build container ({os.cpu_count()} vCPU, {threads} torch threads), best of 3 "
                   "after a B=1 warm-up; reference = /root/reference classes imported as tests/golden/gen_goldens.py does, oracle = "
                   "oracle/mumpy_oracle.py",
           "reference_s_per_pass": round(min(tr), 3), "oracle_s_per_pass": round(min(to), 3),
           "reference_clips_s": round(8 / min(tr), 3), "oracle_clips_s": round(8 / min(to), 3),
           "oracle_over_reference": round(min(tr) / min(to), 3),
           "max_rel_diff_logits": float((yo - yr).abs().max() / yr.abs().max())}
    print(json.dumps(out))
    with open(os.path.join(ROOT, "tests", "golden", "cpu_ref_ratio.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:2])
