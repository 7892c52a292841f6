"""Long-temporal golden (config 4's T = 9 at 224x224) from the REAL reference, on CPU, in the build container.

    python tests/golden/gen_goldens_t9.py [--ref /root/reference] [--out tests/golden]

Same rules as gen_goldens.py (whose stand-ins and patches it reuses): the reference's classes run unmodified with
view configs for tubelets (9, 8, 1) and temporal dims [1, 1, 9] -- the construction SURVEY 8d verified for T = 9 --
weights from weight_fill.py by state_dict key, input seeded.  Writes full_model_t9.npz (logits, final_x, digests) and the T = 9 state_dict manifests.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402
from weight_fill import fill_module_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()
    G.install_stubs()
    sys.path.insert(0, args.ref)
    torch.Tensor.cuda = lambda self, *a, **k: self          # dct.py:16,18,61,62
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    import ml_collections
    import models.factory.modelFactory as factory
    factory.load_model_weights = lambda model, path, strict=False: model      # factory:70-71
    from models.decoder.decoder import Decoder
    from models.encoder.multiTemporalViewEncoder import ThreeViewSwinTransformer

    T = 9
    cvc = factory.create_view_config
    res = [(56, 56), (28, 28), (14, 14), (7, 7)]
    vcs = [cvc([96, 192, 384, 768], (4, 4, T), [2, 2, 6, 2], [3, 6, 12, 24], 768, 1, res, 1, [1, 1]),
           cvc([96, 192, 384, 768], (4, 4, T - 1), [2, 2, 18, 2], [3, 6, 12, 24], 1536, 1, res, 1, [1, T]),
           cvc([128, 256, 512, 1024], (4, 4, 1), [2, 2, 18, 2], [4, 8, 16, 32], 3072, T, res, T)]
    gcfg = ml_collections.ConfigDict({'num_heads': 12, 'mlp_dim': 3072, 'num_layers': 12, 'hidden_size': 768,
                                      'merge_axis': 'channel', 'num_frames': T})
    model = ThreeViewSwinTransformer(view_configs=vcs, input_token_temporal_dims=[1, 1, T], global_encoder_config=gcfg).eval()

    class Enc(torch.nn.Module):           # same wrapper arithmetic as encoder.py:11-18
        def __init__(self):
            super().__init__()
            self.base = model

        def forward(self, x):
            fx, vx, dx = self.base(x)
            return fx.reshape(fx.shape[0], 7, 7, 2304).permute(0, 3, 1, 2), vx, dx

    enc = fill_module_(Enc().eval())
    dec = fill_module_(Decoder(input_token_temporal_dims=[1, 1, T]).eval())
    import json
    with open(os.path.join(args.out, "state_dict_encoder_t9.json"), "w") as f:
        json.dump(G.sd_manifest(enc), f)
    with open(os.path.join(args.out, "state_dict_decoder_t9.json"), "w") as f:
        json.dump(G.sd_manifest(dec), f)
    store = {}
    x = G.inp(store, "b1t9/x", 1239, 1, T, 3, 224, 224)
    fx, vx, dx = enc(x)
    logits, feats = dec(fx, vx, dx)
    store["b1t9/logits"] = logits.numpy().astype(np.float32)
    store["b1t9/final_x"] = fx.numpy().astype(np.float32)
    G.put_digest(store, "b1t9/dct_x", dx)
    G.put_digest(store, "b1t9/x_feats", feats)
    for s in range(4):
        for v in range(3):
            store[f"b1t9/view_shape_{s}_{v}"] = np.array(vx[s][v].shape, dtype=np.int32)
            G.put_digest(store, f"b1t9/view_{s}_{v}", vx[s][v])
    print("b1t9 logits", tuple(logits.shape), float(logits.abs().max()), "final_x", float(fx.abs().max()))
    np.savez_compressed(os.path.join(args.out, "full_model_t9.npz"), **{k: np.ascontiguousarray(v) for k, v in store.items()})


if __name__ == "__main__":
    main()
