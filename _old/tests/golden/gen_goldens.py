"""Generate the committed parity fixtures from the REAL reference, on CPU, in the build container.

    python tests/golden/gen_goldens.py [--ref /root/reference] [--out tests/golden]

This is the only place the reference's Python is ever imported.  It never travels to the
GPU box: tests read only the small .npz/.json files this script writes.

The reference needs two third-party packages that are absent offline; the three symbols of
`timm.models.layers` it touches and `ml_collections.ConfigDict` are stood in for by throw-away
modules written to a temp dir (own code, a few lines each; eval-mode DropPath is the identity).
`torch.Tensor.cuda` is made a no-op because dct.py calls `.cuda()` in constructors, and
`modelFactory.load_model_weights` is made a no-op because the weight file is not available.

Weights: tests/golden/weight_fill.py (deterministic by state_dict key).  Inputs: seeded randn.
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from weight_fill import digest, fill_module_, seeded_randn  # noqa: E402

STUB_TIMM = '''
import collections.abc, torch, torch.nn as nn
def to_2tuple(x):
    return tuple(x) if isinstance(x, collections.abc.Iterable) and not isinstance(x, str) else (x, x)
def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)
class DropPath(nn.Module):
    def __init__(self, p=0.):
        super().__init__(); self.p = p
    def forward(self, x):
        assert not self.training, "golden generation is eval-only"
        return x
'''

STUB_MLC = '''
class ConfigDict(dict):
    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = ConfigDict(v) if isinstance(v, dict) else v
    def __getattr__(self, k):
        try: return self[k]
        except KeyError as e: raise AttributeError(k) from e
    def __setattr__(self, k, v): self[k] = v
'''


def install_stubs():
    d = tempfile.mkdtemp(prefix="mumpy_stubs_")
    os.makedirs(os.path.join(d, "timm", "models"))
    open(os.path.join(d, "timm", "__init__.py"), "w").close()
    open(os.path.join(d, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "models", "layers.py"), "w") as f:
        f.write(STUB_TIMM)
    with open(os.path.join(d, "ml_collections.py"), "w") as f:
        f.write(STUB_MLC)
    sys.path.insert(0, d)


def sd_manifest(m):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}


def inp(store, name, seed, *shape):
    """Seeded input: only (seed, shape) is stored; tests regenerate it with weight_fill.seeded_randn."""
    store[name + "/seed_shape"] = np.array([seed, *shape], dtype=np.int64)
    return seeded_randn(seed, *shape)


def put_digest(store, name, t):
    s, v = digest(t)
    store[name + "/stats"] = s
    store[name + "/samples"] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=HERE)
    args = ap.parse_args()

    install_stubs()
    sys.path.insert(0, args.ref)
    torch.Tensor.cuda = lambda self, *a, **k: self          # dct.py:16,18,61,62
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.manual_seed(0)
    torch.set_grad_enabled(False)

    import models.factory.modelFactory as factory
    factory.load_model_weights = lambda model, path, strict=False: model      # factory:70-71
    from models.decoder.decoder import Decoder
    from models.encoder.encoder import BaselineEncoder, Encoder
    from models.encoder.multiTemporalViewEncoder import (CrossSwinBlock, CrossThreeViewTokenize,
                                                         ThreeViewSwinTransformer)
    from models.modules.blocks import Block
    from models.modules.dct import FAF
    from models.modules.deformableAttention import SwinDAttention
    from models.modules.swinTransformer import (PatchMerging, SwinTransformerBlock, WindowAttention,
                                                window_partition, window_reverse)

    out = args.out
    f32 = np.float32

    # ------------------------------------------------------------------ 1. state_dict manifests
    enc = Encoder().eval()
    dec = Decoder().eval()
    with open(os.path.join(out, "state_dict_encoder.json"), "w") as f:
        json.dump(sd_manifest(enc), f)
    with open(os.path.join(out, "state_dict_decoder.json"), "w") as f:
        json.dump(sd_manifest(dec), f)
    print("encoder keys", len(enc.state_dict()), "decoder keys", len(dec.state_dict()))

    # ------------------------------------------------------------------ 2. integer index maps
    idx = {}
    for (hs, w) in [(56, 56), (168, 56), (280, 56), (42, 14)]:
        ramp = torch.arange(hs * w, dtype=torch.int64).view(1, hs, w, 1)
        idx[f"part_{hs}x{w}"] = window_partition(ramp, 7).reshape(-1).to(torch.int32).numpy()
        rolled = torch.roll(ramp, shifts=(-3, -3), dims=(1, 2))
        idx[f"rollpart_{hs}x{w}"] = window_partition(rolled, 7).reshape(-1).to(torch.int32).numpy()
        # reverse path: windows -> window_reverse -> roll(+3): position j of the result holds ramp value ?
        wins = torch.arange(hs * w, dtype=torch.int64).view(-1, 7, 7, 1)
        back = torch.roll(window_reverse(wins, 7, hs, w), shifts=(3, 3), dims=(1, 2))
        idx[f"revroll_{hs}x{w}"] = back.reshape(-1).to(torch.int32).numpy()
    for (res, t) in [(56, 1), (56, 3), (56, 5), (14, 3), (28, 5)]:
        blk = SwinTransformerBlock(dim=32, input_resolution=(res, res), num_heads=1, window_size=7,
                                   shift_size=3, temporal_dim=t)
        idx[f"mask_{res}_t{t}"] = np.packbits((blk.attn_mask != 0).numpy().reshape(-1))
        idx[f"mask_{res}_t{t}_shape"] = np.array(blk.attn_mask.shape, dtype=np.int32)
    wa = WindowAttention(32, (7, 7), 1)
    idx["relative_position_index"] = wa.relative_position_index.to(torch.int32).numpy()
    np.savez_compressed(os.path.join(out, "index_maps.npz"), **idx)

    # ------------------------------------------------------------------ 3. per-operator goldens
    ops = {}

    # 3a WindowAttention (swin:134-166): 8 windows, C=96, 3 heads; with and without mask (nW=4)
    wa = fill_module_(WindowAttention(96, (7, 7), 3).eval(), "wa/")
    x = inp(ops, "wa/x", 11, 8, 49, 96)
    blk = SwinTransformerBlock(dim=96, input_resolution=(14, 14), num_heads=3, shift_size=3)
    ops["wa/y_nomask"] = wa(x).numpy()
    ops["wa/mask"] = blk.attn_mask.numpy()
    ops["wa/y_mask"] = wa(x, mask=blk.attn_mask).numpy()

    # 3b SwinTransformerBlock (swin:259-307): shifted, frames stacked on rows (temporal_dim=3), B=2
    for tag, shift, t in [("stb_s3_t3", 3, 3), ("stb_s0_t1", 0, 1)]:
        blk = fill_module_(SwinTransformerBlock(dim=96, input_resolution=(14, 14), num_heads=3,
                                                shift_size=shift, temporal_dim=t).eval(), tag + "/")
        x = inp(ops, tag + "/x", 12, 2, t * 196, 96)
        ops[tag + "/y"] = blk(x).numpy()

    # 3c SwinDAttention (deform:324-405): B1 = 4 q-windows, ratio r in {1,3,5}
    for r in (1, 3, 5):
        tag = f"sda_r{r}"
        m = fill_module_(SwinDAttention(96, 3, 0.0, n_groups=3).eval(), tag + "/")
        x1 = inp(ops, tag + "/x1", 13, 4, 49, 96)
        x2 = inp(ops, tag + "/x2", 14 + r, 4 * r, 49, 96)
        y, attn = m(x1, x2)
        ops[tag + "/y"] = y.numpy()
        put_digest(ops, tag + "/attn", attn)

    # 3d CrossSwinBlock (mTVE:228-291): view2 <- view3 (dim 96 <- 128, T2 = 3) and last_view
    m = fill_module_(CrossSwinBlock(96, 128, (14, 14), 3, temporal_dims=1).eval(), "csb/")
    x1 = inp(ops, "csb/x1", 21, 2, 196, 96)
    x2 = inp(ops, "csb/x2", 22, 2, 3 * 196, 128)
    y, o = m(x1, x2)
    ops["csb/y"], ops["csb/out"] = y.numpy(), o.numpy()
    m = fill_module_(CrossSwinBlock(128, 128, (14, 14), 4, last_view=True, temporal_dims=3).eval(), "csbl/")
    x1 = inp(ops, "csbl/x1", 23, 2, 3 * 196, 128)
    y, o = m(x1, x1)
    ops["csbl/y"], ops["csbl/out"] = y.numpy(), o.numpy()

    # 3e PatchMerging (swin:344-367) on a stacked (42,14) grid
    m = fill_module_(PatchMerging((42, 14), 96).eval(), "pm/")
    x = inp(ops, "pm/x", 31, 2, 42 * 14, 96)
    ops["pm/y"] = m(x).numpy()

    # 3f FAF (dct:71-79): one clip, keep frame index 1 (mTVE:734); store strided + a band
    faf = FAF()
    x = inp(ops, "faf/x", 41, 1, 3, 3, 224, 224)
    y = faf(x)[:, 1]
    ops["faf/y_sub4"] = y[:, :, ::4, ::4].contiguous().numpy()
    ops["faf/y_rows"] = y[:, :, 100:104, :].contiguous().numpy()
    put_digest(ops, "faf/y", y)
    ops["faf/dct_row5"] = faf._DCT_all[5].numpy()

    # 3g global ViT Block (blocks:77-92) on (2,3,768)
    m = fill_module_(Block(768, 12, 3072, 0.0, 0.0).eval(), "gb/")
    x = inp(ops, "gb/x", 51, 2, 3, 768)
    ops["gb/y"] = m(x).numpy()

    # 3h tokenizer (mTVE:574-618), default configs
    base = enc.base
    tk = fill_module_(CrossThreeViewTokenize(enc.configs).eval(), "tok/")
    x = inp(ops, "tok/x", 61, 1, 3, 3, 224, 224)
    for i, t in enumerate(tk(x)):
        ops[f"tok/shape{i}"] = np.array(t.shape, dtype=np.int32)
        put_digest(ops, f"tok/y{i}", t)
        ops[f"tok/y{i}_head"] = t.reshape(-1, t.shape[-1])[:64].numpy()
    np.savez_compressed(os.path.join(out, "ops.npz"), **{k: np.ascontiguousarray(v) for k, v in ops.items()})

    # ------------------------------------------------------------------ 4. whole-model goldens
    def run_full(tag, encm, decm, seed_shape, store):
        x = inp(store, tag + "/x", *seed_shape)
        fx, vx, dx = encm(x)
        logits, feats = decm(fx, vx, dx)
        store[tag + "/logits"] = logits.numpy().astype(f32)
        store[tag + "/final_x"] = fx.numpy().astype(f32)
        put_digest(store, tag + "/dct_x", dx)
        put_digest(store, tag + "/x_feats", feats)
        for s in range(4):
            for v in range(3):
                store[f"{tag}/view_shape_{s}_{v}"] = np.array(vx[s][v].shape, dtype=np.int32)
                put_digest(store, f"{tag}/view_{s}_{v}", vx[s][v])
        print(tag, "logits", tuple(logits.shape), float(logits.abs().max()), "final_x", float(fx.abs().max()))

    full = {}
    fill_module_(enc)
    fill_module_(dec)
    run_full("b1t3", enc, dec, (1234, 1, 3, 3, 224, 224), full)
    run_full("b2t3", enc, dec, (1235, 2, 3, 3, 224, 224), full)       # pins cross-sample coupling
    del enc, dec

    # T=5: tubelets (5,4,1), temporal dims [1,1,5]  (SURVEY 8d config 2 construction)
    cvc = factory.create_view_config
    res = [(56, 56), (28, 28), (14, 14), (7, 7)]
    vcs = [cvc([96, 192, 384, 768], (4, 4, 5), [2, 2, 6, 2], [3, 6, 12, 24], 768, 1, res, 1, [1, 1]),
           cvc([96, 192, 384, 768], (4, 4, 4), [2, 2, 18, 2], [3, 6, 12, 24], 1536, 1, res, 1, [1, 5]),
           cvc([128, 256, 512, 1024], (4, 4, 1), [2, 2, 18, 2], [4, 8, 16, 32], 3072, 5, res, 5)]
    import ml_collections
    gcfg = ml_collections.ConfigDict({'num_heads': 12, 'mlp_dim': 3072, 'num_layers': 12, 'hidden_size': 768,
                                      'merge_axis': 'channel', 'num_frames': 5})
    model5 = ThreeViewSwinTransformer(view_configs=vcs, input_token_temporal_dims=[1, 1, 5],
                                      global_encoder_config=gcfg).eval()

    class Enc5(torch.nn.Module):          # same wrapper arithmetic as encoder.py:11-18
        def __init__(self):
            super().__init__()
            self.base = model5

        def forward(self, x):
            fx, vx, dx = self.base(x)
            b = fx.shape[0]
            return fx.reshape(b, 7, 7, 2304).permute(0, 3, 1, 2), vx, dx

    enc5 = fill_module_(Enc5().eval())
    dec5 = fill_module_(Decoder(input_token_temporal_dims=[1, 1, 5]).eval())
    with open(os.path.join(out, "state_dict_encoder_t5.json"), "w") as f:
        json.dump(sd_manifest(enc5), f)
    with open(os.path.join(out, "state_dict_decoder_t5.json"), "w") as f:
        json.dump(sd_manifest(dec5), f)
    run_full("b1t5", enc5, dec5, (1236, 1, 5, 3, 224, 224), full)
    del enc5, dec5, model5

    # config 1: single-scale baseline encoder (factory:76-93), B=1,T=3
    benc = BaselineEncoder().eval()
    with open(os.path.join(out, "state_dict_baseline_encoder.json"), "w") as f:
        json.dump(sd_manifest(benc), f)
    fill_module_(benc)
    y = benc(inp(full, "base_b1t3/x", 1237, 1, 3, 3, 224, 224))
    full["base_b1t3/y"] = y.numpy().astype(f32)
    print("baseline", tuple(y.shape), float(y.abs().max()))
    # config 1 decoder: BaselineDecoder(in_channels=1024) on the baseline encoder's output (decoder.py:228-284)
    from models.decoder.decoder import BaselineDecoder
    bdec = BaselineDecoder(in_channels=1024).eval()
    with open(os.path.join(out, "state_dict_baseline_decoder.json"), "w") as f:
        json.dump(sd_manifest(bdec), f)
    fill_module_(bdec)
    z = bdec(y)
    full["base_b1t3/logits"] = z.numpy().astype(f32)
    print("baseline decoder", tuple(z.shape), float(z.abs().max()))
    np.savez_compressed(os.path.join(out, "full_model.npz"), **{k: np.ascontiguousarray(v) for k, v in full.items()})

    for fn in sorted(os.listdir(out)):
        if fn.endswith((".npz", ".json")):
            print(f"{fn}: {os.path.getsize(os.path.join(out, fn)) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
