"""CPU: the oracle (oracle/mumpy_oracle.py) against the fixtures generated from the real reference.
This is what pins the oracle; the GPU tests then compare the HIP path with the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_digest, golden_input, rel_err
from oracle import mumpy_oracle as O
from weight_fill import fill_tensor_

TOL = 2e-5      # oracle-vs-reference: same ATen CPU kernels, different op order -> fp32 round-off only


def synth_sd(manifest_file, prefix_filter=None, salt=""):
    """Build a state_dict from a committed manifest and fill it deterministically."""
    man = json.load(open(os.path.join(GOLDEN, manifest_file)))
    sd = {}
    for k, (shape, dt) in man.items():
        t = torch.zeros(shape, dtype=getattr(torch, dt))
        sd[k] = t
    return man, sd


def local_sd(spec, salt):
    """spec: {name: shape}; returns filled float tensors (per-operator goldens use local key names)."""
    sd = {}
    for k, shape in spec.items():
        t = torch.zeros(shape)
        fill_tensor_(k, t, salt)
        sd[k] = t
    return sd


def rel_index():
    c = torch.stack(torch.meshgrid(torch.arange(7), torch.arange(7), indexing="ij")).flatten(1)
    r = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + 6
    return r[..., 0] * 13 + r[..., 1]


def wa_spec(c, nh, pre=""):
    return {pre + "relative_position_bias_table": (169, nh), pre + "qkv.weight": (3 * c, c), pre + "qkv.bias": (3 * c,),
            pre + "proj.weight": (c, c), pre + "proj.bias": (c,)}


def block_spec(c, nh):
    s = {"norm1.weight": (c,), "norm1.bias": (c,), "norm2.weight": (c,), "norm2.bias": (c,),
         "mlp.fc1.weight": (4 * c, c), "mlp.fc1.bias": (4 * c,), "mlp.fc2.weight": (c, 4 * c), "mlp.fc2.bias": (c,)}
    s.update(wa_spec(c, nh, "attn."))
    return s


def sda_spec(c, pre=""):
    cg = c // 3
    s = {pre + "conv_offset.0.weight": (cg, 1, 5, 5), pre + "conv_offset.0.bias": (cg,),
         pre + "conv_offset.1.norm.weight": (cg,), pre + "conv_offset.1.norm.bias": (cg,),
         pre + "conv_offset.3.weight": (2, cg, 1, 1)}
    for n in ("q", "k", "v", "out"):
        s[pre + f"proj_{n}.weight"] = (c, c, 1, 1)
        s[pre + f"proj_{n}.bias"] = (c,)
    return s


# ------------------------------------------------------------------ integer index maps: bit exact
@pytest.mark.parametrize("hs,w", [(56, 56), (168, 56), (280, 56), (42, 14)])
def test_window_index_bit_exact(index_golden, hs, w):
    assert np.array_equal(O.window_token_index(hs, w, 0).numpy(), index_golden[f"part_{hs}x{w}"])
    assert np.array_equal(O.window_token_index(hs, w, 3).numpy(), index_golden[f"rollpart_{hs}x{w}"])
    # reverse: scatter windows 0..N-1 through the same map
    idx = O.window_token_index(hs, w, 3)
    back = torch.empty(hs * w, dtype=torch.int64)
    back[idx] = torch.arange(hs * w)
    assert np.array_equal(back.numpy(), index_golden[f"revroll_{hs}x{w}"])


@pytest.mark.parametrize("res,t", [(56, 1), (56, 3), (56, 5), (14, 3), (28, 5)])
def test_shift_mask_bit_exact(index_golden, res, t):
    m = O.shift_attn_mask(res * t, res, 3)
    assert list(m.shape) == list(index_golden[f"mask_{res}_t{t}_shape"])
    assert np.array_equal(np.packbits((m != 0).numpy().reshape(-1)), index_golden[f"mask_{res}_t{t}"])
    assert set(m.unique().tolist()) <= {0.0, -100.0}


def test_relative_position_index(index_golden):
    assert np.array_equal(rel_index().numpy(), index_golden["relative_position_index"])


# ------------------------------------------------------------------ per-operator parity
def test_window_attention(ops_golden):
    sd = local_sd(wa_spec(96, 3), "wa/")
    sd["relative_position_index"] = rel_index()
    x = golden_input(ops_golden, "wa/x")                       # (8,49,96) windows; treat as (B=8, 7x7 grid)
    sdp = {"a." + k: v for k, v in sd.items()}
    y = O.window_attention(x, sdp, "a", 7, 7, 0, None)
    assert rel_err(y, ops_golden["wa/y_nomask"]) < TOL
    # with mask: 8 windows = 2 images x nW=4, image grid (14,14) shift 3; feed in window order via identity map
    mask = torch.tensor(ops_golden["wa/mask"])
    assert torch.equal(mask, O.shift_attn_mask(14, 14, 3))
    idx = O.window_token_index(14, 14, 3)
    xr = torch.empty(2, 196, 96)
    xr[:, idx] = x.reshape(2, 196, 96)                         # raster image whose shifted windows are x
    y = O.window_attention(xr, sdp, "a", 14, 14, 3, mask)
    assert rel_err(y[:, idx].reshape(8, 49, 96), ops_golden["wa/y_mask"]) < TOL


@pytest.mark.parametrize("tag,shift,t", [("stb_s3_t3", 3, 3), ("stb_s0_t1", 0, 1)])
def test_swin_block(ops_golden, tag, shift, t):
    sd = local_sd(block_spec(96, 3), tag + "/")
    sd["attn.relative_position_index"] = rel_index()
    sd = {"b." + k: v for k, v in sd.items()}
    x = golden_input(ops_golden, tag + "/x")
    y = O.swin_block(x, sd, "b", 14 * t, 14, shift)
    assert rel_err(y, ops_golden[tag + "/y"]) < TOL


@pytest.mark.parametrize("r", [1, 3, 5])
def test_swin_dattention(ops_golden, r):
    tag = f"sda_r{r}"
    sd = {"d." + k: v for k, v in local_sd(sda_spec(96), tag + "/").items()}
    x1, x2 = golden_input(ops_golden, tag + "/x1"), golden_input(ops_golden, tag + "/x2")
    y = O.swin_dattention(x1, x2, sd, "d")
    assert rel_err(y, ops_golden[tag + "/y"]) < TOL


def test_cross_swin_block(ops_golden):
    spec = block_spec(96, 3)
    spec.update({"pre.weight": (96, 128), "pre.bias": (96,)})
    spec.update(sda_spec(96, "cva.crossattn."))
    sd = local_sd(spec, "csb/")
    sd["attn.relative_position_index"] = rel_index()
    sd = {"c." + k: v for k, v in sd.items()}
    x1, x2 = golden_input(ops_golden, "csb/x1"), golden_input(ops_golden, "csb/x2")
    y, out = O.cross_swin_block(x1, x2, sd, "c", 14, False)
    assert rel_err(out, ops_golden["csb/out"]) < TOL
    assert rel_err(y, ops_golden["csb/y"]) < TOL
    sd = local_sd(block_spec(128, 4), "csbl/")
    sd["attn.relative_position_index"] = rel_index()
    sd = {"c." + k: v for k, v in sd.items()}
    x1 = golden_input(ops_golden, "csbl/x1")
    y, out = O.cross_swin_block(x1, None, sd, "c", 14, True)
    assert rel_err(out, ops_golden["csbl/out"]) < TOL
    assert rel_err(y, ops_golden["csbl/y"]) < TOL


def test_patch_merging(ops_golden):
    sd = {"m." + k: v for k, v in local_sd({"reduction.weight": (192, 384), "norm.weight": (384,),
                                            "norm.bias": (384,)}, "pm/").items()}
    y = O.patch_merging(golden_input(ops_golden, "pm/x"), sd, "m", 42, 14)
    assert rel_err(y, ops_golden["pm/y"]) < TOL


def test_faf(ops_golden):
    y = O.faf_frame1(golden_input(ops_golden, "faf/x"))
    assert rel_err(y[:, :, ::4, ::4], ops_golden["faf/y_sub4"]) < TOL
    assert rel_err(y[:, :, 100:104], ops_golden["faf/y_rows"]) < TOL
    check_digest(y, ops_golden, "faf/y", TOL)
    assert np.array_equal(O.dct_matrix(224)[5].numpy(), ops_golden["faf/dct_row5"])


def test_global_block(ops_golden):
    c = 768
    spec = {"norm1.weight": (c,), "norm1.bias": (c,), "norm2.weight": (c,), "norm2.bias": (c,),
            "attn.qkv.weight": (3 * c, c), "attn.qkv.bias": (3 * c,), "attn.proj.weight": (c, c),
            "attn.proj.bias": (c,), "mlp.fc1.weight": (4 * c, c), "mlp.fc1.bias": (4 * c,),
            "mlp.fc2.weight": (c, 4 * c), "mlp.fc2.bias": (c,)}
    sd = {"g." + k: v for k, v in local_sd(spec, "gb/").items()}
    y = O.global_block(golden_input(ops_golden, "gb/x"), sd, "g", 12)
    assert rel_err(y, ops_golden["gb/y"]) < TOL


def test_tokenizer(ops_golden):
    spec = {"project1.weight": (96, 3, 3, 4, 4), "project1.bias": (96,), "project2.weight": (96, 3, 2, 4, 4),
            "project2.bias": (96,), "project3.weight": (128, 3, 1, 4, 4), "project3.bias": (128,),
            "norm1.weight": (96,), "norm1.bias": (96,), "norm2.weight": (96,), "norm2.bias": (96,),
            "norm3.weight": (128,), "norm3.bias": (128,)}
    sd = {"t." + k: v for k, v in local_sd(spec, "tok/").items()}
    ys = O.tokenize(golden_input(ops_golden, "tok/x"), sd, O.MumpyConfig(frames=3), "t")
    for i, y in enumerate(ys):
        shp = ops_golden[f"tok/shape{i}"]                     # reference: (B, t, 3136, C)
        assert y.shape == (shp[0], shp[1] * shp[2], shp[3])
        assert rel_err(y.reshape(-1, y.shape[-1])[:64], ops_golden[f"tok/y{i}_head"]) < TOL
        check_digest(y, ops_golden, f"tok/y{i}", TOL)


# ------------------------------------------------------------------ whole model
FULL_TOL = 1e-4   # oracle vs reference through 36 layers (fp32 reassociation); the HIP bar is 1e-3


def _filled(manifest):
    from weight_fill import fill_state_dict_
    man = json.load(open(os.path.join(GOLDEN, manifest)))
    sd = {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, (shape, dt) in man.items()}
    for k in sd:
        if k.endswith("relative_position_index"):
            sd[k] = rel_index()
        elif k.endswith("attn_mask"):
            sd[k] = O.shift_attn_mask(49 * sd[k].shape[0] // _mask_w(k), _mask_w(k), 3)
    return fill_state_dict_(sd)


def _mask_w(key):
    stage = int(key.split("layers.layers.")[1].split(".")[0]) if "layers.layers." in key else int(
        key.split("base.layers.")[1].split(".")[0])
    return [56, 28, 14, 7][stage]


def _check_full(store, tag, logits, feats, fx, vx, dx, tol):
    assert rel_err(logits, store[tag + "/logits"]) < tol
    assert rel_err(fx, store[tag + "/final_x"]) < tol
    check_digest(dx, store, tag + "/dct_x", tol)
    check_digest(feats, store, tag + "/x_feats", tol)
    for s in range(4):
        for v in range(3):
            assert list(vx[s][v].shape) == list(store[f"{tag}/view_shape_{s}_{v}"])
            check_digest(vx[s][v], store, f"{tag}/view_{s}_{v}", tol)


@pytest.fixture(scope="module")
def sd_t3():
    return _filled("state_dict_encoder.json"), _filled("state_dict_decoder.json")


@pytest.mark.parametrize("tag", ["b1t3", "b2t3"])
def test_full_model_t3(full_golden, sd_t3, tag):
    x = golden_input(full_golden, tag + "/x")
    with torch.no_grad():
        out = O.full_forward(sd_t3[0], sd_t3[1], x)
    _check_full(full_golden, tag, *out, FULL_TOL)


def test_batch_coupling_is_reproduced(full_golden, sd_t3):
    """B=2 result differs from two B=1 runs; the oracle must show the same coupling."""
    x = golden_input(full_golden, "b2t3/x")
    with torch.no_grad():
        l2 = O.full_forward(sd_t3[0], sd_t3[1], x)[0]
        l1 = O.full_forward(sd_t3[0], sd_t3[1], x[:1])[0]
    assert rel_err(l2[:1], l1) > 1e-3


def test_full_model_t5(full_golden):
    sde, sdd = _filled("state_dict_encoder_t5.json"), _filled("state_dict_decoder_t5.json")
    x = golden_input(full_golden, "b1t5/x")
    with torch.no_grad():
        out = O.full_forward(sde, sdd, x)
    _check_full(full_golden, "b1t5", *out, FULL_TOL)


def test_full_model_b1t9(full_golden_t9):
    """Config 4's temporal length at 224x224: tubelets (9,8,1), r = 9 window sums in the deformable attention, 9x9
    temporal attention; golden from the reference's own classes (gen_goldens_t9.py)."""
    enc_sd, dec_sd = _filled("state_dict_encoder_t9.json"), _filled("state_dict_decoder_t9.json")
    x = golden_input(full_golden_t9, "b1t9/x")
    with torch.no_grad():
        out = O.full_forward(enc_sd, dec_sd, x)
    _check_full(full_golden_t9, "b1t9", *out, FULL_TOL)


def test_baseline_encoder(full_golden):
    sd = _filled("state_dict_baseline_encoder.json")
    with torch.no_grad():
        y = O.baseline_encoder_forward(sd, golden_input(full_golden, "base_b1t3/x"))
    assert rel_err(y, full_golden["base_b1t3/y"]) < FULL_TOL


def test_baseline_decoder(full_golden):
    """config 1 tail: BaselineDecoder(in_channels=1024) on the golden encoder output (decoder.py:228-284)."""
    sd = _filled("state_dict_baseline_decoder.json")
    with torch.no_grad():
        z = O.baseline_decoder_forward(sd, torch.from_numpy(full_golden["base_b1t3/y"]))
    assert z.shape == (1, 1, 224, 224)
    assert rel_err(z, full_golden["base_b1t3/logits"]) < FULL_TOL


@pytest.mark.parametrize("hs,ws", [(240, 432), (480, 854), (224, 224), (1080, 1920), (100, 37)])
def test_stage_frames_matches_pil_nearest(hs, ws):
    """SURVEY 8f-4 / config 4's 432x240 footage: the oracle's resize rule is PIL's NEAREST (the default filter of the pinned
    pillow==4.0.0 in `img.resize(self.inputRes)`, universaldataset.py:75-79), checked against PIL itself, bit exact."""
    from PIL import Image
    g = torch.Generator().manual_seed(hs * 7 + ws)
    frame = torch.randint(0, 256, (hs, ws, 3), generator=g, dtype=torch.uint8)
    pil = torch.from_numpy(np.array(Image.fromarray(frame.numpy()).resize((224, 224), Image.NEAREST)))     # PIL takes (W, H)
    ref = (pil.permute(2, 0, 1).float() / 255.0 - torch.tensor([0.4776, 0.479, 0.4465]).view(3, 1, 1)) / torch.tensor(
        [0.230, 0.2085, 0.2324]).view(3, 1, 1)
    out = O.stage_frames(frame, size=(224, 224))
    assert out.shape == (3, 224, 224)
    assert torch.equal(out, ref)
