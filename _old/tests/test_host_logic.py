"""CPU: host-side logic of the drop-in package — construction API, state_dict contract, derived tables, and the
guarantee that nothing silently runs on the CPU."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import mumpy_oracle as O


def manifest(m):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}


@pytest.mark.parametrize("fname,ctor", [
    ("state_dict_encoder.json", lambda: __import__("models.encoder.encoder", fromlist=["Encoder"]).Encoder()),
    ("state_dict_decoder.json", lambda: __import__("models.decoder.decoder", fromlist=["Decoder"]).Decoder()),
    ("state_dict_encoder_t5.json", lambda: __import__("models.encoder.encoder", fromlist=["Encoder"]).Encoder(num_frames=5)),
    ("state_dict_decoder_t5.json",
     lambda: __import__("models.decoder.decoder", fromlist=["Decoder"]).Decoder(input_token_temporal_dims=[1, 1, 5])),
    ("state_dict_baseline_encoder.json", lambda: __import__("models.encoder.encoder", fromlist=["BaselineEncoder"]).BaselineEncoder()),
    ("state_dict_baseline_decoder.json",
     lambda: __import__("models.decoder.decoder", fromlist=["BaselineDecoder"]).BaselineDecoder(in_channels=1024)),
])
def test_state_dict_contract(fname, ctor):
    """Same keys, shapes, dtypes AND order as the reference model (test.py:60-61 loads strictly)."""
    ref = json.load(open(os.path.join(GOLDEN, fname)))
    got = manifest(ctor())
    assert list(got) == list(ref)
    assert got == ref


def test_strict_load_of_a_reference_shaped_checkpoint():
    from models.decoder.decoder import Decoder
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_decoder.json")))
    sd = {k: torch.zeros(s, dtype=getattr(torch, d)) for k, (s, d) in ref.items()}
    Decoder().load_state_dict(sd, strict=True)


def test_factory_api():
    from models.factory import modelFactory as F
    cfg = F.create_view_config([96, 192, 384, 768], (4, 4, 3), [2, 2, 6, 2], [3, 6, 12, 24], 768, 1,
                               [(56, 56), (28, 28), (14, 14), (7, 7)], 1, [1, 1])
    assert cfg["window_size"] == 7 and cfg.window_size == 7
    assert cfg["patches"].size == (4, 4, 3) and cfg["hidden_size"][-1] == 768
    assert cfg["temporal_ratio"] == [1, 1]
    model, vcs = F.create_multiswin()
    assert len(vcs) == 3 and vcs[2]["temporal_dim"] == 3
    assert any("cva" in n for n, _ in model.named_parameters())      # train.py:205-209 splits params on "cva"


def test_encoder_attributes_used_by_callers():
    from models.encoder.encoder import Encoder
    e = Encoder()
    assert e.configs[0]["window_size"] == 7 and e.configs[1]["hidden_size"][-1] * 3 == 2304
    assert hasattr(e, "base")


def test_shift_mask_and_relpos_index_match_oracle(index_golden):
    from models.modules.swinTransformer import build_shift_mask, relative_position_index
    for res, t in [(56, 1), (56, 3), (56, 5), (14, 3)]:
        m = build_shift_mask(res * t, res, 7, 3)
        assert torch.equal(m, O.shift_attn_mask(res * t, res, 3))
        assert np.array_equal(np.packbits((m != 0).numpy().reshape(-1)), index_golden[f"mask_{res}_t{t}"])
    assert np.array_equal(relative_position_index(7, 7).numpy(), index_golden["relative_position_index"])


def test_window_partition_reverse_api(index_golden):
    from models.modules.swinTransformer import window_partition, window_reverse
    ramp = torch.arange(168 * 56).view(1, 168, 56, 1)
    wins = window_partition(ramp, 7)
    assert np.array_equal(wins.reshape(-1).numpy(), index_golden["part_168x56"])
    assert torch.equal(window_reverse(wins, 7, 168, 56), ramp)


def test_mask_compaction():
    """(The padded relative-position bias is built by a kernel since round 2: tests/test_hip_parity.py::test_relpos_bias_expand.)"""
    from mumpy_hip import ops
    mask = O.shift_attn_mask(168, 56, 3)
    tab, ids = ops.compact_attn_mask(mask)
    assert ids.shape == (192,) and ids.dtype == torch.int32
    assert tab.shape[0] == 3                      # last-row, last-col and corner patterns
    for n in range(192):
        if ids[n] < 0:
            assert not bool(mask[n].any())
        else:
            assert torch.equal(tab[ids[n], :49, :49], mask[n])


def test_no_cpu_fallback():
    """Ops refuse CPU tensors outright; the product never computes on the host."""
    from mumpy_hip import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm(torch.zeros(4, 96), torch.ones(96), torch.zeros(96))
    from models.modules.swinTransformer import SwinTransformerBlock
    blk = SwinTransformerBlock(96, (14, 14), 3, shift_size=3).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        blk(torch.zeros(1, 196, 96))


def test_product_does_not_import_oracle():
    import subprocess
    import sys
    from conftest import PKG
    code = ("import sys; sys.path.insert(0, %r); import models.encoder.encoder, models.decoder.decoder, mumpy_hip; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imports oracle'" % PKG)
    subprocess.run([sys.executable, "-c", code], check=True)
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(root, f)).read(), f"{f} mentions the oracle"


def test_eval_metric_formulas():
    """measure.py:57-62,86-89 restated in the oracle; checked on a hand-computed case."""
    pred = torch.zeros(1, 1, 4, 4)
    gt = torch.zeros(1, 1, 4, 4)
    pred[0, 0, :2] = 1           # 8 predicted
    gt[0, 0, 1:3] = 1            # 8 true, 4 overlap
    f1, iou = O.f1_iou_per_clip(pred, gt)
    recall = 4 / (8 + 16e-6)
    precision = 4 / (8 + 1e-6)
    assert abs(float(f1) - 2 * precision * recall / (precision + recall + 1e-6)) < 1e-12
    assert abs(float(iou) - (4 + 1e-5) / (12 + 1e-5)) < 1e-12


def test_check_parallel_strips_dataparallel_prefix(tmp_path):
    """utils/utils.py:156-176: 'module.'-prefixed checkpoints (trained under nn.DataParallel) load strictly."""
    from models.decoder.decoder import BaselineDecoder
    from mumpy_hip import checkpoint as C
    dec = BaselineDecoder(in_channels=64, features=[32] * 5)
    enc = torch.nn.Linear(4, 4)
    wrapped_e = {"module." + k: v for k, v in enc.state_dict().items()}
    wrapped_d = {"module." + k: v for k, v in dec.state_dict().items()}
    e, d = C.check_parallel(wrapped_e, wrapped_d)
    assert list(e) == list(enc.state_dict()) and list(d) == list(dec.state_dict())
    e2, d2 = C.check_parallel(enc.state_dict(), dec.state_dict())          # already clean: untouched
    assert list(e2) == list(enc.state_dict()) and list(d2) == list(dec.state_dict())
    # file round trip in the reference's naming (encoder_{epoch}.pt / decoder_{epoch}.pt), DataParallel-style keys on disk
    torch.save(wrapped_e, tmp_path / "encoder_7.pt")
    torch.save(wrapped_d, tmp_path / "decoder_7.pt")
    e3, d3, args = C.load_checkpoint(str(tmp_path), epoch=7)
    assert args is None
    enc.load_state_dict(e3, strict=True)
    dec.load_state_dict(d3, strict=True)
    C.save_checkpoint(str(tmp_path / "out"), enc, dec, epoch=None, args={"length_clip": 3, "batch_size": 8})
    e4, d4, args = C.load_checkpoint(str(tmp_path / "out"))
    assert args == {"length_clip": 3, "batch_size": 8}
    assert all(torch.equal(d4[k], v) for k, v in dec.state_dict().items())


def test_clip_frame_indices_clamp():
    """universaldataloader.py:41-46: one clip per frame, centred, clamped at both ends."""
    from mumpy_hip.checkpoint import clip_frame_indices
    assert clip_frame_indices(4, 3) == [[0, 0, 1], [0, 1, 2], [1, 2, 3], [2, 3, 3]]
    c5 = clip_frame_indices(3, 5)
    assert c5 == [[0, 0, 0, 1, 2], [0, 0, 1, 2, 2], [0, 1, 2, 2, 2]]
    assert all(len(c) == 5 for c in clip_frame_indices(10, 4))              # even length_clip -> 2k+1 = 5 frames
    assert clip_frame_indices(0, 3) == []
