#!/usr/bin/env python3
"""Per-wave phase trace of win_attn_self_kernel (MUMPY_WA_DBG=8): stamps [unit start, loads landed, unit end] per unit."""
import os, sys, ctypes, torch, numpy as np
os.environ["MUMPY_WA_DBG"] = "8"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd")]
from mumpy_hip import ops, lib as L
from models.modules.swinTransformer import relative_position_index
dev = torch.device("cuda:0")
b, hs, w, c = 40, 56, 56, 128
nblocks = 768
qkv = torch.randn(b, hs * w, 3 * c, device=dev)
bias = ops.expand_relpos_bias(torch.randn(169, c // 32, device=dev) * 0.2, relative_position_index(7, 7).to(dev))
n_out = b * hs * w * c
out = torch.zeros(n_out + nblocks * 4 * 32 * 2, device=dev)
lib = L.load_library()
for _ in range(3):
    rc = lib.mumpy_window_attention_fwd(qkv.data_ptr(), out.data_ptr(), bias.data_ptr(), None, None, 0, b, hs, w, c, 0,
                                        ctypes.c_float(32 ** -0.5), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
tr = out[n_out:].cpu().numpy().view(np.uint64).reshape(nblocks * 4, 32)
hw, xcc = tr[:, 0], tr[:, 1]
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; sh = (hw >> 12) & 1
t0 = tr[:, 2].min()
st = (tr[:, 2:].astype(np.int64) - np.int64(t0))
st[tr[:, 2:] == 0] = -1
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu.astype(np.int64)
print("distinct CUs:", len(set(key.tolist())), "waves:", len(key), " max stamp (ticks):", st.max())
# print all waves of one CU, one SIMD
k0 = key[0]
for sd in range(4):
    idx = [i for i in range(len(key)) if key[i] == k0 and simd[i] == sd]
    print(f"CU {k0:#x} SIMD {sd}: {len(idx)} waves")
    for i in idx:
        row = st[i]; units = [(row[3 * u], row[3 * u + 1], row[3 * u + 2]) for u in range(9) if row[3 * u + 2] > 0]
        print("   wave", i, " ".join(f"[{a}|{b_ - a}|{c_ - b_}]" for a, b_, c_ in units))
# aggregate phase durations
ld, cp = [], []
for i in range(len(key)):
    row = st[i]
    for u in range(9):
        if row[3 * u + 2] > 0:
            ld.append(row[3 * u + 1] - row[3 * u]); cp.append(row[3 * u + 2] - row[3 * u + 1])
print("mean load-wait ticks %.0f, mean compute+store ticks %.0f, n=%d" % (np.mean(ld), np.mean(cp), len(ld)))
