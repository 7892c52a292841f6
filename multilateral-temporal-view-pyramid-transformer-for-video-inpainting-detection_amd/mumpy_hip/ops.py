"""Tensor-level wrappers over the C ABI.  Each takes/returns torch CUDA(HIP) fp32 tensors, allocates outputs with
torch (caching allocator => graph-capturable), and enqueues on torch's current stream.  No fallbacks."""
import math
from typing import Optional

import torch

from .lib import load_library

ACT_NONE, ACT_GELU = 0, 1
MATH_FP32, MATH_BF16, MATH_BF16X3, MATH_BF16X2 = 0, 0x100, 0x200, 0x400
_MATH = MATH_FP32          # OR-ed into the `act` argument of every linear / conv launch
_WS_BYTES = {}


def set_matrix_math(mode: str) -> None:
    """"fp32" (default): exact fp32 products on v_mfma_f32_32x32x2_f32.  "bf16": operands of every GEMM / convolution are
    rounded to bf16 while being staged and multiplied on the bf16 MFMA with fp32 accumulation (config 3's arithmetic);
    tensors in memory, LayerNorm / softmax / GroupNorm statistics and all other kernels stay fp32.
    "bf16x3": fp32 products on the bf16 matrix pipe -- each operand is split into three bf16 pieces while staged and the
    six significant piece products are accumulated in fp32; fp32-level accuracy (see include/mumpy_hip.h).
    "bf16x2": two pieces / three products: 16-bit-mantissa operands (TF32-class and better), a reduced-precision mode."""
    global _MATH
    if mode not in _MODES:
        raise ValueError(f"unknown matrix math mode {mode!r}")
    _MATH = _MODES[mode]


_MODES = {"fp32": MATH_FP32, "bf16": MATH_BF16, "bf16x3": MATH_BF16X3, "bf16x2": MATH_BF16X2}


def matrix_math() -> str:
    return {v: k for k, v in _MODES.items()}[_MATH]


_STORAGE = "fp32"


def set_storage(mode: str) -> None:
    """"fp32" (default) or "bf16": BASELINE config 3 as written -- inside the Swin blocks (and the MLPs of the global
    blocks) LayerNorm writes bf16, the qkv / fc1 GEMMs read bf16 activations and bf16 copies of their weights and write
    bf16, window attention reads and writes bf16, and the proj / fc2 GEMMs read bf16 and add into the fp32 residual stream.
    Accumulation, softmax and LayerNorm statistics stay fp32.  Implies set_matrix_math("bf16") for the remaining GEMMs."""
    global _STORAGE
    if mode not in ("fp32", "bf16"):
        raise ValueError(f"unknown storage mode {mode!r}")
    _STORAGE = mode
    set_matrix_math("bf16" if mode == "bf16" else "fp32")


def storage() -> str:
    return _STORAGE


def _chk16(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.bfloat16:
        raise RuntimeError(f"mumpy_hip: {name} must be a bfloat16 GPU tensor, got {t.dtype} on {t.device}")
    return t if t.is_contiguous() else t.contiguous()


def layernorm_bf16(x, gamma, beta, eps=1e-5):
    """LayerNorm of an fp32 tensor, written as bf16 (statistics in fp32)."""
    x = _chk(x, "x")
    c = x.shape[-1]
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _call("mumpy_layernorm_bf16_fwd", _p(x), _p(_chk(gamma, "gamma")), _p(_chk(beta, "beta")), _p(out), x.numel() // c, c, eps,
          _stream(), work=6.0 * x.numel())
    return out


def linear_bf16s(x16, w16, bias=None, act=ACT_NONE, residual=None, out_bf16=True):
    """y = act(x16 @ w16.T + bias) + residual with bf16 x / W in memory, fp32 accumulate; y bf16 or fp32 (residual fp32)."""
    x16, w16 = _chk16(x16, "x"), _chk16(w16, "weight")
    n, k = w16.shape[0], w16.numel() // w16.shape[0]
    if x16.shape[-1] != k:
        raise RuntimeError(f"linear_bf16s: x has {x16.shape[-1]} features, weight expects {k}")
    m = x16.numel() // k
    out = torch.empty(*x16.shape[:-1], n, device=x16.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
        if residual.numel() != m * n or out_bf16:
            raise RuntimeError("linear_bf16s: the residual is fp32 and needs an fp32 output of the same shape")
    _call("mumpy_linear_bf16s_fwd", _p(x16), _p(w16), _p(None if bias is None else _chk(bias, "bias")), _p(residual), _p(out),
          m, n, k, act, 1 if out_bf16 else 0, _stream(), work=2.0 * m * n * k)
    return out


def window_attention_bf16(qkv16, bias_pad, b, hs, w, c, shift, scale, mask_tab=None, mask_id=None):
    """bf16 qkv (B, hs*w, 3C) -> bf16 (B, hs*w, C)."""
    qkv16 = _chk16(qkv16, "qkv")
    if qkv16.numel() != b * hs * w * 3 * c:
        raise RuntimeError("window_attention_bf16: qkv shape mismatch")
    out = torch.empty(b, hs * w, c, device=qkv16.device, dtype=torch.bfloat16)
    n_mask = 0 if mask_id is None else mask_id.numel()
    _call("mumpy_window_attention_bf16_fwd", _p(qkv16), _p(out), _p(_chk(bias_pad, "bias")), _p(mask_tab), _p(mask_id), n_mask,
          b, hs, w, c, shift, scale, _stream(), work=307328.0 * b * (hs // 7) * (w // 7) * (c // 32))
    return out
NEG = -1e30


def _lib():
    return load_library()


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"mumpy_hip: {name} is on {t.device}; the HIP kernels need a GPU tensor (there is no CPU path)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"mumpy_hip: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py): when PROFILE is a dict, every C-ABI call is bracketed by events on the
# stream it is launched on and recorded as PROFILE[name] -> [(start_event, end_event, work), ...].
PROFILE = None


def _call(name, *args, work=0.0):
    lib = _lib()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        PROFILE.setdefault(name, []).append((e0, e1, work))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.mumpy_last_error().decode()}")


# ------------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps=1e-5, out=None):
    x = _chk(x, "x")
    c = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    _call("mumpy_layernorm_fwd", _p(x), _p(_chk(gamma, "gamma")), _p(_chk(beta, "beta")), _p(out), x.numel() // c, c,
          eps, _stream(), work=8.0 * x.numel())
    return out


def linear(x, weight, bias=None, act=ACT_NONE, residual=None, out=None, emit_stats=False):
    """y = act(x @ weight.T + bias) + residual; weight (N,K) or a 1x1-conv kernel (N,K,1,1).
    emit_stats: y feeds a LayerNorm whose consumer can fold it (linear_ln): when the shape runs on the persistent kernel the
    epilogue also writes the per-tile row statistics, attached to the result as y._mumpy_ln_stats (else nothing happens)."""
    x = _chk(x, "x")
    weight = _chk(weight, "weight")
    n, k = weight.shape[0], weight.numel() // weight.shape[0]
    if x.shape[-1] != k:
        raise RuntimeError(f"linear: x has {x.shape[-1]} features, weight expects {k}")
    m = x.numel() // k
    if out is None:
        out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
        if residual.numel() != m * n:
            raise RuntimeError("linear: residual shape mismatch")
    if _in_background():
        _call("mumpy_linear_rd_fwd", _p(x), _p(weight), _p(None if bias is None else _chk(bias, "bias")), _p(residual), _p(out), m, n, k,
              act, _stream(), work=2.0 * m * n * k)
        return out
    key = (m, n, k)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_linear_workspace_bytes(m, n, k))
    if emit_stats:
        # measured (tools/ln_fold_shapes.py, profiles/r03_ln_fold_shapes.txt): the statistics epilogue costs 4-5 us on a producer with
        # >= 16 chunks per tile (K >= 512) -- less than the LayerNorm launch it saves -- but 9-21 us on the short-K producers of
        # stages 0 / 1 (K = 128, 256: their epilogue waves are the bottleneck already), more than the launch
        gn = linear_ln_tiles(m, n, k) if k >= LN_FOLD_MIN_K else 0
        if gn:
            stats = torch.empty(m, gn, 2, device=x.device, dtype=torch.float32)
            ws = _kept_workspace(max(wsb, 4096), x.device)
            _call("mumpy_linear_lnx_fwd", _p(x), _p(weight), _p(None if bias is None else _chk(bias, "bias")), _p(residual), _p(out), m, n, k,
                  act, _p(ws), ws.numel() * 4, _p(stats), None, 0, None, 0.0, _stream(), work=2.0 * m * n * k)
            out._mumpy_ln_stats = stats
            return out
    ws = _kept_workspace(wsb, x.device) if wsb else None      # split-K slabs / the persistent kernel's flags + slabs
    _call("mumpy_linear_wsz_fwd", _p(x), _p(weight), _p(None if bias is None else _chk(bias, "bias")), _p(residual),
          _p(out), m, n, k, act | _MATH, _p(ws), 0 if ws is None else ws.numel() * 4, _stream(), work=2.0 * m * n * k)
    return out


# ---- "background" kernels (no LDS: resident beside the persistent GEMM, csrc/gemm_rd.hip) ---------------------------------------
_BACKGROUND = [0]
# Measured and NOT adopted (default off; profiles/r03_coresidency_probe.txt): the LDS-free GEMM is resident-compatible with the
# persistent kernel but two MFMA-bound kernels gain nothing from sharing a CU (together = sum on the two-stream probe), and alone
# it is slower than the tiled kernels -- the forward went 21.5 -> 23.3 ms with it.  MUMPY_BACKGROUND=1 turns it on for A/B runs.
BACKGROUND_ON = __import__("os").environ.get("MUMPY_BACKGROUND", "0") == "1"


class background:
    """Context manager for work that is forked beside a chain of large GEMMs (views 1 / 2 beside view 3 inside a pyramid stage):
    inside it, fp32 `linear` and `window_attention` launch their LDS-free forms (mumpy_linear_rd_fwd, mumpy_window_attention_bg_fwd),
    which can be resident on a CU whose whole LDS belongs to the persistent GEMM.  Same results; slower when run alone."""

    def __enter__(self):
        _BACKGROUND[0] += 1
        return self

    def __exit__(self, *exc):
        _BACKGROUND[0] -= 1
        return False


def _in_background():
    return _BACKGROUND[0] > 0 and BACKGROUND_ON and _MATH == MATH_FP32 and _STORAGE == "fp32"


# ---- LayerNorm folded into the GEMMs either side of it (mumpy_linear_lnx_fwd; swin:266,305 / blocks:86-88) -----------------
LN_FOLD = __import__("os").environ.get("MUMPY_LN_FOLD", "1") != "0"      # A/B switch
LN_FOLD_MIN_K = 512
_LN_TILES = {}


def linear_ln_tiles(m, n, k):
    """Column tiles of a shape whose fp32 launch runs on the persistent 128x128 kernel (the one that can emit / consume the
    per-tile LayerNorm statistics), else 0.  0 as well whenever folding is off or another arithmetic mode is active."""
    if not LN_FOLD or _MATH != MATH_FP32 or _STORAGE != "fp32":
        return 0
    key = (m, n, k)
    t = _LN_TILES.get(key)
    if t is None:
        t = _LN_TILES[key] = int(_lib().mumpy_linear_ln_tiles(m, n, k))
    return t


def ln_stats_of(x):
    """The per-tile row statistics the producing GEMM attached to x (linear(..., emit_stats=True)), or None."""
    return getattr(x, "_mumpy_ln_stats", None)


def linear_ln(x, stats, wg, colsum, bprime, eps, act=ACT_NONE):
    """y = act(LayerNorm(x) W^T + b) with the LayerNorm folded into the GEMM: x RAW, stats from the producer of x, wg = W diag(gamma),
    colsum = row sums of wg, bprime = W beta + b (see fold_ln_weights).  The caller checked linear_ln_tiles(m, n, k) > 0."""
    x, wg = _chk(x, "x"), _chk(wg, "wg")
    n, k = wg.shape
    m = x.numel() // k
    out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    wsb = _WS_BYTES.get((m, n, k))
    if wsb is None:
        wsb = _WS_BYTES[(m, n, k)] = int(_lib().mumpy_linear_workspace_bytes(m, n, k))
    ws = _kept_workspace(max(wsb, 4096), x.device)
    _call("mumpy_linear_lnx_fwd", _p(x), _p(wg), _p(_chk(bprime, "bprime")), None, _p(out), m, n, k, act, _p(ws), ws.numel() * 4, None,
          _p(_chk(stats, "stats")), stats.shape[-2], _p(_chk(colsum, "colsum")), eps, _stream(), work=2.0 * m * n * k)
    return out


def fold_ln_weights(weight, bias, gamma, beta):
    """(W diag(gamma), its row sums, W beta + b) in fp32, formed in float64: the operands of linear_ln."""
    w64 = weight.detach().double()
    wg = w64 * gamma.detach().double()[None, :]
    bp = w64 @ beta.detach().double()
    if bias is not None:
        bp = bp + bias.detach().double()
    return wg.float().contiguous(), wg.sum(1).float().contiguous(), bp.float().contiguous()


_KEPT_WS = {}
_RETIRED_WS = []          # superseded workspaces: never freed (a captured hipGraph may have their address baked in)


def _kept_workspace(nbytes, device):
    """One zero-initialised workspace per (device, stream), grown on demand and kept: launches on one stream execute in
    order, so they can share it; the persistent GEMM's arrival flags (its first page) return to zero after every launch
    (mumpy_linear_wsz_fwd), so nothing has to be reset between launches.  Allocated outside any graph capture (the eager
    warm-up pass that precedes a capture creates the buffers the captured launches then point at).
    Lifetime: a workspace that is outgrown is RETIRED, not freed -- GraphedForward / GraphedTrainStep replays keep writing
    flags and slabs through the address they captured, so handing that memory back to the caching allocator would let a
    later tensor alias it.  (Two torch Stream objects with one raw handle are one HIP stream: sharing a workspace between
    them is the in-order case above; mumpy_hip.streams never hands out a side stream whose handle aliases its parent.)"""
    s = torch.cuda.current_stream(device)
    key = (str(device), s.cuda_stream)
    ws = _KEPT_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(nbytes // 4, device=device, dtype=torch.float32)     # (captured fill: still correct, just not free)
        if ws is not None:
            _RETIRED_WS.append(ws)
        ws = _KEPT_WS[key] = torch.zeros(max(nbytes // 4, 1024), device=device, dtype=torch.float32)
    return ws


def check_workspaces():
    """Synchronising check of every kept workspace's sticky status word: raises if a split GEMM launch reported a part that
    never arrived (its output would be incomplete), after re-zeroing the workspaces so that the process can go on."""
    import ctypes
    bad = []
    for key, ws in list(_KEPT_WS.items()):
        st = ctypes.c_int(0)
        rc = _lib().mumpy_workspace_status(ws.data_ptr(), ctypes.byref(st))
        if rc:
            raise RuntimeError(f"mumpy_workspace_status failed (rc={rc}): {_lib().mumpy_last_error().decode()}")
        if st.value == -1:
            # LayerNorm-folding precision guard: a row with |mean| > 256 sigma went through the folded form (~1e-4 instead of ~1e-6
            # relative error on that row).  Results are complete; from here on this process takes the two-launch route.
            global LN_FOLD
            LN_FOLD = False
            ws.view(torch.int32)[1022] = 0
            import warnings
            warnings.warn("mumpy_hip: a folded LayerNorm met a row with |mean| > 256 sigma; LayerNorm folding is now off "
                          "(ops.LN_FOLD = False) -- re-capture any hipGraph to apply it")
        elif st.value:
            bad.append((key, st.value))
    if bad:
        reset_workspaces()
        raise RuntimeError(f"mumpy_hip: split GEMM launch(es) timed out waiting for a partial tile {bad}; results since the last "
                           "check are invalid (workspaces re-zeroed)")


def reset_workspaces():
    """Re-zero every kept workspace (arrival-flag pages included).  For use after a launch reported an error or was
    aborted mid-kernel: the persistent GEMM's flags are only guaranteed to be back at zero after a launch that completed."""
    for ws in list(_KEPT_WS.values()) + _RETIRED_WS:
        ws.zero_()


def linear_rows(x_view, weight, bias=None, residual=None, out=None):
    """Linear over a (nblk, rows, K) VIEW whose rows are contiguous but whose blocks are strided (no copy)."""
    if not x_view.is_cuda or x_view.dtype != torch.float32:
        raise RuntimeError("mumpy_hip: linear_rows needs a float32 GPU tensor (there is no CPU path)")
    nblk, rows, k = x_view.shape
    if x_view.stride(2) != 1 or x_view.stride(1) != k:
        raise RuntimeError("linear_rows: rows of a block must be contiguous")
    weight = _chk(weight, "weight")
    n = weight.shape[0]
    m = nblk * rows
    if out is None:
        out = torch.empty(m, n, device=x_view.device, dtype=torch.float32)
    key = (m, n, k)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_linear_workspace_bytes(m, n, k))
    ws = torch.empty(wsb // 4, device=x_view.device, dtype=torch.float32) if wsb else None
    _call("mumpy_linear_rows_fwd", x_view.data_ptr(), rows, x_view.stride(0), _p(weight),
          _p(None if bias is None else _chk(bias, "bias")), _p(residual), _p(out), m, n, k, ACT_NONE | _MATH, _p(ws), wsb, _stream(),
          work=2.0 * m * n * k)
    return out


def linear_time_slices(x4, weight, bias=None, residual=None):
    """x4 (B, T, n, C) contiguous, weight (N, T*C) with K index (t, c): y[(b, i), :] = sum_t x4[b, t, i, :] @ weight[:, t*C:(t+1)*C].T
    -- a Conv3d(k = s = (T,1,1)) head on channels-last tokens (decoder.py:62-66) in ONE launch (segmented-K rows mode)."""
    x4, weight = _chk(x4, "x"), _chk(weight, "weight")
    b, t, n, c = x4.shape
    nout, k = weight.shape
    if k != t * c or c % 32:
        raise RuntimeError(f"linear_time_slices: weight expects K = T*C = {t * c} (got {k}); C % 32 == 0")
    m = b * n
    out = torch.empty(m, nout, device=x4.device, dtype=torch.float32)
    if residual is not None:
        residual = _chk(residual, "residual")
    key = (m, nout, k)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_linear_workspace_bytes(m, nout, k))
    ws = torch.empty(wsb // 4, device=x4.device, dtype=torch.float32) if wsb else None
    _call("mumpy_linear_rows_kseg_fwd", _p(x4), n, t * n * c, c, n * c, _p(weight), _p(None if bias is None else _chk(bias, "bias")),
          _p(residual), _p(out), m, nout, k, ACT_NONE | _MATH, _p(ws), wsb, _stream(), work=2.0 * m * nout * k)
    return out


def conv2d_nhwc(x, w_krsc, bias=None, act=ACT_NONE, residual=None):
    """x logical (B,Cin,H,W) with NHWC memory; w_krsc (Cout,kh,kw,Cin) contiguous; stride 1, same padding.
    Returns logical (B,Cout,H,W) with NHWC memory."""
    x = _nhwc(x, "x")
    b, cin, h, w = x.shape
    w_krsc = _chk(w_krsc, "weight")
    cout, kh, kw, cin2 = w_krsc.shape
    if cin2 != cin:
        raise RuntimeError(f"conv2d: input has {cin} channels, weight expects {cin2}")
    out = empty_nhwc(b, cout, h, w, x.device)
    if residual is not None:
        residual = _nhwc(residual, "residual")
    key = ("conv", b, h, w, cin, cout, kh, kw)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_conv2d_workspace_bytes(b, h, w, cin, cout, kh, kw))
    ws = torch.empty(wsb // 4, device=x.device, dtype=torch.float32) if wsb else None
    _call("mumpy_conv2d_nhwc_fwd", _p(x), _p(w_krsc), _p(None if bias is None else _chk(bias, "bias")), _p(residual), _p(out),
          b, h, w, cin, cout, kh, kw, act | _MATH, _p(ws), wsb, _stream(), work=2.0 * b * h * w * cout * kh * kw * cin)
    return out


def final_conv(x, w_krsc, bias, with_mask=False, thr=0.5):
    """x logical (B,C,H,W) NHWC, C a multiple of 32 -> logits (B,1,H,W) [, uint8 mask (B,1,H,W)]."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    if c % 32 or tuple(w_krsc.shape) != (1, 3, 3, c):
        raise RuntimeError(f"final_conv is built for Conv2d(32k, 1, 3, padding=1); got C={c}, weight {tuple(w_krsc.shape)}")
    logits = torch.empty(b, 1, h, w, device=x.device, dtype=torch.float32)
    mask = torch.empty(b, 1, h, w, device=x.device, dtype=torch.uint8) if with_mask else None
    _call("mumpy_final_conv_fwd", _p(x), _p(_chk(w_krsc, "weight")), _p(_chk(bias, "bias")), _p(logits), _p(mask), b, h, w, c, thr,
          _stream(), work=4.0 * (x.numel() + logits.numel()))
    return (logits, mask) if with_mask else logits


def _nhwc(t: torch.Tensor, name: str) -> torch.Tensor:
    """Logical (B,C,H,W) tensor whose memory is NHWC (torch.channels_last)."""
    if not t.is_cuda:
        raise RuntimeError(f"mumpy_hip: {name} is on {t.device}; the HIP kernels need a GPU tensor (there is no CPU path)")
    if t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError(f"mumpy_hip: {name} must be a 4-D float32 tensor")
    b, c, h, w = t.shape
    if t.stride() != (h * w * c, 1, w * c, c):
        t = t.contiguous(memory_format=torch.channels_last)
        if t.stride() != (h * w * c, 1, w * c, c):           # degenerate sizes: force the exact NHWC strides
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return t


def empty_nhwc(b, c, h, w, device):
    return torch.empty(b, h, w, c, device=device, dtype=torch.float32).permute(0, 3, 1, 2)


ACT_RELU, ACT_SIGMOID = 1, 2
EP_NONE, EP_ADD_MUL, EP_MUL = 0, 1, 2


def gn_stats(x, groups):
    """x: logical (B,C,H,W), NHWC memory -> (partial, nsplit)."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    # 64 KB of the image per workgroup (16 dependent 16-byte loads per thread): the first rule (256 KB, at most 64 splits) left the
    # decoder's 28x28 .. 112x112 maps on 24-96 workgroups of 256 CUs, each walking 64 loads in sequence (19 us per launch)
    nsplit = max(1, min(256, (h * w * c) // 16384))
    partial = torch.empty(b, nsplit, groups, 2, device=x.device, dtype=torch.float32)
    _call("mumpy_gn_stats_nhwc_fwd", _p(x), _p(partial), b, h * w, c, groups, nsplit, _stream(), work=4.0 * x.numel())
    return x, partial, nsplit


def gn_apply_resample(x, gn=None, act=0, mean4=False, scale=1, align_corners=False, ep_mode=0, ep_a=None, ep_b=None,
                      out=None, out_coff=0):
    """x logical (B,C,H,W) NHWC.  gn = (partial, nsplit, gamma, beta, groups, eps) or None.  Returns logical NCHW / NHWC memory."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    cout = c // 4 if mean4 else c
    ho, wo = h * scale, w * scale
    if out is None:
        out = empty_nhwc(b, cout, ho, wo, x.device)
    else:
        assert out.shape[0] == b and out.shape[2] == ho and out.shape[3] == wo and out.stride(1) == 1
    if ep_a is not None:
        ep_a = _nhwc(ep_a, "ep_a")
        assert tuple(ep_a.shape) == (b, cout, ho, wo), "epilogue operand shape"
    if ep_b is not None:
        ep_b = _nhwc(ep_b, "ep_b")
        assert tuple(ep_b.shape) == (b, cout, ho, wo), "epilogue operand shape"
    partial, nsplit, gamma, beta, groups, eps = gn if gn is not None else (None, 0, None, None, 1, 0.0)
    _call("mumpy_gn_apply_resample_nhwc_fwd", _p(x), _p(partial), nsplit, _p(gamma), _p(beta), groups, eps, act,
          1 if mean4 else 0, scale, 1 if align_corners else 0, ep_mode, _p(ep_a), _p(ep_b), _p(out), out.shape[1], out_coff,
          b, h, w, c, _stream(), work=4.0 * (x.numel() + b * cout * ho * wo))
    return out


def avgpool2_pad(x, cpad=None, nchw_in=False):
    """2x2 average pooling -> logical (B,Cpad,H/2,W/2) with NHWC memory; channels [C,Cpad) are zero.  x: logical (B,C,H,W) with
    NHWC memory, or (nchw_in) a contiguous NCHW tensor such as the FAF output."""
    if nchw_in:
        x = _chk(x, "x")
    else:
        x = _nhwc(x, "x")
    b, c, h, w = x.shape
    cpad = c if cpad is None else cpad
    out = empty_nhwc(b, cpad, h // 2, w // 2, x.device)
    _call("mumpy_avgpool2_pad_nhwc_fwd", _p(x), _p(out), b, h, w, c, cpad, 1 if nchw_in else 0, _stream(),
          work=4.0 * (x.numel() + out.numel()))
    return out


def copy_rows(src, src_stride, dst, dst_stride, rows, cols):
    """dst[r, :cols] = src[r, :cols] for r < rows, row pitches in floats; src / dst are tensors whose data_ptr() is row 0."""
    if not (src.is_cuda and dst.is_cuda and src.dtype == dst.dtype == torch.float32):
        raise RuntimeError("mumpy_hip: copy_rows needs float32 GPU tensors (there is no CPU path)")
    _call("mumpy_copy_rows_fwd", src.data_ptr(), src_stride, dst.data_ptr(), dst_stride, rows, cols, _stream(), work=8.0 * rows * cols)
    return dst


def set_channels(dst, coff, src):
    """dst[:, coff:coff+C] = src for logical (B,*,H,W) tensors with NHWC memory (a slice of a channel-concatenated map).
    src may be any per-pixel-contiguous strided view whose pixels are uniformly pitched (e.g. the first 3 of 5 temporal slices)."""
    b, c, h, w = src.shape
    ctot = dst.shape[1]
    if dst.stride() != (h * w * ctot, 1, w * ctot, ctot) or tuple(dst.shape[2:]) != (h, w) or dst.shape[0] != b:
        raise RuntimeError("set_channels: dst must be a dense NHWC map of the same batch and size")
    pitch = src.stride(3)
    if src.stride(1) != 1 or src.stride(2) != w * pitch or (b > 1 and src.stride(0) != h * w * pitch) or pitch < c:
        src = _nhwc(src, "src")
        pitch = c
    return copy_rows(src, pitch, dst[:, coff:], ctot, b * h * w, c)


def merge_views(views, token_t, n=49):
    """[(B, t_v * n, C_v)] x 3 -> ((B n T), sum C_v): the channel merge in front of the global embedding (mTVE:710-718, 739)."""
    v = [_chk(t, "view") for t in views]
    b = v[0].shape[0]
    tmax = max(token_t)
    out = torch.empty(b * n * tmax, sum(t.shape[2] for t in v), device=v[0].device, dtype=torch.float32)
    _call("mumpy_merge_views_fwd", _p(v[0]), _p(v[1]), _p(v[2]), _p(out), b, tmax, n, v[0].shape[2], v[1].shape[2], v[2].shape[2],
          token_t[0], token_t[1], token_t[2], _stream(), work=8.0 * out.numel())
    return out


def trunk_head(g, f, gcn, freq):
    """gcn * freq + PixelShuffle(2)(g * f): g, f logical (B,4C,h,w), gcn, freq (B,C,2h,2w), all NHWC memory (decoder.py:198-205)."""
    g, f, gcn, freq = _nhwc(g, "g"), _nhwc(f, "f"), _nhwc(gcn, "gcn"), _nhwc(freq, "freq")
    b, c4, h, w = g.shape
    c = c4 // 4
    if f.shape != g.shape or tuple(gcn.shape) != (b, c, 2 * h, 2 * w) or freq.shape != gcn.shape:
        raise RuntimeError("trunk_head: shape mismatch")
    z = empty_nhwc(b, c, 2 * h, 2 * w, g.device)
    _call("mumpy_trunk_head_fwd", _p(g), _p(f), _p(gcn), _p(freq), _p(z), b, h, w, c, _stream(), work=4.0 * (2 * g.numel() + 3 * z.numel()))
    return z


def add(a, b, out=None):
    a, b = _chk(a, "a"), _chk(b, "b")
    out = torch.empty_like(a) if out is None else out
    _call("mumpy_add_fwd", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def rel_index32(index: torch.Tensor) -> torch.Tensor:
    """The (49*49,) int32 image of a `relative_position_index` buffer on its device, kept ON the buffer object (the buffer is a
    constant of the model; the conversion used to be two launches per W-MSA call of a training step).  A module moved with
    `.to(device)` gets new buffer objects, hence a fresh image."""
    t = getattr(index, "_mumpy_i32", None)
    if t is None or t.device != index.device or getattr(index, "_mumpy_i32_version", -1) != index._version:
        t = index.to(torch.int32).reshape(-1).contiguous()
        index._mumpy_i32, index._mumpy_i32_version = t, index._version
    return t


def expand_relpos_bias(table: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """relative_position_bias_table (169,nH) + relative_position_index (49,49) -> (nH,64,64) padded bias
    [head][query][key]: rows >= 49 zero, key columns >= 49 = -1e30 (swin:148-151).  One launch
    (mumpy_relpos_bias_expand_fwd)."""
    table = _chk(table, "table")
    nh = table.shape[1]
    if not index.is_cuda:
        index = index.to(table.device)
    idx = index if index.dtype == torch.int32 and index.dim() == 1 else rel_index32(index)
    out = torch.empty(nh, 64, 64, device=table.device, dtype=torch.float32)
    _call("mumpy_relpos_bias_expand_fwd", _p(table), _p(idx), _p(out), nh, _stream())
    return out


_PAD_MASK = {}


def pad_mask(device=None) -> torch.Tensor:
    """(1,64,64) key-padding mask of the deformable attention (0 / -1e30 on key columns >= 49); cached per device."""
    if device is None:
        m = torch.zeros(1, 64, 64, dtype=torch.float32)
        m[:, :, 49:] = NEG
        return m
    key = str(device)
    if key not in _PAD_MASK:
        _PAD_MASK[key] = pad_mask().to(device)
    return _PAD_MASK[key]


def compact_attn_mask(mask: torch.Tensor):
    """attn_mask (nW,49,49) of 0/-100 (swin:252) -> (mask_tab (nU,64,64), mask_id (nW) int32, -1 = all-zero)."""
    m = mask.detach().float().cpu().reshape(mask.shape[0], -1)
    uniq, inv = torch.unique(m, dim=0, return_inverse=True)
    nz = [i for i in range(uniq.shape[0]) if bool((uniq[i] != 0).any())]
    remap = torch.full((uniq.shape[0],), -1, dtype=torch.int32)
    for j, i in enumerate(nz):
        remap[i] = j
    tab = torch.zeros(max(len(nz), 1), 64, 64)
    for j, i in enumerate(nz):
        tab[j, :49, :49] = uniq[i].reshape(49, 49)
    return tab.to(mask.device), remap[inv].to(torch.int32).to(mask.device)


def window_attention(qkv, bias_pad, b, hs, w, c, shift, scale, mask_tab=None, mask_id=None, out=None):
    """qkv (B, hs*w, 3C) raster -> (B, hs*w, C) raster attention output (before proj)."""
    qkv = _chk(qkv, "qkv")
    if qkv.numel() != b * hs * w * 3 * c:
        raise RuntimeError("window_attention: qkv shape mismatch")
    out = torch.empty(b, hs * w, c, device=qkv.device, dtype=torch.float32) if out is None else out
    n_mask = 0 if mask_id is None else mask_id.numel()
    _call("mumpy_window_attention_bg_fwd" if _in_background() else "mumpy_window_attention_fwd", _p(qkv), _p(out), _p(_chk(bias_pad, "bias")), _p(mask_tab), _p(mask_id), n_mask,
          b, hs, w, c, shift, scale, _stream(), work=307328.0 * b * (hs // 7) * (w // 7) * (c // 32))
    return out


def deform_offsets(q, dw_w, dw_b, ln_g, ln_b, pw_w, b, h, w, c):
    q = _chk(q, "q")
    nwin = b * (h // 7) * (w // 7)
    pos = torch.empty(nwin, 3, 49, 2, device=q.device, dtype=torch.float32)
    _call("mumpy_deform_offsets_fwd", _p(q), _p(_chk(dw_w, "dw_w")), _p(_chk(dw_b, "dw_b")), _p(_chk(ln_g, "ln_g")),
          _p(_chk(ln_b, "ln_b")), _p(_chk(pw_w, "pw_w")), _p(pos), b, h, w, c, _stream())
    return pos


def deform_sample(x2, pos, b, hs2, w, c, nq):
    x2 = _chk(x2, "x2")
    nw2 = b * (hs2 // 7) * (w // 7)
    out = torch.empty(nw2, 49, c, device=x2.device, dtype=torch.float32)
    _call("mumpy_deform_sample_fwd", _p(x2), _p(_chk(pos, "pos")), _p(out), b, hs2, w, c, nq, _stream(),
          work=4.0 * (2 * nw2 * 49 * c + nw2 * 3 * 49 * 2))       # bytes: read kv once, write sampled once, read offsets
    return out


def deform_sample_kv(x2, pos, wkv, bkv, b, hs2, w, c, nq):
    """[proj_k | proj_v] of the bilinearly sampled kv windows in one launch (the sampled map is never materialised)."""
    x2 = _chk(x2, "x2")
    nw2 = b * (hs2 // 7) * (w // 7)
    kv = torch.empty(nw2, 49, 2 * c, device=x2.device, dtype=torch.float32)
    _call("mumpy_deform_sample_kv_fwd", _p(x2), _p(_chk(pos, "pos")), _p(_chk(wkv, "wkv")), _p(_chk(bkv, "bkv")), _p(kv), b, hs2, w, c, nq,
          _stream(), work=2.0 * nw2 * 49 * 2 * c * c)
    return kv


def deform_out_combine(o, wout, bout, x1, b, h, w, c):
    """x1 + x1[window order] + scrambled proj_out(o) in one launch (replaces linear + deform_combine)."""
    o, x1 = _chk(o, "o"), _chk(x1, "x1")
    out = torch.empty_like(x1)
    _call("mumpy_deform_out_combine_fwd", _p(o), _p(_chk(wout, "wout")), _p(_chk(bout, "bout")), _p(x1), _p(out), b, h, w, c, _stream(),
          work=2.0 * o.numel() * c)
    return out


def deform_attention(q, kv, padmask, b, h, w, c, r, scale):
    q, kv = _chk(q, "q"), _chk(kv, "kv")
    b1w = b * (h // 7) * (w // 7)
    if kv.numel() != b1w * r * 49 * 2 * c:
        raise RuntimeError("deform_attention: kv shape mismatch")
    out = torch.empty(b1w, 49, c, device=q.device, dtype=torch.float32)
    _call("mumpy_deform_attention_fwd", _p(q), _p(kv), _p(_chk(padmask, "padmask")), _p(out), b, h, w, c, r, scale,
          _stream(), work=307328.0 * b1w * r * (c // 32))
    return out


def deform_combine(x1, yt, b, h, w, c):
    x1, yt = _chk(x1, "x1"), _chk(yt, "yt")
    out = torch.empty_like(x1)
    _call("mumpy_deform_combine_fwd", _p(x1), _p(yt), _p(out), b, h, w, c, _stream())
    return out


def faf(x, d, dt, frame, lo_hi, mid_lo, mid_hi):
    x = _chk(x, "x")
    b, t = x.shape[0], x.shape[1]
    if tuple(x.shape[2:]) != (3, 224, 224):
        raise RuntimeError(f"faf: expects (B,T,3,224,224) clips (dct.py:57,72), got {tuple(x.shape)}")
    scratch = torch.empty(b, 3, 224, 224, device=x.device, dtype=torch.float32)
    out = torch.empty(b, 9, 224, 224, device=x.device, dtype=torch.float32)
    _call("mumpy_faf_fwd", _p(x), _p(_chk(d, "D")), _p(_chk(dt, "Dt")), _p(scratch), _p(out), b, t, frame, lo_hi, mid_lo,
          mid_hi, _stream())
    return out


def patch_embed(x, wt, bias, gamma, beta, t, eps=1e-5):
    """x (B,T,3,H,W), wt (48t, C) -> (B, t_out*H/4*W/4, C)."""
    x = _chk(x, "x")
    b, tt, _, h, w = x.shape
    c = wt.shape[1]
    t_out = (tt - t) // t + 1
    out = torch.empty(b, t_out * (h // 4) * (w // 4), c, device=x.device, dtype=torch.float32)
    _call("mumpy_patch_embed_fwd", _p(x), _p(_chk(wt, "wt")), _p(_chk(bias, "bias")), _p(_chk(gamma, "gamma")),
          _p(_chk(beta, "beta")), _p(out), b, tt, h, w, t, c, eps, _stream())
    return out


def patch_merge_ln(x, gamma, beta, b, hs, w, c, eps=1e-5):
    x = _chk(x, "x")
    out = torch.empty(b, (hs // 2) * (w // 2), 4 * c, device=x.device, dtype=torch.float32)
    _call("mumpy_patch_merge_ln_fwd", _p(x), _p(_chk(gamma, "gamma")), _p(_chk(beta, "beta")), _p(out), b, hs, w, c, eps,
          _stream())
    return out


def temporal_attention(qkv, s, t, c, heads, scale, tq=None):
    """tq: number of leading temporal tokens that are queries (default all): out (s, tq, c)."""
    qkv = _chk(qkv, "qkv")
    tq = t if tq is None else tq
    out = torch.empty(s, tq, c, device=qkv.device, dtype=torch.float32)
    _call("mumpy_temporal_attention_q_fwd", _p(qkv), _p(out), s, t, tq, c, heads, scale, _stream())
    return out


def attention_probs(q, k, outer, heads, nq, nk, d, q_strides, k_strides, scale, q_mod=None):
    """softmax(scale * q k^T) maps (outer, heads, nq, nk) for the `return_attention=True` variants; q_strides / k_strides =
    (outer stride, row stride) in floats, head h adds h * d; q_mod: q outer index = outer % q_mod."""
    for t in (q, k):                      # views into a larger tensor are fine: addressing is by the strides given (no copy)
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError("mumpy_hip: attention_probs needs float32 GPU tensors (there is no CPU path)")
    out = torch.empty(outer, heads, nq, nk, device=q.device, dtype=torch.float32)
    _call("mumpy_attention_probs_fwd", q.data_ptr(), k.data_ptr(), _p(out), outer, heads, nq, nk, d, q_strides[0], q_strides[1], k_strides[0],
          k_strides[1], outer if q_mod is None else q_mod, scale, _stream())
    return out


def sigmoid_threshold(logits, thr=0.5):
    logits = _chk(logits, "logits")
    mask = torch.empty(logits.shape, device=logits.device, dtype=torch.uint8)
    _call("mumpy_sigmoid_threshold_fwd", _p(logits), _p(mask), logits.numel(), thr, _stream())
    return mask


EVAL_MEAN, EVAL_STD = (0.4776, 0.479, 0.4465), (0.230, 0.2085, 0.2324)      # test.py:23-24


_NEAREST_TABLES = {}


def _nearest_table(src: int, dst: int, device) -> torch.Tensor:
    """Pillow's NEAREST source indices for a src -> dst resize (built by the library's host helper, cached on the device)."""
    import ctypes
    key = (src, dst, str(device))
    t = _NEAREST_TABLES.get(key)
    if t is None:
        host = (ctypes.c_int32 * dst)()
        rc = _lib().mumpy_resize_nearest_table(src, dst, host)
        if rc:
            raise RuntimeError(f"mumpy_resize_nearest_table failed ({rc}): {_lib().mumpy_last_error().decode()}")
        t = _NEAREST_TABLES[key] = torch.tensor(list(host), dtype=torch.int32, device=device)
    return t


def normalize_u8(frames, mean=EVAL_MEAN, std=EVAL_STD, size=None):
    """frames (..., H, W, 3) uint8 on the GPU -> (..., 3, H, W) float32, ToTensor + Normalize (test.py:22-25).
    size=(H_out, W_out): the loader's `img.resize(inputRes)` (universaldataset.py:75-79, PIL NEAREST as in the pinned
    pillow==4.0.0) runs in the same kernel, e.g. size=(224, 224) for 432x240 footage."""
    import ctypes
    if not frames.is_cuda or frames.dtype != torch.uint8 or frames.shape[-1] != 3:
        raise RuntimeError("mumpy_hip: normalize_u8 needs a uint8 GPU tensor (..., H, W, 3) (there is no CPU path)")
    frames = frames.contiguous()
    lead, (h, w) = frames.shape[:-3], frames.shape[-3:-1]
    n = 1
    for d in lead:
        n *= d
    if size is not None and tuple(size) != (h, w):
        ho, wo = size
        out = torch.empty(*lead, 3, ho, wo, device=frames.device, dtype=torch.float32)
        m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
        ytab, xtab = _nearest_table(h, ho, frames.device), _nearest_table(w, wo, frames.device)
        _call("mumpy_resize_normalize_u8_fwd", _p(frames), _p(out), _p(ytab), _p(xtab), n, h, w, ho, wo, m3, s3, _stream(),
              work=float(5 * out.numel()))
        return out
    out = torch.empty(*lead, 3, h, w, device=frames.device, dtype=torch.float32)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    _call("mumpy_normalize_u8_fwd", _p(frames), _p(out), n, h, w, m3, s3, _stream(), work=float(frames.numel() + 4 * out.numel()))
    return out


# ---------------------------------------------------------------------------------------------- training tail (8f-2)
def mask_loss(logits, target, need_grad=True, eps=0.0, loss_scale=1.0):
    """softIoULoss + WeightedFocalLoss as train.py:107-113 calls them (utils/loss.py:6-55).  logits (B,...) and 0/1 target of the
    same number of elements per sample -> (loss3 = [total*loss_scale, iou, focal] device tensor, dlogits or None)."""
    logits = _chk(logits, "logits")
    b = logits.shape[0]
    p = logits.numel() // b
    target = _chk(target.to(torch.float32).reshape(b, p), "target")
    loss3 = torch.empty(3, device=logits.device, dtype=torch.float32)
    dz = torch.empty_like(logits) if need_grad else None
    wsb = int(_lib().mumpy_mask_loss_workspace_bytes(b, p))
    ws = torch.empty(wsb // 4, device=logits.device, dtype=torch.float32)
    _call("mumpy_mask_loss_fwd_bwd", _p(logits), _p(target), _p(dz), _p(loss3), _p(ws), wsb, b, p, eps, loss_scale, _stream(),
          work=4.0 * logits.numel() * (5 if need_grad else 2))
    return loss3, dz


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    """In-place fused AdamW over flat fp32 buffers (torch.optim.AdamW defaults; step counts from 1)."""
    n = param.numel()
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _chk(t, name)
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError(f"adamw_step: {name} must be a contiguous buffer of {n} elements (in-place update)")
    _call("mumpy_adamw_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lr, betas[0], betas[1], eps, weight_decay,
          int(step), grad_scale, _stream(), work=28.0 * n)
    return param


# ------------------------------------------------------------------------------------------ Swin block backward (8f-2)
def _ws(nbytes, device):
    return torch.empty(max(nbytes // 4, 4), device=device, dtype=torch.float32)


def layernorm_bwd(x, gamma, dy, eps=1e-5, dx_add=None, dg_out=None, db_out=None):
    """-> (dx like x, dgamma (C), dbeta (C)) of nn.LayerNorm over the last dim.  dx_add: a gradient to add to dx (the residual
    branch that bypasses the norm).  dg_out / db_out (both or neither): gradient buffers to ACCUMULATE into; the matching
    return values are then None."""
    x, dy, gamma = _chk(x, "x"), _chk(dy, "dy"), _chk(gamma, "gamma")
    c = x.shape[-1]
    rows = x.numel() // c
    dx = torch.empty_like(x)
    if dx_add is not None:
        dx_add = _chk(dx_add, "dx_add")
        if dx_add.shape != x.shape:
            raise RuntimeError("layernorm_bwd: dx_add must have x's shape")
    if (dg_out is None) != (db_out is None):
        raise RuntimeError("layernorm_bwd: dg_out and db_out go together")
    acc = dg_out is not None
    dg = _chk(dg_out, "dg_out") if acc else torch.empty(c, device=x.device, dtype=torch.float32)
    db = _chk(db_out, "db_out") if acc else torch.empty(c, device=x.device, dtype=torch.float32)
    key = ("lnbwd", rows, c)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_layernorm_bwd_workspace_bytes(rows, c))
    ws = _ws(wsb, x.device)
    _call("mumpy_layernorm_bwd", _p(x), _p(gamma), _p(dy), _p(dx_add), _p(dx), _p(dg), _p(db), _p(ws), wsb, rows, c, eps, int(acc),
          _stream(), work=12.0 * x.numel())
    return (dx, None, None) if acc else (dx, dg, db)


def gelu(x):
    x = _chk(x, "x")
    y = torch.empty_like(x)
    _call("mumpy_gelu_fwd", _p(x), _p(y), x.numel(), _stream(), work=8.0 * x.numel())
    return y


def gelu_bwd(x, dy):
    x, dy = _chk(x, "x"), _chk(dy, "dy")
    dx = torch.empty_like(x)
    _call("mumpy_gelu_bwd", _p(x), _p(dy), _p(dx), x.numel(), _stream(), work=12.0 * x.numel())
    return dx


def transpose(x2d, pad_rows_to=1):
    """(R,C) -> (C, Rp) with Rp = R rounded up to a multiple of `pad_rows_to` (extra columns zero): the K-contiguous
    operand layout of the weight-gradient GEMMs, whose reduction dim (the token count) must be a multiple of 32."""
    x2d = _chk(x2d, "x")
    r, c = x2d.shape
    rp = (r + pad_rows_to - 1) // pad_rows_to * pad_rows_to
    if rp != r:
        xp = torch.zeros(rp, c, device=x2d.device, dtype=torch.float32)
        xp[:r] = x2d
        x2d, r = xp, rp
    out = torch.empty(c, r, device=x2d.device, dtype=torch.float32)
    _call("mumpy_transpose_fwd", _p(x2d), _p(out), r, c, _stream(), work=8.0 * x2d.numel())
    return out


def linear_bwd(x2d, weight, dy2d, need_dx=True, dw_out=None, db_out=None, need_dw=True, need_db=False):
    """Backward of y = x W^T + b in ONE C-ABI call (mumpy_linear_bwd), no transposed copies: -> (dx, dW, db).
    dw_out / db_out: gradient buffers to ACCUMULATE into (e.g. views of FlatAdamW's flat gradient); the matching return
    value is then None (nothing left for autograd to add)."""
    dy2d = _chk(dy2d, "dy")
    if x2d is None and weight is None:                           # bias gradient only (column sums of dY, e.g. a convolution's bias)
        if need_dx or need_dw or not need_db:
            raise RuntimeError("linear_bwd: without x and weight only the bias gradient can be asked for")
        m, n = dy2d.shape
        k = 32
    else:
        x2d, weight = _chk(x2d, "x"), _chk(weight, "weight")
        m, k = x2d.shape
        n = weight.shape[0]
        if dy2d.shape != (m, n) or weight.shape[1] != k:
            raise RuntimeError(f"linear_bwd: x {tuple(x2d.shape)}, weight {tuple(weight.shape)}, dy {tuple(dy2d.shape)} do not match")
    dev = dy2d.device
    dx = torch.empty(m, k, device=dev, dtype=torch.float32) if need_dx else None
    acc = 0
    dw = db = None
    if need_dw:
        if dw_out is not None:
            dw, acc = _chk(dw_out, "dw_out"), acc | 1
        else:
            dw = torch.empty(n, k, device=dev, dtype=torch.float32)
    if need_db:
        if db_out is not None:
            db, acc = _chk(db_out, "db_out"), acc | 2
        else:
            db = torch.empty(n, device=dev, dtype=torch.float32)
    key = ("lbwd", m, n, k)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_linear_bwd_workspace_bytes(m, n, k))
    ws = _ws(wsb, dev)
    if _MATH == MATH_BF16:
        acc |= MATH_BF16                                      # bf16 operands on the bf16 MFMA, fp32 accumulate (config 5's arithmetic)
    elif _MATH != MATH_FP32:
        raise RuntimeError("linear_bwd: the split-precision modes have no one-call backward (autograd routes them through linear)")
    _call("mumpy_linear_bwd", _p(x2d), _p(weight), _p(dy2d), _p(dx), _p(dw), _p(db), m, n, k, acc, _p(ws), wsb, _stream(),
          work=2.0 * m * n * k * (int(need_dx) + int(need_dw)))
    return dx, (None if dw_out is not None else dw), (None if db_out is not None else db)


def final_conv_bwd(x, w_krsc, dy):
    """Backward of final_conv for C = 32: x logical (B,32,H,W) NHWC, w_krsc (1,3,3,32), dy (B,1,H,W) ->
    (dx like x, dw (1,3,3,32), db (1,)).  One streaming pass + a small fixed-tree reduce."""
    x = _nhwc(x, "x")
    b, c, h, w = x.shape
    dy = _chk(dy.reshape(b, h, w), "dy")
    if c != 32 or tuple(w_krsc.shape) != (1, 3, 3, 32):
        raise RuntimeError(f"final_conv_bwd is built for Conv2d(32, 1, 3, padding=1); got C={c}, weight {tuple(w_krsc.shape)}")
    dx = empty_nhwc(b, c, h, w, x.device)
    dw = torch.empty(1, 3, 3, 32, device=x.device, dtype=torch.float32)
    db = torch.empty(1, device=x.device, dtype=torch.float32)
    wsb = int(_lib().mumpy_final_conv_bwd_workspace_bytes(b, h, w))
    ws = _ws(wsb, x.device)
    _call("mumpy_final_conv_bwd", _p(x), _p(_chk(w_krsc, "weight")), _p(dy), _p(dx), _p(dw), _p(db), _p(ws), wsb, b, h, w, c, _stream(),
          work=4.0 * (2 * x.numel() + dy.numel()))
    return dx, dw, db


def patch_gather(x, b, h, w, c, inverse=False):
    """PatchMerging's 2x2 gather (swin:357-361): x (b,h,w,c) -> (b, h/2 * w/2, 4c); inverse=True maps a merged-layout tensor back to
    (b, h*w, c) (the gather's backward) -- one permutation launch either way."""
    x = _chk(x, "x")
    if x.numel() != b * h * w * c:
        raise RuntimeError(f"patch_gather: {tuple(x.shape)} is not {b} x {h} x {w} x {c}")
    out = torch.empty((b, h * w, c) if inverse else (b, (h // 2) * (w // 2), 4 * c), device=x.device, dtype=torch.float32)
    _call("mumpy_patch_gather_fwd", _p(x), _p(out), b, h, w, c, int(inverse), _stream(), work=8.0 * x.numel())
    return out


def conv_weight_dgrad(w_krsc):
    """(Cout,kh,kw,Cin) -> (Cin,kh,kw,Cout) with the taps flipped: the weight of the data-gradient convolution, one launch."""
    w_krsc = _chk(w_krsc, "weight")
    cout, kh, kw, cin = w_krsc.shape
    out = torch.empty(cin, kh, kw, cout, device=w_krsc.device, dtype=torch.float32)
    _call("mumpy_conv_weight_dgrad_fwd", _p(w_krsc), _p(out), cout, cin, kh, kw, _stream(), work=8.0 * out.numel())
    return out


def conv2d_wgrad(x, dy, kh, kw, dw_out=None):
    """Weight gradient of conv2d_nhwc in one launch over all taps: x logical (B,Cin,H,W), dy logical (B,Cout,H,W), both NHWC
    memory -> dW (Cout,kh,kw,Cin).  dw_out: a buffer to ACCUMULATE into (returns None then)."""
    x, dy = _nhwc(x, "x"), _nhwc(dy, "dy")
    b, cin, h, w = x.shape
    cout = dy.shape[1]
    if dy.shape != (b, cout, h, w):
        raise RuntimeError(f"conv2d_wgrad: x {tuple(x.shape)} and dy {tuple(dy.shape)} do not match")
    acc = dw_out is not None
    dw = _chk(dw_out, "dw_out") if acc else torch.empty(cout, kh, kw, cin, device=x.device, dtype=torch.float32)
    if dw.shape != (cout, kh, kw, cin):
        raise RuntimeError(f"conv2d_wgrad: gradient buffer {tuple(dw.shape)} != {(cout, kh, kw, cin)}")
    key = ("cwgrad", b, h, w, cin, cout, kh, kw)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_conv2d_wgrad_workspace_bytes(b, h, w, cin, cout, kh, kw))
    ws = _ws(wsb, x.device) if wsb else None
    if _MATH not in (MATH_FP32, MATH_BF16):
        raise RuntimeError("conv2d_wgrad: the split-precision modes have no one-launch weight gradient")
    _call("mumpy_conv2d_wgrad_nhwc", _p(x), _p(dy), _p(dw), b, h, w, cin, cout, kh, kw, int(acc) | (MATH_BF16 if _MATH == MATH_BF16 else 0),
          _p(ws), wsb, _stream(),
          work=2.0 * b * h * w * cout * kh * kw * cin)
    return None if acc else dw


def col_sum(x2d):
    x2d = _chk(x2d, "x")
    r, c = x2d.shape
    out = torch.empty(c, device=x2d.device, dtype=torch.float32)
    wsb = int(_lib().mumpy_col_sum_workspace_bytes(r, c))
    ws = _ws(wsb, x2d.device)
    _call("mumpy_col_sum_fwd", _p(x2d), _p(out), _p(ws), wsb, r, c, _stream(), work=4.0 * x2d.numel())
    return out


def rel_index_csr(index: torch.Tensor) -> torch.Tensor:
    """Inverse of a `relative_position_index` buffer for the table-gradient kernel: int32 [ptr (170) | pairs (2401)], the pairs
    p = 49 i + j of one table entry contiguous and in increasing order.  Built once (host side, one synchronisation) and kept ON the
    buffer object like rel_index32's image: call it from the forward (eager warm-up), never for the first time under graph capture."""
    t = getattr(index, "_mumpy_csr", None)
    if t is None or t.device != index.device or getattr(index, "_mumpy_csr_version", -1) != index._version:
        idx = index.reshape(-1).to("cpu", torch.int64)
        if idx.numel() != 49 * 49 or int(idx.min()) < 0 or int(idx.max()) >= 169:
            raise RuntimeError("rel_index_csr: expected a (49,49) relative_position_index with values in [0, 169)")
        order = torch.sort(idx, stable=True).indices
        ptr = torch.zeros(170, dtype=torch.int64)
        ptr[1:] = torch.cumsum(torch.bincount(idx, minlength=169), 0)
        t = torch.cat([ptr, order]).to(torch.int32).to(index.device)
        index._mumpy_csr, index._mumpy_csr_version = t, index._version
    return t


def window_attention_bwd(qkv, dout, bias_pad, rel_index32, b, hs, w, c, shift, scale, mask_tab=None, mask_id=None, dtable_out=None,
                         rel_csr=None):
    """-> (dqkv (B, hs*w, 3C), dtable (169, C/32)): gradients of the W-MSA core wrt qkv and the relative position bias table.
    dtable_out: a gradient buffer to ACCUMULATE into (the returned dtable is then None).  rel_csr = rel_index_csr(index): the table
    gradient reads its pairs through the inverse index instead of scanning the index."""
    qkv, dout = _chk(qkv, "qkv"), _chk(dout, "dout")
    if qkv.numel() != b * hs * w * 3 * c or dout.numel() != b * hs * w * c:
        raise RuntimeError("window_attention_bwd: shape mismatch")
    if rel_index32.dtype != torch.int32 or rel_index32.numel() != 49 * 49 or not rel_index32.is_cuda:
        raise RuntimeError("window_attention_bwd: rel_index32 must be the (49*49) int32 relative_position_index on the GPU")
    dqkv = torch.empty_like(qkv)
    acc = dtable_out is not None
    dtable = _chk(dtable_out, "dtable_out") if acc else torch.empty(169, c // 32, device=qkv.device, dtype=torch.float32)
    if dtable.shape != (169, c // 32):
        raise RuntimeError(f"window_attention_bwd: table gradient buffer {tuple(dtable.shape)} != {(169, c // 32)}")
    key = ("wabwd", b, hs, w, c)
    wsb = _WS_BYTES.get(key)
    if wsb is None:
        wsb = _WS_BYTES[key] = int(_lib().mumpy_window_attention_bwd_workspace_bytes(b, hs, w, c))
    ws = _ws(wsb, qkv.device)
    n_mask = 0 if mask_id is None else mask_id.numel()
    if rel_csr is not None and (rel_csr.dtype != torch.int32 or rel_csr.numel() != 170 + 49 * 49 or rel_csr.device != qkv.device):
        raise RuntimeError("window_attention_bwd: rel_csr must come from ops.rel_index_csr on this device")
    csr_args = () if rel_csr is None else (_p(rel_csr),)
    _call("mumpy_window_attention_bwd" + ("" if rel_csr is None else "_csr"), _p(qkv), _p(dout), _p(_chk(bias_pad, "bias")), _p(mask_tab),
          _p(mask_id), n_mask, _p(rel_index32.contiguous()), *csr_args, _p(dqkv), _p(dtable), _p(ws), wsb, b, hs, w, c, shift, scale, int(acc), _stream(),
          work=5 * 153664.0 * b * (hs // 7) * (w // 7) * (c // 32))
    return dqkv, (None if acc else dtable)


GN_ACCUMULATE = 0x100     # mumpy_hip.h: MUMPY_GN_ACCUMULATE


def gn_bwd(z, stats, gamma, beta, dy, groups, eps=1e-5, act=ACT_RELU, dg_out=None, db_out=None):
    """GroupNorm(+activation) backward on NHWC: z, dy logical (B,C,H,W) with NHWC memory; stats = (partial, nsplit) from
    gn_stats(z); act = 0 / ACT_RELU / ACT_SIGMOID is the activation that followed the norm.  -> (dz like z, dgamma, dbeta);
    with dg_out AND db_out the parameter gradients are ACCUMULATED into those buffers (grad slots) and returned as None."""
    z, dy = _nhwc(z, "z"), _nhwc(dy, "dy")
    b, c, h, w = z.shape
    partial, nsplit = stats
    dz = empty_nhwc(b, c, h, w, z.device)
    acc = dg_out is not None and db_out is not None
    dg = _chk(dg_out, "dgamma") if acc else torch.empty(c, device=z.device, dtype=torch.float32)
    db = _chk(db_out, "dbeta") if acc else torch.empty(c, device=z.device, dtype=torch.float32)
    if dg.numel() != c or db.numel() != c:
        raise RuntimeError("gn_bwd: parameter gradient buffers must hold C values")
    wsb = int(_lib().mumpy_gn_bwd_workspace_bytes(b, h * w, c))
    ws = _ws(wsb, z.device)
    _call("mumpy_gn_bwd_nhwc", _p(z), _p(partial), nsplit, _p(_chk(gamma, "gamma")), _p(_chk(beta, "beta")), _p(dy), _p(dz), _p(dg),
          _p(db), _p(ws), wsb, b, h * w, c, groups, eps, int(act) | (GN_ACCUMULATE if acc else 0), _stream(), work=20.0 * z.numel())
    return (dz, None, None) if acc else (dz, dg, db)


def upsample_bwd(dy, scale=2, align_corners=True):
    """dy logical (B,C,sH,sW) NHWC -> dx (B,C,H,W) NHWC: backward of the bilinear x2 / x4 upsample."""
    dy = _nhwc(dy, "dy")
    b, c, ho, wo = dy.shape
    dx = empty_nhwc(b, c, ho // scale, wo // scale, dy.device)
    _call("mumpy_upsample_bwd_nhwc", _p(dy), _p(dx), b, ho // scale, wo // scale, c, scale, 1 if align_corners else 0, _stream(),
          work=4.0 * (dy.numel() + dx.numel()))
    return dx


def upsample2x_bwd(dy, align_corners=True):
    return upsample_bwd(dy, 2, align_corners)


def scale_samples(x, scale):
    """out[b] = x[b] * scale[b] (stochastic depth); x (B, ...), scale (B,) on the GPU."""
    x, scale = _chk(x, "x"), _chk(scale, "scale")
    b = x.shape[0]
    if scale.numel() != b:
        raise RuntimeError("scale_samples: one scale per sample")
    out = torch.empty_like(x)
    _call("mumpy_scale_samples_fwd", _p(x), _p(scale), _p(out), b, x.numel() // b, _stream(), work=8.0 * x.numel())
    return out


def temporal_attention_bwd(qkv, dout, s_, t, c, heads, scale):
    qkv, dout = _chk(qkv, "qkv"), _chk(dout, "dout")
    dqkv = torch.empty_like(qkv)
    _call("mumpy_temporal_attention_bwd", _p(qkv), _p(dout), _p(dqkv), s_, t, c, heads, scale, _stream(), work=4.0 * (2 * qkv.numel() + dout.numel()))
    return dqkv


# ------------------------------------------------------------------------------- deformable attention, training (row 10)
def dwconv5_window(x, w25, b):
    """x (N,49,C) token-major windows, w25 (C,25), b (C) -> (N,49,C): depthwise 5x5 conv (padding 2) inside each 7x7 window."""
    x = _chk(x, "x")
    n, _, c = x.shape
    u = torch.empty_like(x)
    _call("mumpy_dwconv5_window_fwd", _p(x), _p(_chk(w25, "w")), _p(_chk(b, "b")), _p(u), n, c, _stream(), work=8.0 * x.numel())
    return u


def dwconv5_window_bwd(x, w25, du):
    """-> (dx (N,49,C), dw (C,25), db (C))."""
    x, du = _chk(x, "x"), _chk(du, "du")
    n, _, c = x.shape
    dx = torch.empty_like(x)
    dw_t = torch.empty(26, c, device=x.device, dtype=torch.float32)       # rows 0..24 = taps; row 25 unused scratch
    db = torch.empty(c, device=x.device, dtype=torch.float32)
    wsb = int(_lib().mumpy_dwconv5_window_bwd_workspace_bytes(n, c))
    ws = _ws(wsb, x.device)
    _call("mumpy_dwconv5_window_bwd", _p(x), _p(_chk(w25, "w")), _p(du), _p(dx), _p(dw_t), _p(db), _p(ws), wsb, n, c, _stream(),
          work=16.0 * x.numel())
    return dx, transpose(dw_t[:25].contiguous()), db


def deform_sample_bwd(x2w, pos, dsampled):
    """window form: x2w, dsampled (B2,49,C), pos (nq,3,49,2) -> (dx2 (B2,49,C), dpos (nq,3,49,2))."""
    x2w, pos, dsampled = _chk(x2w, "x2"), _chk(pos, "pos"), _chk(dsampled, "dsampled")
    b2, _, c = x2w.shape
    nq = pos.shape[0]
    dx2 = torch.empty_like(x2w)
    part = torch.empty(b2, 3, 49, 2, device=x2w.device, dtype=torch.float32)
    _call("mumpy_deform_sample_bwd", _p(x2w), _p(pos), _p(dsampled), _p(dx2), _p(part), b2, c, nq, _stream(), work=12.0 * x2w.numel())
    # kv windows qw + m*nq share q window qw: a (r, nq, ...) view summed over r (a few KB: left to torch)
    return dx2, part.view(b2 // nq, nq, 3, 49, 2).sum(0)


def deform_attention_bwd(q, kv, dout, r, scale):
    """window form: q (B1,49,C), kv (B1*r,49,2C), dout (B1,49,C) -> (dq (B1,49,C), dkv (B1*r,49,2C))."""
    q, kv, dout = _chk(q, "q"), _chk(kv, "kv"), _chk(dout, "dout")
    b1, _, c = q.shape
    b2 = b1 * r
    part = torch.empty(b2, 49, c, device=q.device, dtype=torch.float32)
    dkv = torch.empty_like(kv)
    wsb = int(_lib().mumpy_deform_attention_bwd_workspace_bytes(b2, c))
    ws = _ws(wsb, q.device)
    _call("mumpy_deform_attention_bwd", _p(q), _p(kv), _p(dout), _p(part), _p(dkv), _p(ws), wsb, b1, r, c, scale, _stream(),
          work=5 * 153664.0 * b2 * (c // 32))
    dq = part[:b1]
    for m in range(1, r):                                                 # kv windows qw + m*B1 pair with q window qw: fixed order
        dq = add(dq.contiguous(), part[m * b1:(m + 1) * b1].contiguous())
    return dq.contiguous(), dkv


def adamw_hyper(step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    """The 8 fp32 constants of one AdamW step as a pinned host tensor (for adamw_step_dev under hipGraph replay)."""
    import ctypes
    out = torch.empty(8, dtype=torch.float32).pin_memory()
    rc = _lib().mumpy_adamw_hyper(ctypes.cast(out.data_ptr(), ctypes.POINTER(ctypes.c_float)), lr, betas[0], betas[1], eps,
                                  weight_decay, int(step), grad_scale)
    if rc != 0:
        raise RuntimeError(f"mumpy_adamw_hyper failed (rc={rc}): {_lib().mumpy_last_error().decode()}")
    return out


def adamw_step_dev(param, grad, exp_avg, exp_avg_sq, hyper_dev):
    """AdamW over flat buffers with the step constants in the device tensor `hyper_dev` (8 floats): capturable."""
    n = param.numel()
    for name, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        _chk(t, name)
        if t.numel() != n or not t.is_contiguous():
            raise RuntimeError(f"adamw_step_dev: {name} must be a contiguous buffer of {n} elements (in-place update)")
    _call("mumpy_adamw_step_dev", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(_chk(hyper_dev, "hyper")), _stream(),
          work=28.0 * n)
    return param
