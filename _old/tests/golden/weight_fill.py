"""Deterministic, name-keyed weight fill shared by the golden generator and the tests.

The reference's trained weights (1.03 GB, Google-Drive link in its README) are not
available offline, so every parity fixture is produced with synthetic weights that are
a pure function of the state_dict key:  w[key] = f(crc32(key), shape, kind).
The same function is applied (by name) to the reference model when the goldens are
generated and to the oracle / HIP product when they are checked.

Kinds
  * integer buffers (relative_position_index) and attn_mask buffers: left as built
  * relative_position_bias_table: N(0, 0.2)      (big enough to matter in softmax)
  * 1-D "...weight"  (LayerNorm / GroupNorm gain): 1 + 0.1*N(0,1)
  * 1-D otherwise    (biases):                     0.05*N(0,1)
  * >=2-D            (Linear / Conv kernels):      N(0,1)/sqrt(fan_in)
    -- this includes SwinDAttention.proj_out, which the reference zero-initialises
       (deformableAttention.py:308-309); zero would hide the whole cross-view branch.
"""
import math
import zlib

import torch


def fill_tensor_(name: str, t: torch.Tensor, salt: str = "") -> None:
    if not t.is_floating_point():
        return
    if name.endswith("attn_mask"):
        return
    g = torch.Generator().manual_seed(zlib.crc32((salt + name).encode()))
    r = torch.randn(t.shape, generator=g, dtype=torch.float32)
    if name.endswith("relative_position_bias_table"):
        v = 0.2 * r
    elif t.ndim == 1 and name.endswith("weight"):
        v = 1.0 + 0.1 * r
    elif t.ndim == 1:
        v = 0.05 * r
    else:
        fan_in = 1
        for d in t.shape[1:]:
            fan_in *= d
        v = r / math.sqrt(fan_in)
    with torch.no_grad():
        t.copy_(v.to(t.dtype))


def fill_state_dict_(sd, salt: str = ""):
    """In-place fill of every tensor of a state_dict (params and float buffers)."""
    for k in sorted(sd.keys()):
        fill_tensor_(k, sd[k], salt)
    return sd


def fill_module_(m: torch.nn.Module, salt: str = ""):
    fill_state_dict_(m.state_dict(), salt)   # state_dict tensors alias the module's storage
    return m


def seeded_randn(seed: int, *shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def digest(t: torch.Tensor, n: int = 512):
    """Size-independent fingerprint of a tensor: [mean, std, l2, maxabs] + n strided samples."""
    f = t.detach().reshape(-1).double()
    numel = f.numel()
    idx = (torch.arange(n, dtype=torch.int64) * 2654435761) % numel
    stats = torch.stack([f.mean(), f.std(), f.norm(), f.abs().max()])
    return stats.float().numpy(), f[idx].float().numpy()
