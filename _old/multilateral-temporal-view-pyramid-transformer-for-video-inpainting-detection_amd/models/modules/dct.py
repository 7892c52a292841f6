"""FAF: DCT band-pass frequency features (reference models/modules/dct.py:56-79) on the HIP row-block kernel.

Holds no parameters or buffers (like the reference, whose DCT matrix and masks are plain attributes, so FAF adds
nothing to the state_dict).  The encoder only consumes frame index 1 (mTVE:734); `forward_frame` computes exactly
that frame (1/T of the reference's work).  `forward` keeps the reference's all-frames signature.
"""
import numpy as np
import torch
import torch.nn as nn

from mumpy_hip import ops


def DCT_mat(size):
    i = np.arange(size, dtype=np.float64)[:, None]
    j = np.arange(size, dtype=np.float64)[None, :]
    m = np.sqrt(2.0 / size) * np.cos((j + 0.5) * np.pi * i / size)
    m[0, :] = np.sqrt(1.0 / size)
    return m


def generate_filter(start, end, size):
    s = np.add.outer(np.arange(size), np.arange(size))
    return ((s >= start) & (s <= end)).astype(np.float64)


def generate_fine_grained_filter(start, end, size):
    m = np.zeros((size, size), dtype=np.float64)
    if 0 <= start < size and 0 <= end < size:
        m[int(start), int(end)] = 1.0
    return m


class Filter(nn.Module):
    """Band mask holder with the reference's constructor (dct.py:11-39).  In the default configuration FAF uses
    (use_learnable=False, norm=False) it has no parameters and `forward` is `x * base`; FAF's HIP kernel applies the three
    masks inside the DCT pass from the band limits, so this module is API surface (attribute `filters`), not the hot path.
    The learnable / norm variants are never built by the reference's model code and are not provided."""

    def __init__(self, size, band_start, band_end, use_learnable=False, norm=False, fine_grain=False):
        super().__init__()
        if use_learnable or norm:
            raise NotImplementedError("Filter: only the fixed band mask FAF uses (use_learnable=False, norm=False)")
        self.use_learnable, self.norm = False, False
        self.band = (band_start, band_end)
        gen = generate_fine_grained_filter if fine_grain else generate_filter
        self.base = torch.tensor(gen(band_start, band_end, size))        # plain attribute, fp64 like the reference

    def forward(self, x):
        return x * self.base.to(x.device)


class FAF(nn.Module):
    def __init__(self, size=224):
        super().__init__()
        if size != 224:
            raise NotImplementedError("the HIP DCT kernel is tiled for 224 = 7*32")
        self.fn = 3
        self.size = size
        self.filters = nn.ModuleList([Filter(size, 0, size // 2.82), Filter(size, size // 2.82, size // 2),
                                      Filter(size, size * 1, size * 2)])                    # dct.py:66-69; no state
        d = torch.tensor(DCT_mat(size)).float()            # fp64 build, then .float() (dct.py:60)
        self._host = (d.contiguous(), d.t().contiguous())
        self._dev = {}
        # band limits on i+j (dct.py:66-68): low [0, size//2.82], mid [size//2.82, size//2], high [size, 2*size]
        self.lo_hi = int(size // 2.82)
        self.mid_lo = int(size // 2.82)
        self.mid_hi = int(size // 2)

    def _mats(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(t.to(device) for t in self._host)
        return self._dev[key]

    def forward_frame(self, x, frame=1):
        d, dt = self._mats(x.device)
        return ops.faf(x, d, dt, frame, self.lo_hi, self.mid_lo, self.mid_hi)

    def forward(self, x):
        """(B,T,3,224,224) -> (B,T,9,224,224), all frames (reference signature)."""
        return torch.stack([self.forward_frame(x, t) for t in range(x.shape[1])], dim=1)
