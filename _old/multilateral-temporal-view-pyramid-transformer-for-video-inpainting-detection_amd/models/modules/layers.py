"""Small host-side helpers the reference takes from timm / ml_collections (neither is a dependency here)."""
import collections.abc

import torch
import torch.nn as nn

from mumpy_hip.state import weights_epoch


def to_2tuple(x):
    if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
        return tuple(x)
    return (x, x)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth.  Identity in eval mode (the inference forward).  The training path (mumpy_hip.autograd
    .drop_path_train) reads `drop_prob` and applies the per-sample mask itself; calling this module in train mode with a
    non-zero rate is refused so that a forward-only call cannot silently skip it."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.training and self.drop_prob > 0.0:
            raise NotImplementedError("the inference forward has no stochastic depth: call .eval(), or train through "
                                      "mumpy_hip.autograd (swin_block_train / baseline_encoder_train)")
        return x

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


class ConfigDict(dict):
    """Nested dict with attribute access: the part of ml_collections.ConfigDict the factory and the encoder use
    (item access `cfg["patches"].size`, attribute access `cfg.window_size`)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class Derived:
    """Cache of a tensor derived from parameters/buffers (transposed weights, expanded bias tables, compacted masks);
    recomputed when any source's storage, version or device changes (load_state_dict, .cuda(), a torch optimizer step) or
    when a HIP kernel rewrote parameters in place (mumpy_hip.state.weights_epoch, bumped by FlatAdamW.step)."""

    def __init__(self):
        self._key = None
        self._val = None

    def get(self, sources, fn):
        # the optimizer epoch only matters for sources an optimizer can rewrite (parameters); buffers such as attn_mask keep
        # their cache across steps (and stay host-sync free inside a captured training step)
        epoch = weights_epoch[0] if any(getattr(s, "requires_grad", False) for s in sources) else 0
        key = (epoch,) + tuple((s.data_ptr(), s._version, str(s.device)) for s in sources)
        if key != self._key:
            with torch.no_grad():
                self._val = fn()
            self._key = key
        return self._val
