"""Checkpoint I/O in the reference's on-disk format and the clip index rule of the data loader (8f-4).

Format (utils/utils.py:264-276): `encoder_{epoch}.pt` / `decoder_{epoch}.pt` (or `encoder.pt` / `decoder.pt`) are plain
`state_dict`s written with `torch.save`, `enc_opt_{epoch}.pt` / `dec_opt_{epoch}.pt` the optimizers' `state_dict`s
(torch.optim.AdamW layout: `mumpy_hip.train.FlatAdamW.state_dict`); a model trained under `nn.DataParallel` carries a
`module.` prefix on every key, which `check_parallel` strips (utils/utils.py:156-176).  The reference keeps a third
optimizer for the cross-view parameters (train.py:211-213) that its own save_checkpoint drops; here it is written too
(`cva_opt_{epoch}.pt`) so that a resumed run continues the same trajectory.  Differences on purpose:
  * files are read with `torch.load(..., weights_only=True)` only — nothing in a checkpoint is executed;
  * the reference's `args.pkl` (a pickled argparse namespace, utils/utils.py:276,319) is never read or written; run
    arguments travel as a JSON sidecar (`args.json`).  Consequence: the reference's own `load_checkpoint`, which
    unpickles `args.pkl` unconditionally (utils/utils.py:319), needs that one file supplied by the user to open a
    directory written here; the four `.pt` files it reads are in its format.
"""
import json
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch


def check_parallel(encoder_dict: Dict[str, torch.Tensor], decoder_dict: Dict[str, torch.Tensor]) -> Tuple[dict, dict]:
    """Strip the DataParallel `module.` prefix.  Like utils/utils.py:156-176 the decision is taken from the FIRST encoder
    key and applied to both dicts; unlike it, a key without the prefix is left alone instead of losing 7 characters."""
    first = next(iter(encoder_dict), "")
    if not first.startswith("module."):
        return encoder_dict, decoder_dict

    def strip(d):
        return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in d.items())
    return strip(encoder_dict), strip(decoder_dict)


def _names(epoch: Optional[int]) -> Tuple[str, str]:
    return (f"encoder_{epoch}.pt", f"decoder_{epoch}.pt") if epoch is not None else ("encoder.pt", "decoder.pt")


def _opt_name(key: str, epoch: Optional[int]) -> str:
    return f"{key}_opt_{epoch}.pt" if epoch is not None else f"{key}_opt.pt"


def save_checkpoint(directory: str, encoder: torch.nn.Module, decoder: torch.nn.Module, epoch: Optional[int] = None,
                    args: Optional[dict] = None, optimizers: Optional[dict] = None) -> None:
    """utils/utils.py:264-276 with JSON instead of pickle.  `optimizers`: the dict of mumpy_hip.train.build_optimizers
    ({"enc", "dec"[, "cva"]}) -> enc_opt / dec_opt / cva_opt files."""
    os.makedirs(directory, exist_ok=True)
    en, dn = _names(epoch)
    torch.save(encoder.state_dict(), os.path.join(directory, en))
    torch.save(decoder.state_dict(), os.path.join(directory, dn))
    for key, opt in (optimizers or {}).items():
        torch.save(opt.state_dict(), os.path.join(directory, _opt_name(key, epoch)))
    if args is not None:
        with open(os.path.join(directory, "args.json"), "w") as f:
            json.dump(args, f, indent=1, sort_keys=True)


def load_optimizer_states(directory: str, epoch: Optional[int] = None, map_location="cpu") -> Dict[str, dict]:
    """{"enc": state_dict, "dec": ..., "cva": ...} for the optimizer files present (weights-only load: tensors, numbers,
    lists and dicts only).  Feed each to FlatAdamW.load_state_dict to resume (train.py:179-188 does this with torch's)."""
    out = {}
    for key in ("enc", "dec", "cva"):
        path = os.path.join(directory, _opt_name(key, epoch))
        if os.path.exists(path):
            out[key] = torch.load(path, map_location=map_location, weights_only=True)
    return out


def load_checkpoint(directory: str, epoch: Optional[int] = None, map_location="cpu") -> Tuple[dict, dict, Optional[dict]]:
    """-> (encoder_dict, decoder_dict, args-or-None), prefixes already stripped; ready for `load_state_dict(strict=True)`
    as test.py:60-61 does."""
    en, dn = _names(epoch)
    enc = torch.load(os.path.join(directory, en), map_location=map_location, weights_only=True)
    dec = torch.load(os.path.join(directory, dn), map_location=map_location, weights_only=True)
    enc, dec = check_parallel(enc, dec)
    args_path = os.path.join(directory, "args.json")
    args = json.load(open(args_path)) if os.path.exists(args_path) else None
    return enc, dec, args


def clip_frame_indices(num_frames: int, length_clip: int) -> List[List[int]]:
    """Frame ids of every clip of a sequence (universaldataloader.py:41-46): one clip per frame, centred on it, `k =
    length_clip // 2` neighbours on each side, indices clamped to the sequence (edge frames repeat).  Note the clip has
    2k+1 frames, i.e. length_clip rounded to odd."""
    k = int(length_clip / 2)
    return [[max(0, min(num_frames - 1, i)) for i in range(idx - k, idx + k + 1)] for idx in range(num_frames)]
