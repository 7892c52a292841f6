"""Checkpoint I/O in the reference's on-disk format and the clip index rule of the data loader (8f-4).

Format (utils/utils.py:264-276): `encoder_{epoch}.pt` / `decoder_{epoch}.pt` (or `encoder.pt` / `decoder.pt`) are plain
`state_dict`s written with `torch.save`; a model trained under `nn.DataParallel` carries a `module.` prefix on every key,
which `check_parallel` strips (utils/utils.py:156-176).  Differences on purpose:
  * files are read with `torch.load(..., weights_only=True)` only — nothing in a checkpoint is executed;
  * the reference's `args.pkl` (a pickled argparse namespace, utils/utils.py:276,319) is never read; run arguments
    travel as a JSON sidecar (`args.json`) instead.
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


def save_checkpoint(directory: str, encoder: torch.nn.Module, decoder: torch.nn.Module, epoch: Optional[int] = None,
                    args: Optional[dict] = None) -> None:
    """utils/utils.py:264-276 minus the optimizer states (forward-only product) and with JSON instead of pickle."""
    os.makedirs(directory, exist_ok=True)
    en, dn = _names(epoch)
    torch.save(encoder.state_dict(), os.path.join(directory, en))
    torch.save(decoder.state_dict(), os.path.join(directory, dn))
    if args is not None:
        with open(os.path.join(directory, "args.json"), "w") as f:
            json.dump(args, f, indent=1, sort_keys=True)


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
