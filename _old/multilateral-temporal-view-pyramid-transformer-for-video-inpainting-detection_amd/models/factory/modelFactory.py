"""Construction API of the Mumpy encoder — same functions, arguments and return values as the reference's
models/factory/modelFactory.py (create_view_config, create_multiswin, create_baseline, load_model_weights), building
the HIP-backed modules.  `create_multiswin(num_frames=T)` additionally builds the T-frame variant the benchmark
configs use (tubelets (T, T-1, 1), token temporal dims [1, 1, T]); T=3 is the reference's own configuration.
"""
import os

import torch

from models.encoder.multiTemporalViewEncoder import ThreeViewSwinTransformer
from models.modules.layers import ConfigDict
from models.modules.swinTransformer import SwinTransformer

RESOLUTIONS = [(56, 56), (28, 28), (14, 14), (7, 7)]


def load_model_weights(model, path, strict=False):
    """Load a state_dict from `path` (reference factory:8-14).  weights_only=True: nothing in the file is executed."""
    state_dict = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(state_dict, strict=strict)
    return model


def create_view_config(hidden_sizes, patches_size, depths, num_heads, mlp_dim, num_frames, input_resolution, temporal_dim,
                       temporal_ratio=None):
    return ConfigDict({
        "hidden_size": hidden_sizes,
        "patches": {"size": patches_size},
        "window_size": 7,
        "depths": depths,
        "num_heads": num_heads,
        "mlp_dim": mlp_dim,
        "num_frames": num_frames,
        "input_resolution": input_resolution,
        "temporal_dim": temporal_dim,
        "temporal_ratio": temporal_ratio or [1] * len(depths),
    })


def multiswin_view_configs(num_frames=3):
    t = num_frames
    return [
        create_view_config([96, 192, 384, 768], (4, 4, t), [2, 2, 6, 2], [3, 6, 12, 24], 768, 1, RESOLUTIONS, 1, [1, 1]),
        create_view_config([96, 192, 384, 768], (4, 4, t - 1), [2, 2, 18, 2], [3, 6, 12, 24], 1536, 1, RESOLUTIONS, 1, [1, t]),
        create_view_config([128, 256, 512, 1024], (4, 4, 1), [2, 2, 18, 2], [4, 8, 16, 32], 3072, t, RESOLUTIONS, t),
    ]


def create_multiswin(num_frames=3, weights="../weights/weight.pth"):
    """-> (model, view_configs).  Like the reference (factory:70-71) this loads `../weights/weight.pth` non-strictly;
    unlike it, a missing file is not fatal (the pretrained file is an external download): the model then keeps its
    random initialisation, and a later strict `load_state_dict` (test.py:60-61) overwrites everything anyway."""
    view_configs = multiswin_view_configs(num_frames)
    global_encoder_config = ConfigDict({"num_heads": 12, "mlp_dim": 3072, "num_layers": 12, "hidden_size": 768,
                                        "merge_axis": "channel", "num_frames": num_frames})
    model = ThreeViewSwinTransformer(view_configs=view_configs, input_token_temporal_dims=[1, 1, num_frames],
                                     global_encoder_config=global_encoder_config)
    if weights and os.path.exists(weights):
        model = load_model_weights(model, weights, strict=False)
    return model, view_configs


def create_baseline(weights="../weughts/weight.pth"):     # sic: the reference's path (factory:90)
    view_config = create_view_config([128, 256, 512, 1024], (4, 4, 3), [2, 2, 18, 2], [4, 8, 16, 32], 3072, 3, RESOLUTIONS, 3)
    model = SwinTransformer(view_config, img_size=224, patch_size=4, in_chans=3, num_classes=0, embed_dim=128,
                            depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4.0, qkv_bias=True,
                            qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1,
                            norm_layer=torch.nn.LayerNorm, ape=False, patch_norm=True)
    if weights and os.path.exists(weights):
        model = load_model_weights(model, weights, strict=True)
    return model
