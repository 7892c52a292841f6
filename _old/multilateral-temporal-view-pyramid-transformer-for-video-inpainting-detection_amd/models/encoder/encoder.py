"""Encoder wrappers — drop-in for the reference's models/encoder/encoder.py (`Encoder()`, `BaselineEncoder()`)."""
from torch import nn

from models.factory.modelFactory import create_baseline, create_multiswin


class Encoder(nn.Module):
    """Encoder() takes no arguments in the reference (test.py:52); `num_frames` (default 3) selects the T-frame
    benchmark variant.  forward(x (B,T,3,224,224)) -> (final_x (B,2304,7,7), view_x[4][3], dct_x (B,9,224,224))."""

    def __init__(self, num_frames=3):
        super().__init__()
        self.base, self.configs = create_multiswin(num_frames)

    def forward(self, x, return_attention=False, layer_id=1):
        ws = self.configs[0]["window_size"]
        final_x, view_x, dct_x = self.base(x)
        if not return_attention:                                   # 'b (h w) c -> b c h w' (encoder.py:16-17)
            b, _, c = final_x.shape
            final_x = final_x.reshape(b, ws, ws, c).permute(0, 3, 1, 2)
        return final_x, view_x, dct_x


class BaselineEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.base = create_baseline()

    def forward(self, x):
        y = self.base(x)
        b, _, c = y.shape
        return y.reshape(b, 7, 7, c).permute(0, 3, 1, 2)
