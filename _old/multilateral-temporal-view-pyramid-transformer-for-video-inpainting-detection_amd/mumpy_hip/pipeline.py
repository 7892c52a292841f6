"""Encoder + decoder as ONE schedule (opt-in; the plain Encoder()/Decoder() calls stay sequential and safe).

The 12 global temporal ViT blocks (M = B*49*T rows: under-filled GEMMs) depend only on the last stage's tokens, and
everything in the decoder except gcm1 and the decoder_2..5 trunk depends only on the per-stage features and the DCT
map.  So after the pyramid stages the graph forks:   [ global blocks ]  ||  [ decoder branches ]   -> join -> trunk.
Bitwise identical to Decoder()(*Encoder()(x)) (same kernels, no atomics)."""
import torch

from .streams import run_parallel


@torch.no_grad()
def fused_forward(encoder, decoder, x, with_mask=False, thr=0.5):
    """encoder: models.encoder.encoder.Encoder, decoder: models.decoder.decoder.Decoder.
    -> (logits, feats) or (logits, uint8 mask, feats)."""
    base = encoder.base
    views, view_x, ffinfo = base.forward_stages(x)
    flat = [t for stage in view_x for t in stage]
    # the global blocks (a plain chain) go to the side stream; the decoder branches, which fork again, stay on the
    # current stream so that every nested fork is rooted on it (forking from a side stream inside hipGraph capture
    # crashed the ROCm 7.2 runtime)
    (tokens,), br = run_parallel([lambda: (base.forward_global(views),), lambda: decoder._branches(view_x, ffinfo)],
                                 [views, flat + [ffinfo]])
    b, _, c = tokens.shape
    final_x = tokens.reshape(b, 7, 7, c).permute(0, 3, 1, 2)              # encoder.py:16-17
    feats = decoder._trunk(final_x, br)
    from . import ops
    if with_mask:
        logits, mask = ops.final_conv(feats, decoder._final_weight(), decoder.final_out.bias, with_mask=True, thr=thr)
        return logits, mask, feats
    return ops.final_conv(feats, decoder._final_weight(), decoder.final_out.bias), feats
