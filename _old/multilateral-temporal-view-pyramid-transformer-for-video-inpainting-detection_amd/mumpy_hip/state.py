"""Process-wide counter bumped whenever a kernel rewrites parameters in place (FlatAdamW.step): a raw HIP kernel does not
advance torch's per-tensor version counters, so caches of weight-derived tensors key on this epoch as well."""
weights_epoch = [0]


def bump_weights_epoch():
    weights_epoch[0] += 1
