"""Deterministic, order-independent parameter fill shared by the fixture
generator (dev container, live reference) and the parity tests (GPU box): the
same (seed, key) always gives the same tensor, so weights never need to be
stored in a fixture."""
import zlib

import numpy as np
import torch


def fill_state(model, seed):
    sd = model.state_dict()
    new = {}
    for key in sorted(sd):
        t = sd[key]
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        if key.endswith('num_batches_tracked'):
            v = np.zeros(t.shape, np.int64)
        elif key.endswith('running_mean'):
            v = rng.normal(0, 0.1, t.shape)
        elif key.endswith('running_var'):
            v = rng.uniform(0.5, 1.5, t.shape)
        elif t.dim() <= 1 and key.endswith('weight'):       # LayerNorm / BatchNorm scale
            v = 1.0 + 0.1 * rng.standard_normal(t.shape)
        elif key.endswith('bias'):
            v = 0.05 * rng.standard_normal(t.shape)
        elif 'token' in key:
            v = 0.5 * rng.standard_normal(t.shape)
        else:                                               # Linear / Conv weights: (out, in, ...)
            fan_in = int(np.prod(t.shape[1:]))
            v = rng.standard_normal(t.shape) / np.sqrt(fan_in)
        new[key] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    model.load_state_dict(new)
    return model
