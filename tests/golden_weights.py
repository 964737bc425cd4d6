"""Deterministic weights for golden cases too large to commit as arrays.

Both `tools/make_goldens.py` (which loads them into the reference's model) and the tests (which
load them into the oracle / HIP model) call `make_state(shapes, seed, gain)` so the fixture only
has to hold inputs and expected outputs.  numpy's PCG64 `standard_normal` stream is stable."""
import zlib

import numpy as np


def make_state(shapes, seed, gain=1.0):
    """shapes: {param_name: shape}.  N(0, s^2) with s = gain / sqrt(fan_in) for matrices,
    0.05*gain for vectors / tables; row 0 of the word table is zero (MIND_corpus.py:121-124)."""
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        x = rng.standard_normal(shape).astype(np.float32)
        if 'embedding' in name and len(shape) == 2:
            x *= 0.3 * gain if name.endswith('word_embedding.weight') else 0.1 * gain
            if name.endswith('word_embedding.weight'):
                x[0] = 0
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            x *= gain / np.sqrt(fan_in)
        else:
            x *= 0.05 * gain
        out[name] = x
    return out
