"""Locating / creating model weights for the plugin constructors.

The reference loads `.tflite` / `.pb` blobs (all absent from its tree, .MISSING_LARGE_BLOBS).  This
build reads its own `.npz` of named f32 arrays (names as in deepdish_amd/nets.py).  A model path of
the form `synthetic[:seed]...`, or any path when DEEPDISH_SYNTHETIC_WEIGHTS=1, yields seeded random
weights of the right architecture (what tests and bench.py use).  Anything else is an error: there
is no TFLite flatbuffer reader here yet.
"""
import os
import re
import numpy as np


def load_named_weights(model_file, synthetic_fn):
    name = os.path.basename(str(model_file))
    m = re.search(r'synthetic(?::|-seed)?(\d+)?', str(model_file))
    if m or os.environ.get('DEEPDISH_SYNTHETIC_WEIGHTS') == '1':
        seed = int(m.group(1)) if (m and m.group(1)) else 1234
        return synthetic_fn(seed)
    if str(model_file).endswith('.npz') and os.path.exists(model_file):
        with np.load(model_file) as z:
            return {k: z[k] for k in z.files}
    raise FileNotFoundError(
        '%s: cannot load model weights (%s). Supply an .npz of named arrays, or use a "synthetic[:seed]" '
        'model path / DEEPDISH_SYNTHETIC_WEIGHTS=1 for seeded random weights.' % (model_file, name))
