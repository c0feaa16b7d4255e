"""Locating / creating model weights for the plugin constructors.

The reference loads `.tflite` / `.pb` blobs (all absent from its tree, .MISSING_LARGE_BLOBS).  This
build reads its own `.npz` of named f32 arrays (names as in deepdish_amd/nets.py).  A model path of
the form `synthetic[:seed]...`, or any path when DEEPDISH_SYNTHETIC_WEIGHTS=1, yields seeded random
weights of the right architecture (what tests and bench.py use).  SSD-MobileNet `.tflite` files (uint8 or
float), MARS encoder `.tflite` files (float) and YOLOv5s `.tflite` files are read by tools/tflite_reader.py, frozen MARS `.pb` graphs by
tools/graphdef.py; other graphs are an explicit error.

`load_ssd_model` is what the SSD-MobileNet plugins call: it returns ('uint8', QModel) for the reference's own kind
of file -- a uint8-quantised model (tools/ssd_mobilenet.py:102 upstream; `synthetic...uint8` names build one from
the seeded f32 weights with deepdish_amd/quantize.py, `...sym...` pins the weight zero points at 128) -- and
('f32', named weights) otherwise.
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
    hint = ''
    if name.endswith('.tflite'):
        hint = (' A .tflite file is parsed by deepdish_amd/tools/tflite_reader.py (tensors, buffers, quantisation, operators); the '
                'SSD-MobileNet-v1 detector (tools/weights_io.load_ssd_model), the MARS encoder (load_mars_weights) and the YOLOv5s detector '
                '(load_yolov5_weights) graphs are mapped onto programs; this path was asked for a model kind that takes none of them.')
    elif name.endswith('.pb'):
        hint = ' Frozen TensorFlow graphs (.pb) are read for the MARS encoder only (tools/graphdef.py, tools/generate_detections.py:118-148 upstream).'
    raise FileNotFoundError(
        '%s: cannot load model weights (%s). Supply an .npz of named arrays, or use a "synthetic[:seed]" '
        'model path / DEEPDISH_SYNTHETIC_WEIGHTS=1 for seeded random weights.%s' % (model_file, name, hint))


def load_mars_weights(model_file):
    """What the encoder constructors call: a `.tflite` file on disk goes through tools/tflite_reader.load_mars (the reference's
    `mars-64x32x3.tflite`, generate_detections.py:151-162), anything else through load_named_weights."""
    from .. import nets
    name = str(model_file)
    if name.endswith('.tflite') and os.path.exists(name):
        from . import tflite_reader
        return tflite_reader.load_mars(name)
    if name.endswith('.pb') and os.path.exists(name):            # a frozen graph (generate_detections.py:118-148,187-189: ImageEncoder)
        from . import graphdef
        return graphdef.load_mars(name)[0]                       # '__in_hw__' carries the crop size its placeholder states
    return load_named_weights(model_file, nets.synthetic_mars_weights)


def load_yolov5_weights(model_file):
    """What the YOLOv5 plugins call: a `.tflite` file on disk goes through tools/tflite_reader.load_yolov5s (the reference's
    `detectors/yolov5/yolov5s-fp16.tflite`, tools/yolov5.py:59-79), anything else through load_named_weights."""
    from .. import nets
    name = str(model_file)
    if name.endswith('.tflite') and os.path.exists(name):
        from . import tflite_reader
        return tflite_reader.load_yolov5s(name)
    return load_named_weights(model_file, nets.synthetic_yolov5s_weights)


def load_ssd_model(model_file):
    """-> ('uint8', QModel) | ('f32', {name: f32 array})."""
    from .. import nets, quantize
    name = str(model_file)
    if name.endswith('.tflite') and os.path.exists(name):
        from . import tflite_reader
        return tflite_reader.load_ssd_mobilenet(name)
    m = re.search(r'synthetic(?::|-seed)?(\d+)?', name)
    if m and ('uint8' in name or 'quant' in name):
        seed = int(m.group(1)) if m.group(1) else 1234
        return 'uint8', quantize.synthetic_ssd_quant_model(seed, symmetric_weights='sym' in name)
    return 'f32', load_named_weights(model_file, nets.synthetic_ssd_weights)


def ssd_post_options(model):
    """TFLite_Detection_PostProcess options of a model load_ssd_model returned: what its .tflite file states (tools/tflite_reader.
    ssd_post_options has validated them), else the stock SSD-MobileNet-v1 export's (max_detections 10, nms_score_threshold 1e-8,
    nms_iou_threshold 0.6 -- the values synthetic models and .npz weights run with)."""
    from .tflite_reader import SSD_POST_DEFAULTS
    post = dict(SSD_POST_DEFAULTS)
    src = model.get('post') if isinstance(model, dict) and model.get('kind') == 'ssd_mobilenet_v1_uint8' else model.get('__post__') if isinstance(model, dict) else None
    if src:
        post.update({k: src[k] for k in SSD_POST_DEFAULTS if k in src})
    return post
