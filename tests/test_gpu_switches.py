"""GPU: the engine's A/B switches (README table) are read once per process, so each variant runs in its own interpreter;
what they must have in common is the output -- the README's claim is "every variant gives the same bits"."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sha(script, args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', script)] + [str(a) for a in args], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r'sha ([0-9a-f]{16})', r.stdout)
    assert m, r.stdout[-500:]
    return m.group(1)


def test_ssd_forward_switches_give_the_same_bits():
    """SSD forward at 192 frames (weight-stationary layers on): the LDS-DMA ring, two K stages per barrier, register-staged fills,
    64 channels per wave, no pointwise + depthwise fusion, no weight-stationary kernel at all, unfused first layers."""
    base = _sha('time_forward.py', ['ssd', 192], {})
    for env in ({'DD_WS_SPB': '2'}, {'DD_WSR': '1'}, {'DD_WSR': '2'}, {'DD_WS_DW_OFF': '1'}, {'DD_WS': '0'},
                {'DD_SSD_FRONT_UNFUSED': '1', 'DD_DWPW_ROWS_OFF': '1'}):
        assert _sha('time_forward.py', ['ssd', 192], env) == base, env


def test_uint8_ssd_forward_switches_give_the_same_bits():
    """uint8 SSD forward at 96 frames (every fused path on): generic conv instead of the register-filter pointwise kernel, predictors as
    separate ops, 64-pixel tiles in block 1, blocks as depthwise + pointwise ops, no split pointwise filters, the depthwise layers on the
    vector ALU, the front end (first layer + blocks 1 / 2) as three launches instead of the row pipeline."""
    base = _sha('time_forward.py', ['ssd_i8', 96], {})
    for env in ({'DD_Q_PWS': '0'}, {'DD_Q_MERGE_HEADS': '0'}, {'DD_Q_QT128': '0'}, {'DD_Q_SPLITK': '0'}, {'DD_Q_DUP32': '0'}, {'DD_Q_FUSE': '0'}, {'DD_Q_SPLIT_PW': '0'}, {'DD_Q_DW_VALU': '1'}, {'DD_Q_FRONT': '0'}, {'DD_Q_MID': '0'}, {'DD_Q_FRONT': '0', 'DD_Q_MID': '0'},
                {'DD_Q_HEADS_ONE': '0'},                               # the two predictors of a small feature map as two launches of the generic kernel instead of one
                {'DD_Q_FRONT_CHUNK': '40'}):                         # first layer + blocks 1 / 2 over chunks of 40 frames that reuse the intermediate image slots (96 = 40 + 40 + 16)
        assert _sha('time_forward.py', ['ssd_i8', 96], env) == base, env


def test_mars_forward_switches_give_the_same_bits():
    """MARS forward at 1280 crops: pair kernel off, residual units unfused, stem unfused, row kernels off."""
    base = _sha('time_forward.py', ['mars', 1280], {})
    for env in ({'DD_RES_PAIR_OFF': '1'}, {'DD_RES_UNIT_UNFUSED': '1'}, {'DD_STEM_UNFUSED': '1'},
                {'DD_C64_ROWS_OFF': '1', 'DD_S2_ROWS_OFF': '1'}, {'DD_POOL_TILED': '1'},
                {'DD_MARS_PAIR': '0'},                               # conv3_x layer by layer instead of one launch per block (csrc/mars_pair.hip)
                {'DD_MARS_PAIR': '0', 'DD_C64_ROWS_OFF': '1', 'DD_S2_ROWS_OFF': '1'}, {'DD_MARS_PAIR_MIN': '2000'}):
        assert _sha('time_forward.py', ['mars', 1280], env) == base, env
    # conv4_x on the generic kernels (DD_MARS_WS=0) sums its K halves in another order: within the encoder's tolerance, not the same bits
    assert _sha('time_forward.py', ['mars', 1280], {'DD_MARS_WS': '0'}) != base


def test_yolo_and_image_switches_give_the_same_bits():
    """YOLOv5s forward at 3 frames with the Focus slicing as its own launch / as a 16-channel tensor; the batched Lanczos
    stretch in two launches; (the crop kernel's two forms are compared through the pipeline tests' golden scenes)."""
    base = _sha('time_forward.py', ['yolo', 3], {})
    for env in ({'DD_FOCUS_UNFUSED': '1'}, {'DD_YOLO_FOCUS_FUSE': '0'}, {'DD_YOLO_SPP_FUSE': '0'},
                {'DD_C64_STRIPS_MIN': '1'}):                         # the 3x3 64 -> 64 layers as 8-column strips of the weight-stationary row kernel (from 1024 strips by default)
        assert _sha('time_forward.py', ['yolo', 3], env) == base, env
    base = _sha('time_resize.py', [24], {})
    assert _sha('time_resize.py', [24], {'DD_LANCZOS_FUSED': '0'}) == base


def test_ssd_post_process_nms_forms_give_the_same_bits():
    """dd_ssd_postprocess_decoded on the crafted cases of scripts/ssd_post_cases.py: the full sort + nms_lazy_k of rounds 1-4
    (DD_NMS_SELECT=0) against nms_greedy_f32_k."""
    assert _sha('ssd_post_cases.py', [], {'DD_NMS_SELECT': '0'}) == _sha('ssd_post_cases.py', [], {})
