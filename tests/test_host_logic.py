"""CPU: host-side logic of the product that needs no GPU (count-line geometry, box hygiene, model
compiler, synthetic scenes) against the golden vectors / the oracle."""
import os
import warnings
import numpy as np

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_intersection_golden():
    from deepdish_amd.tools.intersection import intersection, any_intersection
    g = np.load(os.path.join(G, 'intersection.npz'))
    for seg, want in zip(g['fixed'], g['fixed_res']):            # the six in-tree asserts of the reference
        assert intersection(*seg) == bool(want)
    assert any_intersection(g['fixed'][0][0], g['fixed'][0][1], g['pts1']) is True
    assert any_intersection(g['fixed'][0][0], g['fixed'][0][1], g['pts2']) is False
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        got = np.array([bool(intersection(*s)) for s in g['segs']])
    np.testing.assert_array_equal(got, g['res'])


def test_clean_boxes_matches_reference_statements():
    from deepdish_amd.pipeline import clean_boxes
    b, l, s = clean_boxes([(-3.7, 10.2, 50.9, 500.0), (100.5, 100.5, 20.2, 30.9), (0, 0, 640, 470), (630, 470, 50, 50)],
                          ['person'] * 4, [0.9, 0.8, 0.7, 0.6], 640, 480)
    assert b == [(0, 10, 50, 470), (100, 100, 20, 30), (630, 470, 10, 10)] and s == [0.9, 0.8, 0.6]
    assert clean_boxes([(1, 2, float('nan'), 4), (1, 2, 3, 4)], ['a', 'b'], [1, 1], 640, 480) == ([], [], [])


def test_crop_box_host_math_matches_oracle():
    """csrc/image.hip crop_box_host is exercised through dd_crop_resize on the GPU; here the
    oracle's replay of generate_detections.py:63-80 is pinned to the survey's probed values."""
    from oracle import image_np
    assert image_np.crop_box(np.array([100, 50, 41, 90]), (64, 32), (480, 640)) == (98, 50, 143, 140)
    assert image_np.crop_box(np.array([100, 50, 40, 91]), (64, 32), (480, 640)) == (97, 50, 142, 141)
    assert image_np.crop_box(np.array([700, 50, 41, 90]), (64, 32), (480, 640)) is None


def test_model_compiler_shapes():
    from deepdish_amd import nets
    p = nets.compile_mars(nets.synthetic_mars_weights(1))
    words, blob = p.serialize()
    assert words[0] == nets.MAGIC and len(p.ops) == 18 and p.tensors[p.out_tensor]['c'] == 128
    assert sum(i['flops'] for i in p.info) > 1.3e8                 # ~0.144 GFLOP / crop
    a, maps = nets.ssd_anchors(300)
    assert a.shape == (1917, 4) and maps == [19, 10, 5, 3, 2, 1]
    wd = nets.synthetic_ssd_weights(1)
    assert abs(sum(v.size for v in wd.values()) - 6.85e6) < 1e5


def test_lanczos_oracle_is_pillow_exact():
    from PIL import Image
    from oracle import image_np
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((75, 75), Image.LANCZOS))
    np.testing.assert_array_equal(image_np.lanczos_resize_u8(img, 75, 75), want)
