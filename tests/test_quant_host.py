"""CPU: host logic of the uint8 detector path -- the fixed-point forms csrc/netsq.hip uses against the literal gemmlowp
statements (oracle/nets_quant.py), the independent SSD anchor generator against the product's, the logistic table, and the
weight packing + per-channel constants replayed in numpy against the oracle's convolution."""
import numpy as np
import pytest


def test_one_step_requantisation_equals_the_two_roundings():
    """z = (x M + 2^30 + 2^(30+e) + (zo << (31+e))) >> (31+e) -- the ReLU-type form of csrc/netsq.hip q_requant -- against
    SaturatingRoundingDoublingHighMul followed by RoundingDivideByPOT, plus the literal two-step form the linear layers take."""
    from oracle import nets_quant as nq
    rng = np.random.default_rng(0)
    for _ in range(200):
        M = int(rng.integers(1 << 30, 1 << 31))
        e = int(rng.integers(0, 14))
        zo = int(rng.integers(0, 256))
        x = np.concatenate([rng.integers(-(1 << 26), 1 << 26, 4000), np.arange(-40, 40), np.array([-(1 << 26), (1 << 26) - 1])]).astype(np.int64)
        want = nq.multiply_by_quantized_multiplier(x, M, -e) + zo
        y = nq.srdhm(x, M)
        # ReLU-type: identical wherever y >= 0; elsewhere both are <= zo (the clamp's lower end)
        C = (1 << 30) + ((1 << (30 + e)) if e > 0 else 0) + (zo << (31 + e))
        got = (x * M + C) >> (31 + e)
        assert (got[y >= 0] == want[y >= 0]).all()
        assert (got[y < 0] <= zo).all() and (want[y < 0] <= zo).all()
        # linear: t = x M + 2^30; y = t >> 31; z = (y + 2^(e-1) + (y >> 63)) >> e
        y2 = (x * M + (1 << 30)) >> 31
        assert (y2 == y).all()
        z = ((y2 + (1 << (e - 1)) + (y2 >> 63)) >> e) if e > 0 else y2
        assert (z + zo == want).all()


def test_independent_anchor_generator_agrees_with_the_product():
    from oracle import nets_quant as nq
    from deepdish_amd import nets
    a, maps = nets.ssd_anchors(300)
    b = nq.ssd_anchors(300)
    assert maps == [19, 10, 5, 3, 2, 1] and a.shape == b.shape == (1917, 4)
    np.testing.assert_array_equal(a, b)
    # spot values of the published generator: first map, first cell = a 0.1 x 0.1 box at (0.5/19, 0.5/19); last anchor = the sqrt(0.95 * 1) square
    np.testing.assert_allclose(b[0], [0.5 / 19, 0.5 / 19, 0.1, 0.1], rtol=1e-6)
    np.testing.assert_allclose(b[-1], [0.5, 0.5, np.sqrt(0.95), np.sqrt(0.95)], rtol=1e-6)
    np.testing.assert_allclose(b[1], [0.5 / 19, 0.5 / 19, 0.2 / np.sqrt(2), 0.2 * np.sqrt(2)], rtol=1e-6)


def test_logistic_tables_agree_and_are_monotone():
    from oracle import nets_quant as nq
    from deepdish_amd import quantize
    for scale, zp in ((0.0419, 211), (0.1, 128), (0.02, 7), (0.5, 255)):
        a, b = nq.logistic_table(scale, zp), quantize.logistic_table(scale, zp)
        np.testing.assert_array_equal(a, b)
        assert (np.diff(a.astype(int)) >= 0).all() and a[zp] == 128


@pytest.mark.parametrize('sym', [False, True])
def test_packing_and_channel_constants_replay_the_oracle(sym):
    """pack_conv's A fragments + cbias + zwc * rowsum, evaluated in numpy exactly as q_conv_k evaluates them (a' = a - 128 operands),
    give the oracle's accumulators -- on a 3x3 stride-2 layer with borders, both epilogue layouts."""
    from oracle import nets_quant as nq
    from deepdish_amd import netsq
    rng = np.random.default_rng(5)
    cin, cout, h = 64, 128, 9
    L = dict(kind='conv', w=rng.integers(0, 256, (3, 3, cin, cout), dtype=np.uint8), w_scale=np.float32(0.01), w_zp=128 if sym else 97,
             bias=rng.integers(-5000, 5000, cout).astype(np.int32), stride=2, act='relu6', in_scale=np.float32(0.0235), in_zp=13,
             out_scale=np.float32(0.0235), out_zp=0)
    x = rng.integers(0, 256, (2, h, h, cin), dtype=np.uint8)
    want = nq.conv_u8(x, L)
    for epi in (netsq.QEPI_Q16, netsq.QEPI_ROWS):
        packed, cb, kcpt = netsq.pack_conv(L, epi)
        n_mfrag, ksteps = packed.shape[:2]
        wf = packed.reshape(n_mfrag, ksteps, 4, 16, 16).astype(np.int64)            # [frag][kstep][fq][row][byte]
        wrow = np.transpose(wf, (0, 3, 1, 2, 4)).reshape(n_mfrag * 16, ksteps * 64)  # row-major: [packed row][k]
        if epi == netsq.QEPI_Q16:
            f, row = np.divmod(np.arange(n_mfrag * 16), 16)
            chan = 64 * (f // 4) + 16 * (row // 4) + 4 * (f % 4) + (row % 4)
        else:
            chan = np.arange(n_mfrag * 16)
        # bordered input, stored as a - 128
        xp = np.full((2, h + 2, h + 2, cin), L['in_zp'], np.int64)
        xp[:, 1:-1, 1:-1] = x
        ap = xp - 128
        ho = (h + 1) // 2
        pad = max((ho - 1) * 2 + 3 - h, 0) // 2
        cols = []
        for dy in range(3):
            for dx in range(3):
                cols.append(ap[:, dy + 1 - pad:dy + 1 - pad + 2 * (ho - 1) + 1:2, dx + 1 - pad:dx + 1 - pad + 2 * (ho - 1) + 1:2, :])
        A = np.concatenate(cols, axis=-1)                                            # [n][ho][wo][9 * cin] in k-step order
        acc = A @ wrow.T + (128 - L['w_zp']) * A.sum(axis=-1, keepdims=True) + cb[chan][None, None, None, :]
        out = np.zeros_like(acc)
        out[..., chan] = acc
        m, shift, lo, hi = nq.layer_fixed_point(L)
        got = np.clip(nq.multiply_by_quantized_multiplier(out, m, shift) + L['out_zp'], lo, hi)
        np.testing.assert_array_equal(got[..., :cout], want)


def test_depthwise_split_covers_nine_bits():
    from deepdish_amd import netsq
    rng = np.random.default_rng(1)
    L = dict(w=rng.integers(0, 256, (3, 3, 32), dtype=np.uint8), w_zp=3, bias=np.zeros(32, np.int32), in_zp=0)
    tab, cb = netsq.pack_dw_mfma(L)
    t8 = tab.view(np.uint8).reshape(2, 64, 2, 4).astype(np.int8).astype(np.int64)
    w9 = (L['w'].astype(np.int64) - 3).reshape(9, 32)
    for ks in range(3):
        for g in range(4):
            t = 4 * ks + g
            got = t8[:, g * 16:(g + 1) * 16, 0, ks] + t8[:, g * 16:(g + 1) * 16, 1, ks]
            np.testing.assert_array_equal(got.reshape(-1), w9[t] if t < 9 else 0)
    L['w_zp'] = 0
    L['w'][0, 0, 0] = 255
    assert netsq.pack_dw_mfma(L) is None                     # 255 is not the sum of two i8: the compiler falls back to the two-op form
