"""Measurement helpers for bench.py: per-kernel device time from HIP events on the launch stream
and the roofline line of the dominant kernel (algorithmic FLOPs / bytes come from the op program)."""
import ctypes
import numpy as np

from ._lib import lib, check
from .runtime import ptr

PEAK_F16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: Peak BF16/FP16 MFMA ~2.5 PF dense
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak BW 8.0 TB/s spec
PEAK_I8_TOPS = 5000.0          # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate per clock


def last_batch(net):
    n = ctypes.c_int()
    check(lib().dd_net_last_batch(net._h, ctypes.byref(n)), 'dd_net_last_batch')
    return n.value


def net_op_times(net):
    """Per-op milliseconds of the last forward: the HIP-event bracket around each op on the launch stream,
    minus the cost of an empty bracket (done inside dd_net_profile_read)."""
    n = len(net.program.ops)
    ms = np.zeros(n, dtype=np.float32)
    cnt = ctypes.c_int()
    check(lib().dd_net_profile_read(net._h, ptr(ms), n, ctypes.byref(cnt)), 'dd_net_profile_read')
    return ms


# dd_net_op_launches codes (include/deepdish_hip.h)
OPK_FOLDED = 1
OPK_NAMES = {2: 'conv3x3_pool_rows_k', 3: 'conv3x3_pool_rows_k<STEM>', 4: 'res_unit_rows_k', 5: 'ssd_front_k', 6: 'conv3x3_c64_rows_k', 7: 'conv3x3_s2_rows_k', 8: 'conv_ws_k', 9: 'conv_ws_dw_k', 10: 'dwpw_rows_k', 11: 'conv_glds_k<ssd_head_decode>', 12: 'res_pair_rows_k', 13: 'conv_glds_k<yolo_head_decode>', 14: 'mars_ws128_k', 16: 'mars_pair64_k', 17: 'q_dwm_k', 18: 'conv3x3_c64_rows_k<strips>', 19: 'q_front_k', 20: 'q_mid_k'}
OPK_FOLDED_PREV = 15      # the op ran inside the PREVIOUS op's launch (a 1x1 stride-2 projection beside its block's 3x3 stride-2 layer)


def net_op_launches(net):
    """Which launch ran each op of the last forward (fused launches are attributed to the kernel that ran)."""
    n = len(net.program.ops)
    codes = np.zeros(n, dtype=np.int32)
    cnt = ctypes.c_int(0)
    check(lib().dd_net_op_launches(net._h, ptr(codes), n, ctypes.byref(cnt)), 'dd_net_op_launches')
    return codes


def launches_of(net, ms, batch):
    """[(kernel, ms, flops, bytes)] per LAUNCH of the last forward: an op folded into the next op's launch adds its FLOPs
    and the bytes it reads to that launch, and the tensor between the two (never written) is not counted."""
    out, pend = [], None
    for t, info, code in zip(ms, net.program.info, net_op_launches(net)):
        fl, by = info['flops'] * batch, info['bytes'] * batch + info.get('wbytes', 0)
        if code == OPK_FOLDED:
            if pend is None:                                # a chain of folded ops: the first one's source is what the launch reads
                pend = [float(t), fl, info.get('src_bytes', 0) * batch + info.get('wbytes', 0)]
            else:
                pend[0] += float(t); pend[1] += fl; pend[2] += info.get('wbytes', 0)
            continue
        if code == OPK_FOLDED_PREV and out:
            k0, t0, f0, b0 = out[-1]
            out[-1] = (k0, t0 + float(t), f0 + fl, b0 + by)
            continue
        if pend is not None:
            t, fl, by = float(t) + pend[0], fl + pend[1], by - info.get('src_bytes', 0) * batch + pend[2]
            pend = None
        out.append((OPK_NAMES.get(int(code), info['kernel']), float(t), fl, by))
    return out


def profile_nets(run_once, nets_with_batch, reps=20):
    """run_once() runs one bench step; nets_with_batch: [(name, Net, images per forward)].
    Returns {kernel: dict(ms, flops, bytes, launches)} averaged per step.

    Every op is bracketed by two HIP events on the launch stream.  An event record costs a few
    microseconds of stream time itself, so each op's interval is corrected by the interval measured
    for an empty bracket (two back-to-back records) on the same stream."""
    for _, net, _ in nets_with_batch:
        check(lib().dd_net_profile(net._h, 1), 'dd_net_profile')
    acc = {}
    try:
        for _ in range(reps):
            run_once()
            for name, net, batch in nets_with_batch:
                ms = net_op_times(net)
                b = batch() if callable(batch) else batch
                for kname, t, fl, by in launches_of(net, ms, b):
                    k = acc.setdefault(kname, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                    k['ms'] += t; k['flops'] += fl; k['bytes'] += by; k['launches'] += 1
    finally:
        for _, net, _ in nets_with_batch:
            check(lib().dd_net_profile(net._h, 0), 'dd_net_profile')
    for k in acc.values():
        for f in ('ms', 'flops', 'bytes'):
            k[f] /= reps
        k['launches'] /= reps
    return acc


def pmc_traffic(kernel, streams):
    """(HBM bytes per launch of `kernel`, where the figure comes from) from the committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE and
    --pmc WRITE_SIZE passes of this same bench; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    None when no summary exists for this stream count -- bench.py cannot run the profiler on itself."""
    import json, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
    path = next((p for p in (os.path.join(root, 'r%02d_pmc_traffic_s%d.json' % (r, streams)) for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(p)), None)
    if path is None:                       # the newest round's summary for this stream count, if one was collected
        return None, None
    keys = [part.replace(',', ', ') for part in kernel.split('+')]
    tot_b = tot_n = 0.0
    for k, v in json.load(open(path)).items():          # a kernel family (all tile shapes of conv_glds_k) is summed
        if any(key in k for key in keys):
            tot_b += (v['hbm_read_bytes_corrected'] + v['hbm_write_bytes']) * v['launches']
            tot_n += v['launches']
    src = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of `bench.py --groups 1 --streams %d`, committed; not measured in this run)' % (os.path.basename(path), streams)
    return (tot_b / tot_n, src) if tot_n else (None, None)


def dominant_kernel_roofline(pipes, step_group, args, reps=20):
    """Extra instrumented pass after the timed region (kept out of `value`).

    Per-op HIP events on the launch stream of ONE worker group while the other groups are idle: an
    event bracket is only a kernel duration when nothing else shares the GPU (with several groups
    running, a bracket on one stream also spans the other streams' kernels; measured: 50 us vs the
    35 us rocprofv3 reports for the same launches).  The matching profiler summary is therefore the one
    of `bench.py --groups 1` (profiles/r01_bench_groups1_kernel_stats.csv); in the default multi-group
    run the same kernels take longer each because the groups' kernels overlap on the chip
    (profiles/r01_bench_default_kernel_stats.csv).  step_group(g, f) advances group g by one step."""
    f0 = args.warmup
    p = pipes[0]
    nets = [('mars', p.enc, lambda: last_batch(p.enc))]
    if p.det is not None:
        nets.append(('ssd', p.det, p.S))
    state = dict(f=f0)

    def run_once():
        step_group(0, state['f'])
        state['f'] = f0 + (state['f'] + 1 - f0) % max(1, args.steps)

    acc = profile_nets(run_once, nets, reps)
    per_kernel = {n: round(v['ms'], 5) for n, v in sorted(acc.items(), key=lambda kv: -kv[1]['ms'])}
    # the dense-convolution GEMMs are one family: conv_ws_k runs the large pointwise layers from 160 images per launch,
    # conv_glds_k the rest and all of them below that (same layers, same arithmetic, batch-dependent choice)
    fam = [n for n in ('conv_glds_k', 'conv_ws_k', 'conv_ws_dw_k', 'conv_glds_k<ssd_head_decode>') if n in acc]     # conv_ws_dw_k: conv_ws_k with the next depthwise layer in its epilogue
    if len(fam) > 1:
        parts = [acc.pop(n) for n in fam]
        acc['+'.join(fam)] = {f: sum(p[f] for p in parts) for f in ('ms', 'flops', 'bytes', 'launches')}
    name, k = max(acc.items(), key=lambda kv: kv[1]['ms'])
    sec = k['ms'] * 1e-3
    avg_us = 1e3 * k['ms'] / max(k['launches'], 1e-9)
    if name.startswith('q_'):
        # uint8 detector: the fused MobileNet blocks move one read and one write of every block tensor and run both contractions on
        # i8 MFMA; priced against HBM (the larger fraction), the matrix fraction next to it
        achieved = k['bytes'] / sec / 1e9
        out = dict(bound='hbm', achieved=achieved, peak=PEAK_HBM_GBS, unit='GB/s', frac=achieved / PEAK_HBM_GBS,
                   mfma_i8=dict(achieved=k['flops'] / sec / 1e12, peak=PEAK_I8_TOPS, unit='TOP/s', frac=k['flops'] / sec / 1e12 / PEAK_I8_TOPS))
    elif name.startswith('conv') or name.startswith('res_unit'):
        achieved = k['flops'] / sec / 1e12
        out = dict(bound='mfma', achieved=achieved, peak=PEAK_F16_TFLOPS, unit='TFLOP/s', frac=achieved / PEAK_F16_TFLOPS)
    else:
        achieved = k['bytes'] / sec / 1e9
        out = dict(bound='hbm', achieved=achieved, peak=PEAK_HBM_GBS, unit='GB/s', frac=achieved / PEAK_HBM_GBS)
    traffic, traffic_source = pmc_traffic(name, p.S)
    out.update(kernel=name, traffic=traffic, traffic_source=traffic_source, launches_per_step=k['launches'], avg_launch_us=avg_us,
               frames_per_launch=p.S,
               algorithmic_per_launch=dict(flops=k['flops'] / max(k['launches'], 1e-9), bytes=k['bytes'] / max(k['launches'], 1e-9)),
               measured='HIP events on the launch stream, one worker group (%d streams) alone on the GPU' % p.S,
               per_kernel_ms_per_step=per_kernel)
    return out
