"""Roofline of mog2_apply_k (csrc/mog2.hip) on the bench workload's frames: S streams of 640x480 synthetic scenes,
a settled model, K timed launches bracketed by HIP events on the launch stream.  Algorithmic bytes per pixel with
n live modes before and n' after the update: 3 + 1 + 20 n read, 20 n' + 1 + 1 written.
Usage: python scripts/bench_mog2.py [streams=96] [warm=24] [timed=12] [still]"""
import json
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from deepdish_amd.background import createBackgroundSubtractorMOG2      # noqa: E402
from deepdish_amd.synth import Scene                                     # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    timed = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    still = len(sys.argv) > 4 and sys.argv[4] == 'still'            # every frame equal to the first: one mode per pixel
    W, H, base = 640, 480, 8
    scenes = [Scene(seed=900 + z, n_obj=20, width=W, height=H, n_frames=warm + timed) for z in range(base)]
    sub = createBackgroundSubtractorMOG2(n_streams=S)
    ts = sub.ctx.torch_stream
    frames = []
    for f in range(warm + timed):                                       # S streams = the base scenes, rolled so no two are equal
        fr = np.stack([np.roll(scenes[z % base].frame(0 if still else f), 7 * (z // base), axis=1) for z in range(S)])
        frames.append(torch.from_numpy(fr).cuda())
    torch.cuda.synchronize()
    for f in range(warm):
        sub.apply_device(frames[f])
    n_before = np.mean([sub.state(z)[3].mean() for z in range(0, S, max(1, S // 4))])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(timed + 1)]
    with torch.cuda.stream(ts):
        ev[0].record(ts)
        for i in range(timed):
            sub.apply_device(frames[warm + i])
            ev[i + 1].record(ts)
    ts.synchronize()
    n_after = np.mean([sub.state(z)[3].mean() for z in range(0, S, max(1, S // 4))])
    us = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(timed)])
    n = 0.5 * (n_before + n_after)
    bytes_px = 3 + 1 + 20 * n + 20 * n + 1 + 1
    total = bytes_px * S * W * H
    fg = float((sub.mask != 0).float().mean().item())
    print(json.dumps({'kernel': 'mog2_apply_k', 'streams': S, 'frame': [W, H], 'avg_launch_us': float(us.mean()), 'min_launch_us': float(us.min()),
                      'live_modes_per_pixel': float(n), 'algorithmic_bytes_per_pixel': float(bytes_px),
                      'algorithmic_bytes_per_launch': float(total), 'achieved_GBps': float(total / us.mean() * 1e-3),
                      'peak_GBps': 8000.0, 'frac': float(total / us.mean() * 1e-3 / 8000.0), 'foreground_fraction': fg,
                      'frames_per_s_kernel_alone': float(S / us.mean() * 1e6)}))


if __name__ == '__main__':
    main()
