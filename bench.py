#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the detect -> encode -> track hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--streams S]

A "step" is one pass of the hot path over one batch of synthetic input: one 640x480 BGR frame from
each of the S independent video streams this rank owns (SSD-MobileNet-v1 forward + post-process on
the frame, NMS + MARS encoder over ~20 synthetic detections per frame, deep_sort predict/update,
count-line logic).  Frames and detections are generated before the timed region and the frames are
resident in HBM when it starts.  With N > 1 (one rank per GPU under torch.distributed.run) every
rank runs its own streams -- there is no data-path collective; the only exchange is one RCCL
all-reduce of the (pos, neg, int, del) count vector after the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline/cpu_baseline fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
N_OBJ = 20
PEAK_F16_TFLOPS = 2500.0      # dense f16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--streams', type=int, default=None,
                    help='independent video streams per GPU (one frame of each per step)')
    ap.add_argument('--groups', type=int, default=int(os.environ.get('DD_BENCH_GROUPS', '4')),
                    help='worker threads per GPU: the streams are split into this many pipelines, each with its own '
                         'HIP stream, so one group\'s host phases (LSAP, count line) overlap the other\'s kernels')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ingest-host', action='store_true',
                    help='secondary figure (never the headline value): frames start in pinned host memory and cross PCIe '
                         'inside the timed region through the ingest ring (deepdish_amd/ingest.py), one step ahead of the compute')
    ap.add_argument('--background-subtraction', type=float, default=None, metavar='RATIO',
                    help='secondary figure (never the headline value): run with the reference\'s default background subtraction '
                         '(MOG2 on every frame + motion test on the detector boxes, deepdish.py:920-924,957; the reference uses '
                         'RATIO 0.25).  BASELINE.json\'s configurations run with --disable-background-subtraction')
    ap.add_argument('--cpu-frames', type=int, default=300)
    args = ap.parse_args()
    if args.streams is None:        # --ingest-host keeps every step's frames in pinned host memory: a smaller default there
        args.streams = int(os.environ.get('DD_BENCH_STREAMS', '256' if args.ingest_host else '768'))
    return args


def _gen_stream(args):
    seed, n_frames = args
    from deepdish_amd.synth import Scene
    sc = Scene(seed=seed, n_obj=N_OBJ, width=W, height=H, n_frames=n_frames)
    frames = np.stack([sc.frame(f) for f in range(n_frames)])
    per = []
    for f in range(n_frames):
        boxes, scores, who, _ = sc.detections(f)
        per.append(([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(x) for x in scores]))
    return frames, per


def make_inputs(rank, streams, n_frames, sink):
    """Seeded frames and injected detections of every stream (untimed).  sink(s, frames u8 [F, H, W, 3]) places a
    stream's frames (HBM tensor slice or pinned slot) as soon as a worker delivers them, so the host never holds
    more than the few streams in flight; returns the per-stream per-frame detections."""
    import multiprocessing as mp
    dets = [None] * streams
    jobs = [(1000 * rank + s, n_frames) for s in range(streams)]
    workers = max(1, min(16, (os.cpu_count() or 2) // 2, streams))
    if workers > 1:
        with mp.get_context('spawn').Pool(workers) as pool:
            for s, (fr, per) in enumerate(pool.imap(_gen_stream, jobs)):
                sink(s, fr)
                dets[s] = per
    else:
        for s, job in enumerate(jobs):
            fr, per = _gen_stream(job)
            sink(s, fr)
            dets[s] = per
    return dets


def cpu_baseline(n_frames):
    """The oracle's CPU path on the same workload (kind "port"), bounded sample, rank 0 only."""
    import torch
    from PIL import Image
    from deepdish_amd import nets
    from deepdish_amd.synth import Scene
    from oracle import deepsort_np as ds, countline_np as cl, image_np, nets_torch
    threads = min(4, os.cpu_count() or 1)              # reference default --num-threads 4 (deepdish.py:1422)
    torch.set_num_threads(threads)
    sc = Scene(seed=0, n_obj=N_OBJ, width=W, height=H, n_frames=n_frames + 2)
    wd_ssd, wd_mars = nets.synthetic_ssd_weights(1234), nets.synthetic_mars_weights(1234)
    anchors, _ = nets.ssd_anchors(300)
    trk = ds.Tracker(ds.Metric(0.2), max_iou_distance=0.7, max_age=60)
    counter = cl.CountLine(sc.countline())

    def one(f):
        frame = sc.frame(f)
        rgba = np.dstack([frame[..., ::-1], np.full((H, W, 1), 255, np.uint8)])
        img = Image.fromarray(rgba, 'RGBA').convert('RGB').resize((300, 300), Image.LANCZOS)
        raw = nets_torch.ssd_forward(wd_ssd, np.asarray(img)[None])
        nets_torch.ssd_postprocess(raw[0], anchors)
        boxes, scores, _, _ = sc.detections(f)
        keep = ds.non_max_suppression(boxes, 0.6, scores)
        patches = np.stack([image_np.extract_image_patch(frame, boxes[i], (64, 32)) for i in keep])
        feats = nets_torch.mars_forward(wd_mars, patches)
        dets = [ds.Det(boxes[i], 'person', scores[i], feats[j]) for j, i in enumerate(keep)]
        trk.predict(); trk.update(dets); counter.step(trk)

    one(0); one(1)                                       # warm-up, as the reference does (deepdish.py:895-898)
    t0 = time.perf_counter()
    for f in range(2, n_frames + 2):
        one(f)
    dt = time.perf_counter() - t0
    table = [(int(t.track_id), int(t.state), int(t.time_since_update), int(t.hits)) for t in trk.tracks]
    base = dict(value=n_frames / dt, unit='frames/s', cores=threads, kind='port',
                sample='%d frames of the same 640x480 / ~20-detection workload, oracle path (Pillow Lanczos + '
                       'torch-CPU f32 SSD-MobileNet-v1 and MARS + numpy deep_sort), %d torch threads' % (n_frames, threads))
    return base, sc, [int(v) for v in np.asarray(counter.vector()).reshape(-1)], table


def gpu_sample_check(sc, n_frames, oracle_counts, oracle_table, device):
    """The HIP path over the very frames the CPU baseline just processed (one stream): crossing counts and the
    final track table (id, state, time_since_update, hits) must be identical."""
    import torch
    from deepdish_amd.multipipe import MultiStreamPipeline
    mp1 = MultiStreamPipeline(1)
    for f in range(n_frames + 2):
        boxes, scores, _, _ = sc.detections(f)
        inj = mp1.pack_injected([([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(x) for x in scores])])
        mp1.step(torch.from_numpy(sc.frame(f)[None]).to(device), inj)
    ints, _ = mp1.tracker(0).table()
    table = [tuple(int(v) for v in r[:4]) for r in ints]
    counts = [int(v) for v in np.asarray(mp1.counts()[0]).reshape(-1)]
    return dict(frames=n_frames + 2, counts_hip=counts, counts_oracle=oracle_counts,
                identical=bool(counts == oracle_counts and table == oracle_table))


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: keep the real stdout aside and point fd 1 at stderr while
    # libraries (RCCL prints a version banner on stdout) are at work.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist_on = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)    # launched by torch.distributed.run
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    import threading
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.runtime import Context
    n_frames = args.warmup + args.steps
    G = max(1, min(args.groups, args.streams))
    bounds = [round(g * args.streams / G) for g in range(G + 1)]
    ctxs = [Context(local_rank) for _ in range(G)]
    pipes = [MultiStreamPipeline(bounds[g + 1] - bounds[g], context=ctxs[g], background_subtraction_ratio=args.background_subtraction)
             for g in range(G)]
    group_of = [g for g in range(G) for _ in range(bounds[g], bounds[g + 1])]
    ings = dev_frames = None
    if args.ingest_host:
        # one pinned slot per step and group, filled before the timed region (a decoder would write there directly)
        from deepdish_amd.ingest import FrameIngest
        ings = [FrameIngest(bounds[g + 1] - bounds[g], (W, H), slots=n_frames, context=ctxs[g]) for g in range(G)]

        def sink(s, fr):
            g = group_of[s]
            for f in range(n_frames):
                ings[g].host(f)[s - bounds[g]] = fr[f]
    else:
        # frames resident in HBM before the timed region: per group [F][S_g][H][W][3]
        dev_frames = [torch.empty((n_frames, bounds[g + 1] - bounds[g], H, W, 3), dtype=torch.uint8, device=f'cuda:{local_rank}')
                      for g in range(G)]

        def sink(s, fr):
            g = group_of[s]
            dev_frames[g][:, s - bounds[g]] = torch.from_numpy(fr).to(f'cuda:{local_rank}')
    dets = make_inputs(rank, args.streams, n_frames, sink)
    injected = [[pipes[g].pack_injected([dets[s][f] for s in range(bounds[g], bounds[g + 1])]) for f in range(n_frames)]
                for g in range(G)]
    torch.cuda.synchronize()

    def run(g, f0, f1):
        if ings is not None:
            ings[g].submit(f0)
            for f in range(f0, f1):
                if f + 1 < f1:
                    ings[g].submit(f + 1)                         # the next step's upload runs under this step's kernels
                nxt = ings[g].frames(f + 1) if f + 1 < f1 else None
                pipes[g].step(ings[g].frames(f), injected[g][f], nxt)
                ings[g].release(f)
            return
        ahead = os.environ.get('DD_BENCH_NO_LOOKAHEAD') is None
        for f in range(f0, f1):                                   # blocking C call, releases the GIL; the detector of
            pipes[g].step(dev_frames[g][f], injected[g][f],       # frame f+1 is queued behind this step's own detections
                          dev_frames[g][f + 1] if ahead and f + 1 < f1 else None)

    def run_all(f0, f1):
        if G == 1:
            return run(0, f0, f1)
        th = [threading.Thread(target=run, args=(g, f0, f1)) for g in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    run_all(0, args.warmup)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    run_all(args.warmup, n_frames)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    stage_ms = pipes[0].stage_ms()
    local_counts = sum(p.counts().sum(axis=0) for p in pipes)
    tmax = torch.tensor([dt], dtype=torch.float64, device=f'cuda:{local_rank}')
    counts = torch.from_numpy(local_counts).to(f'cuda:{local_rank}')
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # RCCL: the only exchange step of the path
    dt = float(tmax.item())
    total_frames = args.steps * args.streams * world

    if rank == 0:
        out = {
            'metric': 'end-to-end frames/sec (detect+encode+track) at 640x480',
            'value': total_frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': 'SSD-MobileNet-v1 (300x300) + MARS-64x32x3 + deep_sort on synthetic 640x480 BGR '
                                   'frames, ~20 synthetic detections/frame (BASELINE.json configs[1])',
                       'streams_per_gpu': args.streams, 'frames_per_step': args.streams * world,
                       'worker_threads_per_gpu': G,
                       'parallelism': 'independent streams, %d per GPU in %d worker groups' % (args.streams, G),
                       'weights': 'seeded synthetic (seed 1234)'},
            'counts_pos_neg_int_del': [int(v) for v in counts.cpu().numpy().reshape(-1)],
            'stage_ms_per_step': {k: round(v, 4) for k, v in stage_ms.items() if k != 'steps'},
        }
        if args.background_subtraction is not None:
            out['background_subtraction'] = {'ratio': args.background_subtraction,
                                             'boxes_rejected_by_motion_test': int(sum(p.motion_mask(read=False)[1] for p in pipes)),
                                             'note': 'reference default configuration; not the headline (BASELINE configs disable it)'}
            if ings is None:
                os.write(real_stdout, (json.dumps(out) + '\n').encode())
                if dist_on:
                    dist.barrier(); dist.destroy_process_group()
                return
        if ings is not None:
            out['frames_start_in'] = 'pinned host memory (PCIe upload inside the timed region; not the headline configuration)'
            os.write(real_stdout, (json.dumps(out) + '\n').encode())
            if dist_on:
                dist.barrier(); dist.destroy_process_group()
            return
        try:
            from deepdish_amd.profile import dominant_kernel_roofline
            out['roofline'] = dominant_kernel_roofline(pipes, lambda g, f: pipes[g].step(dev_frames[g][f], injected[g][f]), args)
        except Exception as e:                            # never let the extra pass hide the headline number
            out['roofline'] = None
            out['roofline_error'] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'], sc0, ocounts, otable = cpu_baseline(args.cpu_frames)
            try:
                out['parity_sample'] = gpu_sample_check(sc0, args.cpu_frames, ocounts, otable, f'cuda:{local_rank}')
            except Exception as e:
                out['parity_sample'] = dict(error=repr(e))
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
