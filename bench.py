#!/usr/bin/env python3
"""bench.py -- end-to-end frames/s of the detect -> encode -> track hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,5}] [--streams S]

A "step" is one pass of the hot path over one batch of synthetic input: one BGR frame from each of the S
independent video streams this rank owns (detector forward + post-process on the frame, NMS + MARS encoder over
~20 synthetic detections per frame, deep_sort predict/update, count-line logic).  Frames and detections are
generated before the timed region and the frames are resident in HBM when it starts.

--gpus N > 1: when this process was not started by a launcher (no RANK in the environment) it starts N ranks of
itself with `python -m torch.distributed.run` (one per GPU, rendezvous on 127.0.0.1) BEFORE anything touches the
GPU and relays rank 0's JSON line; under a launcher it checks WORLD_SIZE == N.  Every rank runs its own streams --
there is no data-path collective; the only exchange is one RCCL all-reduce of the (pos, neg, int, del) count
vector after the timed region (deepdish_amd/multistream.py).

--config picks the BASELINE.json configuration (1-based): 2 = SSD-MobileNet-v1 + MARS + deep_sort on 640x480 frames
(the headline; default), 3 = YOLOv5s-f16 detector on 640x640 frames, 5 = 1280x720 streams, one per GPU.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_OBJ = 20
DEFAULT_DETECTOR_DTYPE = 'i8'       # the reference's own detector arithmetic (its ssdmobilenetv1.tflite is uint8-quantised); --detector-dtype f16 = the float model
CONFIGS = {
    # 3 072 streams in 4 worker groups = 768 frames and ~15 000 crops per launch: with the crop-resident encoder kernels of round 5 bigger launches pay
    # (same box, 20 steps: 4 x 384 streams 102.6 / 108.0 k frames/s, 4 x 768 114.7 / 116.2 k, 4 x 1 024 119.4 k); at 768 per launch every stream still
    # advances at 37 frames/s, above camera rate
    # round 6 (the SSD front end as one row pipeline, zero lo parts skipped): one group alone now reaches 95 % of the multi-group figure, and two groups of
    # 1 536 beat four of 768 (same box, alternating: 135.2 / 135.1 k against 131.5 / 129.0 k; six of 512: 125 k; one of 3 072: 125 k)
    2: dict(W=640, H=480, model='synthetic-ssd_mobilenet_v1', streams=3072, groups=2,
            workload='SSD-MobileNet-v1 (300x300) + MARS-64x32x3 + deep_sort on synthetic 640x480 BGR frames, '
                     '~20 synthetic detections/frame (BASELINE.json configs[1])'),
    # (round 5, same box, 20 steps: 2 x 256 streams 21.9 k frames/s, 3 x 256 22.2 k, 2 x 512 and 4 x 256 22.5 k)
    3: dict(W=640, H=640, model='synthetic-yolov5s-fp16', streams=1024, groups=4,
            workload='YOLOv5s-f16 (640x640) + HIP NMS / IoU + MARS-64x32x3 + deep_sort on synthetic 640x640 BGR frames, '
                     '~20 synthetic detections/frame (BASELINE.json configs[2])'),
    5: dict(W=1280, H=720, model='synthetic-ssd_mobilenet_v1', streams=1, groups=1,
            workload='SSD-MobileNet-v1 + MARS-64x32x3 + deep_sort on synthetic 1280x720 BGR streams, one stream per GPU '
                     '(BASELINE.json configs[4]); latency-bound by construction: one frame per step'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)          # (steps + warmup) x streams x 0.92 MB of frames are resident in HBM: 141 GB at the defaults
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', type=int, default=2, choices=sorted(CONFIGS),
                    help='BASELINE.json configuration (1-based): 2 headline, 3 YOLOv5s detector, 5 one 1280x720 stream per GPU')
    ap.add_argument('--streams', type=int, default=None,
                    help='independent video streams per GPU (one frame of each per step)')
    ap.add_argument('--groups', type=int, default=None,
                    help='worker threads per GPU: the streams are split into this many pipelines, each with its own '
                         'HIP stream, so one group\'s host phases (LSAP, count line) overlap the other\'s kernels')
    ap.add_argument('--detector-dtype', choices=('f16', 'i8'), default=DEFAULT_DETECTOR_DTYPE,
                    help='SSD-MobileNet arithmetic (configs 2 and 5): i8 = the uint8-quantised model the reference ships '
                         '(tools/ssd_mobilenet.py:102), integer kernels bit-exact against the TFLite restatement; f16 = the float model')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ingest-host', action='store_true',
                    help='secondary figure (never the headline value): frames start in pinned host memory and cross PCIe '
                         'inside the timed region through the ingest ring (deepdish_amd/ingest.py), one step ahead of the compute')
    ap.add_argument('--background-subtraction', type=float, default=None, metavar='RATIO',
                    help='secondary figure (never the headline value): run with the reference\'s default background subtraction '
                         '(MOG2 on every frame + motion test on the detector boxes, deepdish.py:920-924,957; the reference uses '
                         'RATIO 0.25).  BASELINE.json\'s configurations run with --disable-background-subtraction')
    ap.add_argument('--cpu-frames', type=int, default=40, help='frames per seed of the CPU baseline sample (3 seeds)')
    ap.add_argument('--rehearse-cpu', action='store_true',
                    help='launcher / rendezvous / count-reduction rehearsal over gloo on the CPU: no frames are processed and no '
                         'throughput is reported (what tests/test_bench_launcher.py runs where there is no GPU)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.streams is None:        # --ingest-host keeps every step's frames in pinned host memory (0.92 MB per frame): a smaller default there
        args.streams = int(os.environ.get('DD_BENCH_STREAMS', '1536' if args.ingest_host and args.config == 2 else cfg['streams']))
    if args.groups is None:
        args.groups = int(os.environ.get('DD_BENCH_GROUPS', cfg['groups']))
    return args


def launch_ranks(args):
    """--gpus N without a launcher: start N fresh ranks (never re-exec a process that has touched the GPU) and relay
    rank 0's JSON line.  Returns the exit status for this parent."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault('OMP_NUM_THREADS', '1')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        ln = ln.strip()
        if ln.startswith('{') and '"metric"' in ln:
            line = ln                                          # the ranks' other chatter (none expected) is dropped
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def _gen_stream(job):
    seed, n_frames, W, H = job
    from deepdish_amd.synth import Scene
    sc = Scene(seed=seed, n_obj=N_OBJ, width=W, height=H, n_frames=n_frames)
    frames = np.stack([sc.frame(f) for f in range(n_frames)])
    per = []
    for f in range(n_frames):
        boxes, scores, who, _ = sc.detections(f)
        per.append(([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(x) for x in scores]))
    return frames, per


def gen_workers(world, streams):
    """Generator processes of this rank: the ranks of one node share its cores, each rank takes its share of half of them."""
    workers = max(1, min(16, (os.cpu_count() or 2) // (2 * max(1, world)), streams))
    if os.environ.get('DD_BENCH_GEN_WORKERS'):          # 1 = generate in-process: under `rocprofv3 --pmc` the profiler has
        workers = int(os.environ['DD_BENCH_GEN_WORKERS'])   # initialised the GPU before main(), and worker processes may not be exec'ed
    return workers


def host_threads(world, groups):
    """Threads of the C++ pipeline's host pool (csrc/hostpool.h: detector-adaptor filter, box hygiene, matching cascade + LSAP,
    count line, one stream per task) for one rank: its share of the node's cores minus its worker-group threads and the main thread."""
    return max(1, min(12, (os.cpu_count() or 2) // max(1, world) - groups - 1))


def start_gen_pool(world, streams):
    """The generator pool, started BEFORE this process touches the GPU (torch.cuda.set_device, RCCL init, the first HIP call):
    a process that has initialised the GPU must not start children on this pool, and N ranks doing so together is where it hurts."""
    import multiprocessing as mp
    workers = gen_workers(world, streams)
    return mp.get_context('spawn').Pool(workers) if workers > 1 else None


def make_inputs(pool, rank, streams, n_frames, W, H, sink):
    """Seeded frames and injected detections of every stream (untimed).  sink(s, frames u8 [F, H, W, 3]) places a
    stream's frames (HBM tensor slice or pinned slot) as soon as a worker delivers them, so the host never holds
    more than the few streams in flight; returns the per-stream per-frame detections."""
    dets = [None] * streams
    seed_rank = int(os.environ.get('DD_BENCH_SEED_RANK', rank))     # tests: a single-GPU run of the streams rank k would own
    jobs = [(1000 * seed_rank + s, n_frames, W, H) for s in range(streams)]
    it = pool.imap(_gen_stream, jobs) if pool is not None else map(_gen_stream, jobs)
    for s, (fr, per) in enumerate(it):
        sink(s, fr)
        dets[s] = per
    if pool is not None:
        pool.close()
        pool.join()
    return dets


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(n_frames, W, H, config=2):
    """The oracle's CPU path on the same workload (kind "port"), bounded sample, rank 0 only -- SURVEY.md 8(d):
    three seeds (median), every stage timed with 1 thread and with min(4, cores) threads (torch intra-op threads for
    the networks, BLAS / OpenMP pool for the numpy tracker; the reference's default is --num-threads 4,
    deepdish.py:1422) and the FASTER setting taken per stage, so the GPU / CPU ratio is not flattered.
    Stages: detector = Pillow Lanczos + f32 SSD-MobileNet forward + post-process; encoder = crops + f32 MARS;
    tracker = deep_sort NMS + predict / update + count line."""
    import torch
    from PIL import Image
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:                                  # not a declared dependency: the numpy tracker then runs at the pool's default
        import contextlib

        def threadpool_limits(limits=None):
            return contextlib.nullcontext()
    from deepdish_amd import nets
    from deepdish_amd.synth import Scene
    from oracle import deepsort_np as ds, countline_np as cl, image_np, nets_torch
    ncores = os.cpu_count() or 1
    settings = sorted({1, min(4, ncores)})
    from oracle import detectors_np
    yolo = config == 3
    wd_mars = nets.synthetic_mars_weights(1234)
    wd_det = nets.synthetic_yolov5s_weights(1234) if yolo else nets.synthetic_ssd_weights(1234)
    from oracle import nets_quant
    anchors = nets_quant.ssd_anchors(300)                # the oracle's own generator (tests/test_quant_host.py holds the product's to it)
    yolo_labels = {i: l.strip() for i, l in enumerate(open(os.path.join(ROOT, 'deepdish_amd', 'assets', 'coco_classes.txt')))}
    per_seed, stage_pick = [], {}
    keep_for_parity = None
    for seed in (0, 1, 2):
        sc = Scene(seed=seed, n_obj=N_OBJ, width=W, height=H, n_frames=n_frames + 2)
        frames = [sc.frame(f) for f in range(n_frames + 2)]
        dets_in = [sc.detections(f) for f in range(n_frames + 2)]
        keeps = [ds.non_max_suppression(d[0], 0.6, d[1]) for d in dets_in]
        feats = [None] * (n_frames + 2)
        best = {}

        def detector(f):
            rgba = np.dstack([frames[f][..., ::-1], np.full((H, W, 1), 255, np.uint8)])
            if yolo:
                img = Image.fromarray(rgba, 'RGBA').convert('RGB').resize((640, 640), Image.LANCZOS)
                raw = nets_torch.yolov5s_forward(wd_det, np.asarray(img)[None])
                detectors_np.yolov5_detect_tail(raw, yolo_labels, ['person'], 0.25, (W, H))
                return
            img = Image.fromarray(rgba, 'RGBA').convert('RGB').resize((300, 300), Image.LANCZOS)
            raw = nets_torch.ssd_forward(wd_det, np.asarray(img)[None])
            nets_torch.ssd_postprocess(raw[0], anchors)

        def encoder(f):
            boxes = dets_in[f][0]
            patches = np.stack([image_np.extract_image_patch(frames[f], boxes[i], (64, 32)) for i in keeps[f]])
            feats[f] = nets_torch.mars_forward(wd_mars, patches)

        for name, fn in (('detector', detector), ('encoder', encoder)):
            for t in settings:
                torch.set_num_threads(t)
                fn(0); fn(1)                                       # warm-up, as the reference does (deepdish.py:895-898)
                t0 = time.perf_counter()
                for f in range(2, n_frames + 2):
                    fn(f)
                best.setdefault(name, {})[t] = (time.perf_counter() - t0) / n_frames
        table = counts = None
        for t in settings:
            with threadpool_limits(limits=t):
                trk = ds.Tracker(ds.Metric(0.2), max_iou_distance=0.7, max_age=60)
                counter = cl.CountLine(sc.countline())
                dt = 0.0
                for f in range(n_frames + 2):
                    boxes, scores = dets_in[f][0], dets_in[f][1]
                    t0 = time.perf_counter()
                    keep = ds.non_max_suppression(boxes, 0.6, scores)
                    dets = [ds.Det(boxes[i], 'person', scores[i], feats[f][j]) for j, i in enumerate(keep)]
                    trk.predict(); trk.update(dets); counter.step(trk)
                    if f >= 2:
                        dt += time.perf_counter() - t0
                best.setdefault('tracker', {})[t] = dt / n_frames
                table = [(int(x.track_id), int(x.state), int(x.time_since_update), int(x.hits)) for x in trk.tracks]
                counts = [int(v) for v in np.asarray(counter.vector()).reshape(-1)]
        if seed == 0:
            keep_for_parity = (sc, counts, table)
        sec = 0.0
        for name, by_t in best.items():
            t_best = min(by_t, key=by_t.get)
            stage_pick.setdefault(name, []).append((t_best, by_t))
            sec += by_t[t_best]
        per_seed.append(1.0 / sec)
    value = float(np.median(per_seed))
    stages = {}
    for name, picks in stage_pick.items():
        stages[name] = {'ms_per_frame_by_threads': {str(t): round(1e3 * float(np.median([p[1][t] for p in picks])), 3) for t in settings},
                        'threads_used': int(np.median([p[0] for p in picks]))}
    used = max(s['threads_used'] for s in stages.values())
    base = dict(value=value, unit='frames/s', cores=used, kind='port',
                sample='3 seeds x %d frames of the same %dx%d / ~20-detection workload (median), oracle path: Pillow Lanczos + '
                       'torch-CPU f32 %s and MARS + numpy deep_sort; per stage the faster of %s threads.  NOTE: a float32 CPU detector '
                       'beside a uint8 GPU detector -- the integer restatement of the reference\'s uint8 TFLite arithmetic (oracle/nets_quant.py) '
                       'is a checker (~2 s per frame in numpy), not a CPU implementation, and tflite_runtime is absent; the reference\'s own '
                       'uint8 CPU path would be faster than this f32 forward.  A reported baseline, not the target.'
                       % (n_frames, W, H, 'YOLOv5s' if yolo else 'SSD-MobileNet-v1', ' / '.join(map(str, settings))),
                per_seed=[round(v, 2) for v in per_seed], stages=stages,
                host=dict(cpu_model=_cpu_model(), logical_cores=ncores))
    return base, keep_for_parity


def gpu_sample_check(sc, n_frames, oracle_counts, oracle_table, device, W, H, model, streams):
    """The HIP path over the very frames the CPU baseline just processed (seed 0), at the LAUNCH SHAPE of the timed region: a
    `streams`-stream pipeline (one worker group of the bench) in which every slot replays those frames, so the detector,
    the encoder and the tracker kernels run at the batch sizes the headline was measured at.  Every stream must end with
    the oracle's crossing counts and final track table (id, state, time_since_update, hits)."""
    import torch
    from deepdish_amd.multipipe import MultiStreamPipeline
    mp1 = MultiStreamPipeline(streams, model=model, input_size=(W, H))
    for f in range(n_frames + 2):
        boxes, scores, _, _ = sc.detections(f)
        one = ([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(x) for x in scores])
        inj = mp1.pack_injected([one] * streams)
        frames = torch.from_numpy(sc.frame(f)[None]).to(device).expand(streams, H, W, 3).contiguous()
        mp1.step(frames, inj)
    all_counts = mp1.counts()
    bad = []
    for z in range(streams):
        ints, _ = mp1.tracker(z).table()
        table = [tuple(int(v) for v in r[:4]) for r in ints]
        counts = [int(v) for v in np.asarray(all_counts[z]).reshape(-1)]
        if counts != oracle_counts or table != oracle_table:
            bad.append(z)
    counts0 = [int(v) for v in np.asarray(all_counts[0]).reshape(-1)]
    out = dict(frames=n_frames + 2, streams=streams, counts_hip=counts0, counts_oracle=oracle_counts,
               identical=not bad, streams_that_differ=bad[:8])
    if 'uint8' in model or 'quant' in model:
        try:
            del mp1
            torch.cuda.empty_cache()
            out['detector'] = detector_sample_check(model, sc, device, W, H, streams)
        except Exception as e:
            out['detector'] = dict(error=repr(e))
    return out


def detector_sample_check(model, sc, device, W, H, streams):
    """One step of a pipeline of the same launch shape WITHOUT injected detections, every label wanted (random weights rarely say
    'person'): the uint8 detector's own host block (what detect_image returns,
    tools/ssd_mobilenet.py:198-213) against the oracle chain for that frame -- Pillow Lanczos -> oracle/nets_quant.py (TFLite's integer
    arithmetic) -> the post-process op (oracle/nets_torch.py) -> predict()'s tail and detect_image()'s filter (oracle/detectors_np.py).
    Every stream slot holds the same frame, so every slot must hold the same rows."""
    import torch
    from PIL import Image
    from oracle import nets_quant, nets_torch, detectors_np
    from deepdish_amd.tools.weights_io import load_ssd_model, ssd_post_options
    from deepdish_amd.pipeline import DEFAULT_LABELS
    from deepdish_amd.multipipe import MultiStreamPipeline
    labels = {i: l.strip() for i, l in enumerate(open(DEFAULT_LABELS))}
    wanted = sorted({l for l in labels.values() if l and l != '???'})
    mp1 = MultiStreamPipeline(streams, model=model, input_size=(W, H), wanted_labels=wanted)
    f = 1
    frame = sc.frame(f)
    mp1.step(torch.from_numpy(frame[None]).to(device).expand(streams, H, W, 3).contiguous())
    kind, qm = load_ssd_model(model)
    post = ssd_post_options(qm)
    rgba = np.dstack([frame[..., ::-1], np.full((H, W, 1), 255, np.uint8)])
    resized = np.asarray(Image.fromarray(rgba, 'RGBA').convert('RGB').resize((300, 300), Image.LANCZOS))
    box_q, cls_q, _ = nets_quant.ssd_quant_forward(qm, resized[None])
    b, s_, c, _ = nets_quant.ssd_quant_decode(qm, box_q[0], cls_q[0], nets_quant.ssd_anchors(300), post['nms_score_threshold'])
    op = nets_torch.ssd_postprocess_decoded(b, s_, c, post['max_detections'], post['nms_score_threshold'], post['nms_iou_threshold'])
    boxes, names, scores = detectors_np.ssd_predict_tail(list(op), labels, original_image_size=(W, H))
    wb, wl, ws = detectors_np.ssd_detect_filter(boxes, names, scores, mp1.wanted, 0.5)
    want = sorted((l, float(s), tuple(float(v) for v in bb)) for bb, l, s in zip(wb, wl, ws))
    bad, worst = [], 0.0
    for z in range(streams):
        gb, gl, gs = mp1.detections(z)
        got = sorted((l, float(s), tuple(float(v) for v in bb)) for bb, l, s in zip(gb, gl, gs))
        ok = len(got) == len(want) and all(g[0] == w[0] and g[1] == w[1] for g, w in zip(got, want))
        if ok and want:
            d = max(abs(a - b) for g, w in zip(got, want) for a, b in zip(g[2], w[2]))
            worst = max(worst, d)
            ok = d <= 2e-6 * max(W, H) * 2                # decoded corners: expf vs numpy's exp, scaled to pixels (tlwh: two corners)
        if not ok:
            bad.append(z)
    return dict(frame=f, streams=streams, rows_oracle=len(want), identical=not bad, streams_that_differ=bad[:8], max_box_abs_diff_px=worst,
                chain='Pillow Lanczos -> nets_quant.ssd_quant_forward -> ssd_quant_decode -> nets_torch.ssd_postprocess_decoded -> detectors_np tail')


# ------------------------------------------------------------------------------------------------ rehearsal
def rehearse(args, rank, world, real_stdout):
    """No GPU: the launcher, the rendezvous, the barrier / max-over-ranks timing protocol and the count reduction,
    over gloo.  Nothing is measured and no frame is processed."""
    import torch
    import torch.distributed as dist
    from deepdish_amd.multistream import reduce_counts
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    local = np.array([[rank + 1, 10 * (rank + 1), 11 * (rank + 1), 0]], dtype=np.int64)      # stands for this rank's counts
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    if dist.is_initialized():
        dist.barrier()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    counts = reduce_counts(local)
    # what every rank would start on the host for the real run (its share of the node's cores)
    G = max(1, min(args.groups, args.streams))
    mine = dict(rank=rank, generator_processes=gen_workers(world, args.streams), worker_groups=G, host_pool_threads=host_threads(world, G))
    budgets = [mine]
    if dist.is_initialized():
        budgets = [None] * world
        dist.all_gather_object(budgets, mine)
    if rank == 0:
        out = {'metric': 'end-to-end frames/sec (detect+encode+track) at 640x480', 'value': None, 'unit': 'frames/s',
               'n_gpus': world, 'steps': 0, 'warmup': 0, 'rehearsal': 'launcher / rendezvous / count reduction only (gloo, CPU): '
               'no frames processed, nothing measured', 'counts_pos_neg_int_del': [int(v) for v in counts.reshape(-1)],
               'host_cores': os.cpu_count(), 'per_rank_host': budgets}
        if dist.is_initialized():
            out['rccl_ranks'] = dist.get_world_size()       # (the rehearsal's communicator is gloo: what the field holds in the real run is RCCL's)
            out['collective_backend'] = 'gloo'
        if os.environ.get('DD_BENCH_REPORT_ENV'):           # tests: what this rank's environment holds when the collectives come up
            out['env_of_rank0'] = {k: os.environ.get(k) for k in ('HSA_ENABLE_IPC_MODE_LEGACY', 'MASTER_ADDR', 'MASTER_PORT')}
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))                         # before torch is imported or the GPU is touched
    if args.ingest_host:
        # two worker groups = six HIP streams (main, detector, copy each); ROCm maps streams onto 4 hardware queues by default and a copy stream
        # that shares one with another group's kernels waits behind them: 47.9 k frames/s with 4 queues, 54.4-54.7 k with 6 / 8 / 16 (same box)
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    if 'RANK' in os.environ:
        # a rank started by a launcher other than launch_ranks (the driver's `python -m torch.distributed.run ...`): RCCL needs dmabuf IPC
        # on this host driver, and the variable must be in the environment before torch / the HIP runtime are loaded
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n' % (args.gpus, world))
        sys.exit(2)
    # stdout carries exactly ONE JSON line: keep the real stdout aside and point fd 1 at stderr while
    # libraries (RCCL prints a version banner on stdout) are at work.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.rehearse_cpu:
        return rehearse(args, rank, world, real_stdout)
    cfg = dict(CONFIGS[args.config])
    i8 = args.detector_dtype == 'i8' and 'mobilenet' in cfg['model']
    if i8:                                                  # the uint8-quantised detector (deepdish_amd/quantize.py, csrc/netsq.hip)
        cfg['model'] += '-uint8'
        cfg['workload'] = cfg['workload'].replace('SSD-MobileNet-v1', 'SSD-MobileNet-v1 uint8-quantised')
    W, H = cfg['W'], cfg['H']
    gen_pool = start_gen_pool(world, args.streams)          # before anything touches the GPU
    G = max(1, min(args.groups, args.streams))
    # host threads of the C++ pipeline's per-stream phases (csrc/hostpool.h): this rank's share of the cores minus its worker groups
    os.environ.setdefault('DD_HOST_THREADS', str(host_threads(world, G)))
    import torch
    dist_on = world > 1 or ('RANK' in os.environ and 'MASTER_PORT' in os.environ)    # launched by torch.distributed.run
    # Rehearsal on a one-GPU box (tests/test_gpu_pipeline.py): DD_BENCH_ONE_DEVICE=1 puts every rank on device 0 and
    # DD_BENCH_BACKEND=gloo carries the count reduction (RCCL refuses two ranks on one device); everything else --
    # launcher, per-rank worker threads, barriers, max-over-ranks timing -- is the N > 1 path as the driver runs it.
    backend = os.environ.get('DD_BENCH_BACKEND', 'nccl')
    if os.environ.get('DD_BENCH_ONE_DEVICE'):
        local_rank = 0
    red_dev = f'cuda:{local_rank}' if backend == 'nccl' else 'cpu'
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
        comm_ranks = dist.get_world_size()                   # what the communicator holds after init (reported as rccl_ranks)

    import threading
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.multistream import reduce_counts
    from deepdish_amd.runtime import Context
    n_frames = args.warmup + args.steps
    # every step's frames are resident before the timed region (HBM, or pinned host memory + the ring's device slots with --ingest-host):
    # say so before allocating rather than dying inside an allocation (engines: ~14 MB per stream at these launch sizes)
    need = n_frames * args.streams * H * W * 3
    free_b, total_b = torch.cuda.mem_get_info(local_rank)
    engines = args.streams * 14 * (1 << 20)
    if need + engines > free_b:
        sys.stderr.write('bench.py: %d steps x %d streams of %dx%d frames = %.1f GB resident (+ ~%.1f GB of engines), %.1f GB free on the device: '
                         'lower --steps / --warmup / --streams\n' % (n_frames, args.streams, W, H, need / 1e9, engines / 1e9, free_b / 1e9))
        sys.exit(2)
    bounds = [round(g * args.streams / G) for g in range(G + 1)]
    ctxs = [Context(local_rank) for _ in range(G)]
    pipes = [MultiStreamPipeline(bounds[g + 1] - bounds[g], model=cfg['model'], input_size=(W, H), context=ctxs[g],
                                 background_subtraction_ratio=args.background_subtraction)
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
    dets = make_inputs(gen_pool, rank, args.streams, n_frames, W, H, sink)
    injected = [[pipes[g].pack_injected([dets[s][f] for s in range(bounds[g], bounds[g + 1])]) for f in range(n_frames)]
                for g in range(G)]
    torch.cuda.synchronize()

    # DD_BENCH_STEP_TIMES=1: when each worker group finished each of its steps (stderr, after the run; a diagnostic, not part of the line)
    step_times = [[] for _ in range(G)] if os.environ.get('DD_BENCH_STEP_TIMES') else None

    def run(g, f0, f1):
        torch.cuda.set_device(local_rank)                    # a fresh thread starts on device 0
        if ings is not None:
            det_streams = [p.detector_stream() for p in pipes]
            ings[g].submit(f0)
            for f in range(f0, f1):
                if f + 1 < f1:
                    ings[g].submit(f + 1)                         # the next step's upload runs under this step's kernels
                # the next frames are consumed first by their look-ahead detector run: ITS stream waits for their upload, not this step's kernels
                nxt = ings[g].frames(f + 1, stream=det_streams[g]) if f + 1 < f1 else None
                pipes[g].step(ings[g].frames(f), injected[g][f], nxt)
                ings[g].release(f)
            return
        ahead = os.environ.get('DD_BENCH_NO_LOOKAHEAD') is None
        for f in range(f0, f1):                                   # blocking C call, releases the GIL; the detector of
            pipes[g].step(dev_frames[g][f], injected[g][f],       # frame f+1 is queued behind this step's own detections
                          dev_frames[g][f + 1] if ahead and f + 1 < f1 else None)
            if step_times is not None:
                step_times[g].append(time.perf_counter())

    # Worker threads exist and have finished the warm-up before the clock starts: they park on `go`, the main thread
    # synchronises the device (and the ranks), takes t0 and releases them -- no thread start inside the timed region.
    warm = threading.Barrier(G + 1)
    go = threading.Event()
    errors = []

    def worker(g):
        try:
            run(g, 0, args.warmup)
        except BaseException as e:                           # noqa: BLE001 -- reported by the main thread
            errors.append(e)
        warm.wait()
        go.wait()
        if errors:
            return
        try:
            run(g, args.warmup, n_frames)
        except BaseException as e:                           # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(g,)) for g in range(G)]
    for t in th:
        t.start()
    warm.wait()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    go.set()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    if errors:
        raise errors[0]
    if step_times is not None and rank == 0:
        for g in range(G):
            ts = step_times[g][args.warmup:]
            print('group %d: ms per timed step: %s' % (g, ' '.join('%.1f' % (1e3 * (b - a)) for a, b in zip([t0] + ts[:-1], ts))), file=sys.stderr)
    stage_ms = pipes[0].stage_ms()
    local_counts = sum(p.counts().sum(axis=0) for p in pipes)
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    counts = reduce_counts(local_counts, device=red_dev)                  # RCCL: the only exchange step of the path
    dt = float(tmax.item())
    total_frames = args.steps * args.streams * world

    if rank == 0:
        out = {
            'metric': 'end-to-end frames/sec (detect+encode+track) at %dx%d' % (W, H),
            'value': total_frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'i8 (reference: uint8); encoder f16' if i8 else 'f16', 'data': 'synthetic',
            'config': {'workload': cfg['workload'], 'baseline_config': args.config,
                       'streams_per_gpu': args.streams, 'frames_per_step': args.streams * world,
                       'worker_threads_per_gpu': G,
                       'parallelism': 'independent streams, %d per GPU in %d worker groups' % (args.streams, G),
                       'weights': 'seeded synthetic (seed 1234)'},
            'counts_pos_neg_int_del': [int(v) for v in np.asarray(counts).reshape(-1)],
            # GPU milliseconds per step of ONE worker group (HIP events on the streams the kernels run on; the reference's timer names,
            # deepdish.py:975-981,1018-1021,1031-1032), `host` = the step's wall time outside its waits for the GPU.  The groups share the
            # GPU and objd runs a frame ahead beside the other stages: the stages need not add up to `wall`, nor `wall` to ms_per_step.
            'stage_ms_per_step': {k: (round(v, 4) if not isinstance(v, dict) else {a: round(b, 4) for a, b in v.items()})
                                  for k, v in stage_ms.items() if k != 'steps'},
        }
        if args.streams == 1:
            out['latency_ms_per_frame'] = 1e3 * dt / args.steps        # the reference's own operating point: one stream
        extra_only = False
        if args.background_subtraction is not None:
            out['background_subtraction'] = {'ratio': args.background_subtraction,
                                             'boxes_rejected_by_motion_test': int(sum(p.motion_mask(read=False)[1] for p in pipes)),
                                             'note': 'reference default configuration; not the headline (BASELINE configs disable it)'}
            extra_only = True
        if dist_on:
            out['rccl_ranks'] = int(comm_ranks)
            out['collective_backend'] = backend
        if ings is not None:
            out['frames_start_in'] = 'pinned host memory (PCIe upload inside the timed region; not the headline configuration)'
            step_bytes = args.streams * H * W * 3                   # per GPU and step, host -> device
            gbs = step_bytes / (1e6 * out['ms_per_step'])
            out['pcie'] = {'bytes_per_step': step_bytes, 'achieved': gbs, 'peak': 63.0, 'unit': 'GB/s', 'frac': gbs / 63.0,
                           'pinned_slots_per_group': n_frames, 'pinned_host_bytes': n_frames * step_bytes,
                           'bound_frames_per_s': 63.0e9 / (H * W * 3),
                           'copy_ceiling_measured': 57.4,
                           'note': 'PCIe Gen5 x16 spec 63 GB/s (MI355X_MICROARCH.md): the most a host can feed one GPU at %dx%d BGR; copy_ceiling_measured = pinned '
                                   'host -> device copies alone on an idle GPU of this pool (scripts/experiments/pcie_rates.py, 2 streams x 708 MB, either NUMA node)' % (W, H)}
            extra_only = True
        if not extra_only:
            try:
                from deepdish_amd.profile import dominant_kernel_roofline
                out['roofline'] = dominant_kernel_roofline(pipes, lambda g, f: pipes[g].step(dev_frames[g][f], injected[g][f]), args)
            except Exception as e:                            # never let the extra pass hide the headline number
                out['roofline'] = None
                out['roofline_error'] = repr(e)
            if world == 1 and not args.no_cpu_baseline:
                try:                                          # never let the extra passes hide the headline number
                    out['cpu_baseline'], (sc0, ocounts, otable) = cpu_baseline(args.cpu_frames, W, H, args.config)
                except Exception as e:
                    out['cpu_baseline'] = None
                    out['cpu_baseline_error'] = repr(e)
                    sc0 = None
                if sc0 is not None:
                    try:
                        per_group = bounds[1] - bounds[0]
                        del injected, dev_frames
                        pipes.clear()
                        torch.cuda.empty_cache()
                        out['parity_sample'] = gpu_sample_check(sc0, args.cpu_frames, ocounts, otable, f'cuda:{local_rank}', W, H,
                                                                cfg['model'], per_group)
                    except Exception as e:
                        out['parity_sample'] = dict(error=repr(e))
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
