#!/usr/bin/env python3
"""Timings of the BASELINE.json configurations that are parity-test cases rather than the bench line:
  config 3  YOLOv5s f16 forward + decode + NMS on 640x640 (single stream through the plugin; batched forward)
  config 4  tracker only, T = D = 256 (deep_sort predict/update, cosine + gate + IoU, count line)
  config 5  1280x720 streams through the multi-stream pipeline (one GPU's share: 1 stream, and 8 for scale)
Each next to the oracle (CPU) on a bounded sample.  Writes one JSON object per line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def config3():
    from PIL import Image
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from deepdish_amd.tools.yolov5 import YOLOV5
    from oracle import nets_torch
    det = YOLOV5(model_file='synthetic-yolov5s', label_file=os.path.join(os.path.dirname(nets.__file__), 'assets', 'coco_classes.txt'))
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    dev = torch.from_numpy(frame).cuda()
    for _ in range(5):
        det.detect_frame_device(dev, 480, 640)
    t0 = time.perf_counter(); n = 100
    for _ in range(n):
        det.detect_frame_device(dev, 480, 640)
    plugin_fps = n / (time.perf_counter() - t0)
    net = Net(nets.compile_yolov5s(det.weights), max_batch=16)
    x = torch.randint(0, 256, (16, 640, 640, 3), dtype=torch.uint8, device='cuda')
    for _ in range(3):
        net.forward(x)
    net.ctx.sync(); t0 = time.perf_counter(); n = 20
    for _ in range(n):
        net.forward(x)
    net.ctx.sync()
    fwd_fps = 16 * n / (time.perf_counter() - t0)
    torch.set_num_threads(4)
    xi = rng.integers(0, 256, (1, 640, 640, 3), dtype=np.uint8)
    nets_torch.yolov5s_forward(det.weights, xi)
    t0 = time.perf_counter()
    for _ in range(3):
        nets_torch.yolov5s_forward(det.weights, xi)
    cpu_fps = 3 / (time.perf_counter() - t0)
    gf = sum(i['flops'] for i in net.program.info) / 1e9
    return dict(config=3, what='YOLOv5s f16 640x640', plugin_frames_per_s=plugin_fps, forward_batch16_frames_per_s=fwd_fps,
                forward_tflops=fwd_fps * gf / 1e3, gflop_per_frame=gf, cpu_oracle_forward_frames_per_s=cpu_fps, cpu_threads=4)


def config4():
    from deepdish_amd.deep_sort import nn_matching, preprocessing
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    from deepdish_amd.synth import tracker_scene
    from oracle import deepsort_np as ds
    sc = tracker_scene(seed=0, n_obj=256, n_frames=60)
    trk = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, None), max_iou_distance=0.7, max_age=60)
    otrk = ds.Tracker(ds.Metric(0.2), max_iou_distance=0.7, max_age=60)
    t_gpu = t_cpu = 0.0
    for f in range(60):
        boxes, scores, who, feats = sc.detections(f)
        keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
        dets = [Detection(boxes[i], 'person', scores[i], feats[i]) for i in keep]
        t0 = time.perf_counter(); trk.predict(); trk.update(dets); dt = time.perf_counter() - t0
        odets = [ds.Det(boxes[i], 'person', scores[i], feats[i]) for i in keep]
        t0 = time.perf_counter(); otrk.predict(); otrk.update(odets); dc = time.perf_counter() - t0
        if f >= 30:
            t_gpu += dt; t_cpu += dc
    same = [(t.track_id, t.state) for t in trk.tracks] == [(t.track_id, t.state) for t in otrk.tracks]
    # the same frames through the flat C ABI only (what the C++ pipeline calls): host tlwh, features already in HBM
    import ctypes
    from deepdish_amd._lib import lib, check, P
    from deepdish_amd.runtime import default_context, ptr
    ctx = default_context()
    h = P()
    check(lib().dd_tracker_create(ctx.handle, 0.2, 0.7, 60, 3, 0, 1024, 256, ctypes.byref(h)))
    t_abi = 0.0
    for f in range(60):
        boxes, scores, who, feats = sc.detections(f)
        keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
        tlwh = np.ascontiguousarray(boxes[keep], dtype=np.float64)
        fd = ctx.to_device(np.ascontiguousarray(feats[keep], dtype=np.float32))
        t0 = time.perf_counter()
        check(lib().dd_tracker_predict(h)); check(lib().dd_tracker_update(h, ptr(tlwh), ptr(fd), 1, len(keep)))
        if f >= 30:
            t_abi += time.perf_counter() - t0
    n = ctypes.c_int(); check(lib().dd_tracker_count(h, 0, ctypes.byref(n)))
    ints = np.zeros((n.value, 6), dtype=np.int64); check(lib().dd_tracker_read(h, 0, ptr(ints), None, None))
    same_abi = [(int(r[0]), int(r[1])) for r in ints] == [(t.track_id, t.state) for t in otrk.tracks]
    lib().dd_tracker_destroy(h)
    # SURVEY.md 8(d), algorithmic bytes per frame at T = D = 256 (the gallery is unbounded, deepdish.py:515: a track holds one sample per
    # frame it was matched in, so G = 256 x (frame + 1); frames 30-59 -> 11 648 rows on average)
    T = D = 256
    G = 256 * 45.5
    by = dict(kalman_predict=T * 1152, gating=T * 576 + D * 32 + T * D * 8, cosine=(G + D) * 512 + T * D * 8, iou=(T + D) * 32 + T * D * 8,
              kalman_update=min(T, D) * 1152)
    total_b, flops = sum(by.values()), 2 * G * D * 128
    ms = 1e3 * t_abi / 30
    roof = dict(bound='latency', algorithmic_bytes_per_frame=by, algorithmic_bytes_total=total_b, cosine_flops=flops,
                achieved_GBps=total_b / (ms * 1e-3) / 1e9, frac_of_8TBps=total_b / (ms * 1e-3) / 8e12,
                cosine_TFLOPs=flops / (ms * 1e-3) / 1e12,
                note='one stream: five dependent launches and two host round trips per frame (LSAP on the host); the bytes are 1e-3 of what the GPU '
                     'moves in that time -- the headline batches 384 such trackers per launch instead (csrc/tracker.hip)')
    return dict(config=4, what='tracker only, 256 targets, frames 30-59', roofline=roof,
                hip_c_abi_ms_per_frame=1e3 * t_abi / 30,
                hip_python_objects_ms_per_frame=1e3 * t_gpu / 30, cpu_oracle_ms_per_frame=1e3 * t_cpu / 30, tracks=len(trk.tracks),
                identical_ids_and_states=bool(same and same_abi))


def config5(S):
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    F = 30
    scenes = [Scene(seed=z, n_obj=20, width=1280, height=720, n_frames=F, vmax=6.0) for z in range(S)]
    mp = MultiStreamPipeline(S, input_size=(1280, 720))
    frames = [torch.from_numpy(np.stack([sc.frame(f) for sc in scenes])).cuda() for f in range(F)]
    inj = []
    for f in range(F):
        d = []
        for sc in scenes:
            boxes, scores, _, _ = sc.detections(f)
            d.append(([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(s) for s in scores]))
        inj.append(mp.pack_injected(d))
    for f in range(5):
        mp.step(frames[f], inj[f])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in range(5, F):
        mp.step(frames[f], inj[f])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(config=5, what='1280x720 streams on one GPU, %d stream(s) per step' % S, frames_per_s=S * (F - 5) / dt,
                ms_per_step=1e3 * dt / (F - 5), counts=[int(v) for v in mp.counts().sum(axis=(0, 1))])


if __name__ == '__main__':
    for fn in (config4, lambda: config5(1), lambda: config5(8)):
        print(json.dumps(fn()), flush=True)
