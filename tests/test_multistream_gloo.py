"""CPU, world_size 2 over gloo: the N>1 path of bench.py (stream sharding + the one count
all-reduce) gives the same totals as a single process running every stream."""
import os
import numpy as np
import torch.multiprocessing as mp


def _stream_counts(s):
    from deepdish_amd.synth import Scene
    from oracle import deepsort_np as ds, countline_np as cl
    sc = Scene(seed=100 + s, n_obj=6, n_frames=40)
    trk = ds.Tracker(ds.Metric(0.2), max_age=60)
    cnt = cl.CountLine(sc.countline())
    for f in range(40):
        boxes, scores, _, feats = sc.detections(f)
        keep = ds.non_max_suppression(boxes, 0.6, scores)
        trk.predict(); trk.update([ds.Det(boxes[i], 'person', scores[i], feats[i]) for i in keep]); cnt.step(trk)
    return cnt.vector()


def _worker(rank, world, port, n_streams, q):
    import torch.distributed as dist
    from deepdish_amd.multistream import shard_streams, reduce_counts
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_streams(n_streams, rank, world)
    local = sum((_stream_counts(s) for s in mine), np.zeros((1, 4), dtype=np.int64))
    total = reduce_counts(local)
    dist.barrier()
    q.put((rank, mine, local.tolist(), total.tolist()))
    dist.destroy_process_group()


def test_sharded_counts_reduce_matches_single_process():
    n_streams, world = 5, 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = sum((_stream_counts(s) for s in range(n_streams)), np.zeros((1, 4), dtype=np.int64))
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    assert res[0][3] == res[1][3] == want.tolist()
    assert (np.array(res[0][2]) + np.array(res[1][2])).tolist() == want.tolist()
    assert want.sum() > 0
