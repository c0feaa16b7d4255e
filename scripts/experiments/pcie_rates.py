#!/usr/bin/env python3
"""Host -> device copy rates of pinned memory on this box (what bounds bench.py --ingest-host): one and two copy streams, a few sizes,
the NUMA node of the GPU and of the process.  python scripts/experiments/pcie_rates.py"""
import glob, os, time
import torch

for p in glob.glob('/sys/class/drm/card*/device/numa_node'):
    print(p, open(p).read().strip())
print('nodes:', [os.path.basename(p) for p in glob.glob('/sys/devices/system/node/node*')])
try:
    print('cpus allowed:', len(os.sched_getaffinity(0)))
except Exception as e:
    print(e)


def rate(nbytes, streams, reps=8):
    hs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    ds = [torch.empty(nbytes, dtype=torch.uint8, device='cuda') for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for h in hs:
        h.fill_(7)
    for i in range(streams):
        with torch.cuda.stream(ss[i]):
            ds[i].copy_(hs[i], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(streams):
            with torch.cuda.stream(ss[i]):
                ds[i].copy_(hs[i], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return nbytes * streams * reps / dt / 1e9


for mb in (64, 256, 708):
    for st in (1, 2, 4):
        print('%4d MB x %d streams: %.1f GB/s' % (mb, st, rate(mb << 20, st)))
