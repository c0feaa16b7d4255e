#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately as MI355X_MICROARCH.md
prescribes) into per-kernel HBM bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled."""
import csv, glob, collections, json, os, sys

fetch_dir, write_dir, out = sys.argv[1:4]


def load(d):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(sorted(glob.glob(d + '/*/*counter_collection.csv') + glob.glob(d + '/*counter_collection.csv'), key=os.path.getmtime)[-1])):
        a = acc[r['Kernel_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


f, w = load(fetch_dir), load(write_dir)
res = {}
for k, (n, tot) in f.items():
    wn, wt = w.get(k, [1, 0.0])
    res[k] = dict(launches=n, fetch_kib_avg=tot / n, write_kib_avg=wt / max(wn, 1),
                  hbm_read_bytes_corrected=tot / n * 1024 * 2, hbm_write_bytes=wt / max(wn, 1) * 1024)
json.dump(dict(sorted(res.items(), key=lambda kv: -(kv[1]['hbm_read_bytes_corrected'] + kv[1]['hbm_write_bytes']) * kv[1]['launches'])),
          open(out, 'w'), indent=1)
for k, v in list(res.items())[:0]:
    pass
print('wrote', out, len(res), 'kernels')
