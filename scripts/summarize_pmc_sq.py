#!/usr/bin/env python3
"""Sum the counters of one rocprofv3 --pmc pass per kernel and print the wave-cycle breakdown the CDNA4 guide
describes (SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue-stalled, SQ_ACTIVE_INST_* = issuing;
SQ_* wave counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles).
usage: summarize_pmc_sq.py <rocprof output dir> <out.json> [command string]"""
import csv, glob, collections, json, os, sys

d, out = sys.argv[1:3]
cmd = sys.argv[3] if len(sys.argv) > 3 else ''
path = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name']
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id'), k)
    if key not in seen:
        seen.add(key); n[k] += 1
res = {}
for k, c in acc.items():
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    e = dict(launches=n[k], **{a: b for a, b in c.items()})
    if wc:
        for name in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
                     'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_MISC', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_INST_LDS'):
            if name in c:
                e['frac_' + name[3:].lower()] = round(c[name] / wc, 4)
    if c.get('SQ_BUSY_CU_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        e['mfma_busy_fraction'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']), 4)
    res[k] = e
res = dict(sorted(res.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get('SQ_BUSY_CU_CYCLES', 0))))
json.dump(dict(command=cmd, kernels=res), open(out, 'w'), indent=1)
for k, e in list(res.items())[:14]:
    print(k[:90].ljust(90), {a: b for a, b in e.items() if a.startswith('frac_') or a in ('launches', 'mfma_busy_fraction')})
