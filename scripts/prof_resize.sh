#!/bin/bash
# Kernel time + HBM traffic + SQ counters of the batched Lanczos stretch (scripts/time_resize.py) on the GPU box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/lf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/scripts/time_resize.py ${1:-384}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ['fetch','write','sq','sq2']:
    for f in glob.glob('$O/'+d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:50]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
            n[(k,r['Counter_Name'])]+=1
        for k,v in acc.items():
            if 'lanczos' in k or 'band' in k:
                print(d, k, {c: round(x/n[(k,c)],1) for c,x in v.items()})
for f in glob.glob('$O/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lanczos' in r['Name'] or 'band' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'])
PY
find $O -name '*counter_collection.csv' -size +20M -delete 2>/dev/null
