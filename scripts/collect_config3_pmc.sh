#!/bin/bash
# HBM traffic per kernel of the YOLOv5s pipeline (bench.py --config 3, one group of 256 streams): gpurun_out/prof3/ -> profiles/r03_pmc_traffic_config3.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DD_BENCH_GEN_WORKERS=1
B="python3 $R/bench.py --config 3 --groups 1 --streams 256 --steps 12 --warmup 3 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.json 2> $O/pmc_write.err
python3 $R/scripts/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_config3.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.json 2> $O/kt.err
find $O -name '*counter_collection.csv' -size +20M -delete 2>/dev/null
find $O -name '*kernel_trace.csv' -delete 2>/dev/null
du -sh $O
