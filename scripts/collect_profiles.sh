#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh [part ...]      parts: tests layers trace pmc sq     (default: all)
# Everything lands under gpurun_out/prof/; scripts/summarize_*.py turn the counter CSVs into the JSON kept under profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DD_BENCH_GEN_WORKERS=1      # every profiled run generates its frames in-process: no generator pool forked from a process the profiler's preloaded library may have initialised the GPU in
PARTS=${*:-layers trace pmc sq}
S=${DD_PROF_STREAMS:-1536}     # streams of the one worker group the passes profile = frames per detector launch of the default bench (two groups of 1 536 since round 6)
B="python3 $R/bench.py --groups 1 --streams $S --steps 20 --warmup 5 --no-cpu-baseline"
for part in $PARTS; do
case $part in
tests)
    (cd $R && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -3 $O/gputests.log) ;;
layers)
    for kb in "ssd_i8 384" "ssd_i8 768" "ssd_i8 1536" "mars 15360" "mars 30720" "yolo 256"; do set -- $kb
        python3 $R/scripts/profile_layers.py $1 $2 > $O/layers_$1_b$2.txt 2>&1; tail -1 $O/layers_$1_b$2.txt; done ;;
trace)
    DD_BENCH_NO_LOOKAHEAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_g1 -- python3 $R/bench.py --groups 1 --streams $S --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_groups1_s$S.json 2> $O/kt_g1.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_traced.json 2> $O/kt_default.err
    echo trace done ;;
pmc)
    export DD_BENCH_GEN_WORKERS=1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.json 2> $O/pmc_fetch.err
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.json 2> $O/pmc_write.err
    python3 $R/scripts/summarize_pmc.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_s$S.json ;;
sq)
    export DD_BENCH_GEN_WORKERS=1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.json 2> $O/pmc_sq.err
    python3 $R/scripts/summarize_pmc_sq.py $O/pmc_sq $O/pmc_sq_bench_s$S.json "bench.py --groups 1 --streams $S --steps 20" > $O/pmc_sq_summary.txt
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $B > $O/pmc_mfma.json 2> $O/pmc_mfma.err
    python3 $R/scripts/summarize_pmc_sq.py $O/pmc_mfma $O/pmc_mfma_busy_bench_s$S.json "bench.py --groups 1 --streams $S --steps 20" > $O/pmc_mfma_summary.txt
    head -20 $O/pmc_mfma_summary.txt ;;
esac
done
# keep the merge-back small: the raw counter CSVs are large
# (gpurun copies back at most 64 MiB: the raw per-dispatch CSVs are summarised above and dropped)
find $O -name '*counter_collection.csv' -delete 2>/dev/null
find $O -name '*kernel_trace.csv' -delete 2>/dev/null
du -sh $O
