#!/bin/bash
# Wave-cycle breakdown and instruction mix of one network forward (gpurun from the repo root):
#   bash scripts/pmc_sq_net.sh <kind> <batch> <tag>         kind: ssd | ssd_i8 | mars | yolo
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
K=$1; B=$2; T=$3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq_$T -- python3 $R/scripts/profile_layers.py $K $B > $O/sq_$T.log 2> $O/sq_$T.err
python3 $R/scripts/summarize_pmc_sq.py $O/sq_$T $O/pmc_sq_$T.json "profile_layers.py $K $B" > $O/pmc_sq_$T.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $O/sqi_$T -- python3 $R/scripts/profile_layers.py $K $B > $O/sqi_$T.log 2> $O/sqi_$T.err
python3 $R/scripts/summarize_pmc_sq.py $O/sqi_$T $O/pmc_sqi_$T.json "profile_layers.py $K $B" > $O/pmc_sqi_$T.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mf_$T -- python3 $R/scripts/profile_layers.py $K $B > $O/mf_$T.log 2> $O/mf_$T.err
python3 $R/scripts/summarize_pmc_sq.py $O/mf_$T $O/pmc_mfma_$T.json "profile_layers.py $K $B" > $O/pmc_mfma_$T.txt
find $O -name '*counter_collection.csv' -size +20M -delete 2>/dev/null
cat $O/pmc_sq_$T.txt | head -12; cat $O/pmc_mfma_$T.txt | head -8
