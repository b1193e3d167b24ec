#!/bin/bash
# Run ON THE GPU BOX from the repo root: bash tools/profile_v2e.sh <tag> [f32|u8] ["variant name"]
# SQ instruction counters of the v2e kernels for one feature variant of tools/v2e_breakdown.py.
set -u
TAG=${1:-v2e}; DT=${2:-f32}; VAR=${3:-all features (cfg 3)}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/pmc_a -o a -- python3 $REPO/tools/v2e_breakdown.py $DT "$VAR" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_b -o b -- python3 $REPO/tools/v2e_breakdown.py $DT "$VAR" > $OUT/b.log 2>&1
cd $REPO
python3 tools/pmc_kernels.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
grep v2e $OUT/summary.txt
