#!/bin/bash
# Run ON THE GPU BOX from the repo root: bash tools/profile_cmd.sh <tag> <python script + args...>
# SQ / LDS / memory-unit counters per kernel for any python command (two --pmc passes), summary via tools/pmc_kernels.py.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/pmc_a -o a -- python3 "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_b -o b -- python3 "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc_c -o c -- python3 "$@" > $OUT/c.log 2>&1
cd $REPO
python3 tools/pmc_kernels.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
cat $OUT/summary.txt
