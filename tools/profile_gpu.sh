#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_gpu.sh <tag> [workload]
# Collects (1) rocprofv3 --kernel-trace --stats of `python3 bench.py`, (2) HBM traffic counters in separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), and writes small summaries under
# gpurun_out/prof_<tag>/ that are then copied into profiles/ and committed.
set -u
TAG=${1:-r01}
WL=${2:-cfg2_esim_f32_256x32x256x256_bilinear5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="$REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o sq -- python3 $ARGS > $OUT/bench_under_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -o sq2 -- python3 $ARGS > $OUT/bench_under_pmc_sq2.json 2> $OUT/pmc_sq2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -o grbm -- python3 $ARGS > $OUT/bench_under_pmc_grbm.json 2> $OUT/pmc_grbm.err
cd $REPO
python3 tools/summarize_prof.py $OUT $WL > $OUT/summary.json 2> $OUT/summary.err
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/summary.json
