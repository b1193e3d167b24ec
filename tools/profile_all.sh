#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/profile_all.sh <tag> "<workloads>"
# For every workload: (1) rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W`, (2) separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (they do not fit one pass on gfx950), (3) one SQ pass (instruction counts).  Small summaries go to
# gpurun_out/prof_<tag>/<workload>/ -- copy what is to be judged into profiles/<tag>/ and commit it.
set -u
TAG=${1:-r02}
WLS=${2:-cfg2_esim_f32_256x32x256x256_bilinear5}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for WL in $WLS; do
  OUT=$REPO/gpurun_out/prof_$TAG/$WL
  mkdir -p $OUT
  cd /tmp
  ARGS="$REPO/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-also --workload $WL"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/bench_under_stats.json 2> $OUT/stats.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $ARGS > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 $ARGS > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -o sq -- python3 $ARGS > $OUT/bench_under_pmc_sq.json 2> $OUT/pmc_sq.err
  cd $REPO
  python3 tools/summarize_workload.py $OUT $WL > $OUT/summary.json 2> $OUT/summary.err
  cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
  find $OUT -name "*.csv" -size +1M -delete
  rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
  python3 -c "
import json; d=json.load(open('$OUT/summary.json')); print('$WL', 'step_us(rocprof)', round(d.get('step_us_rocprof',0),1), 'bench_ms', d.get('bench',{}).get('kernel_ms_avg'), 'traffic/alg', d.get('traffic_over_algorithmic'))"
done
