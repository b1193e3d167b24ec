#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the event-list voxelisers (tools/events_time.py).  usage: profile_events.sh <tag>
TAG=${1:-r02b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG/events
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python3 $REPO/tools/events_time.py > $OUT/events_time.jsonl 2> $OUT/time.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/tools/events_time.py > /dev/null 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats
cat $OUT/events_time.jsonl; head -8 $OUT/kernel_stats.csv | cut -c1-160
