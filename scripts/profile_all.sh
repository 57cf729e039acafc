#!/bin/bash
# rocprofv3 passes + summaries for one workload (run through gpurun from the repo root):
#   scripts/profile_all.sh <tag> <neurons> <states> <channels per plan> [bench args...]
# The chain length / warm-up the plan chose are read from the bench line of the trace pass.
set -e
TAG=$1; NEU=$2; STA=$3; CHN=$4; shift 4
scripts/profile_bench.sh $TAG "$@"
OUT=gpurun_out/prof_$TAG
LINE=$(grep '^{' $OUT/trace.log | tail -1)
read SAMPLES BLOCK HALO <<< $(python3 -c "
import json,sys
d=json.loads(sys.argv[1]); c=d['config']; print(c['samples_per_channel'], c['block'], c.get('halo', 0) or 0)" "$LINE")
python3 profiles/summarize.py $TAG $OUT $SAMPLES $BLOCK $HALO $NEU $STA $CHN > $OUT/summary.md
echo "$LINE" > profiles/${TAG}_bench.json
cp profiles/${TAG}_summary.* profiles/${TAG}_kernel_stats.csv profiles/${TAG}_bench.json $OUT/ 2>/dev/null || true
tail -30 $OUT/summary.md
