#!/bin/bash
# rocprofv3 passes of bench.py on the GPU box (run through gpurun from the repo root):
#   scripts/profile_bench.sh <tag> [bench args...]
# kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (the TCC block cannot
# hold both), then one SQ pass; raw output under gpurun_out/prof_<tag>/, summary via profiles/summarize.py
set -e
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --quick --steps 5 --warmup 2 $@"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o b --output-format csv -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o b --output-format csv -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/pmc_sq -o b --output-format csv -- python3 $ARGS > $OUT/sq.log 2>&1 || true
cd $ROOT
find $OUT -name "*.csv" | head -20
