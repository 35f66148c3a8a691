#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + HBM traffic counters for the S2 bench.
# Usage: scripts/profile_s2.sh <tag> [extra bench args]
set -u
TAG=${1:-r1}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-overlap $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1



cd $OUT
python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files
find $OUT -name "*.csv" -size +3M -delete
