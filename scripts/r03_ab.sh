#!/bin/bash
# Round 3: interleaved A/B of builds through bench.py (single stream + overlapped).  Usage: r03_ab.sh "<build dirs>" "<workload args>"...
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
BUILDS=$1; shift
one() { local d=$1; shift; if [ "$d" != product ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  python3 bench.py --no-cpu-baseline --no-pcie --steps 800 --warmup 100 $* 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for w in "$@"; do
    for d in $BUILDS; do
      echo -n "round $round [$w] $d: "; one $d $w
    done
  done
done
