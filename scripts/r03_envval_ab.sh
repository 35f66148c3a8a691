#!/bin/bash
# GPU box: one library, an environment variable at several values, interleaved.  Usage: r03_envval_ab.sh VAR "v1 v2 ..." "<bench args>"...
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; VALS=$2; shift 2
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-verify --steps 800 --warmup 100 $* 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for w in "$@"; do
    for v in $VALS; do
      echo -n "round $round [$w] $VAR=$v: "
      if [ "$v" = default ]; then unset $VAR; else export $VAR=$v; fi
      one $w
    done
  done
done
