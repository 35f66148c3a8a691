#!/bin/bash
# GPU box: one library, an environment switch on / off, interleaved.  Usage: r03_env_ab.sh VAR "<bench args>"...
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; shift
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-verify --steps 800 --warmup 100 $* 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for w in "$@"; do
    echo -n "round $round [$w] $VAR=1: "; env $VAR=1 python3 -c "pass"; export $VAR=1; one $w; unset $VAR
    echo -n "round $round [$w] default: "; one $w
  done
done
