#!/bin/bash
# Round 2: RAGGED, reference models E / F and DLRM through bench.py (single stream + the 3-worker pass, no
# CPU baseline), interleaved over builds.  Usage: r02_suite.sh [build dirs...] ("product" = recom_amd/)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
BUILDS=${*:-product}
one() { # $1 = lib dir or "product", rest = bench args
  local d=$1; shift
  if [ "$d" != product ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  python3 bench.py --no-cpu-baseline --steps 800 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, 3 workers %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"
}
for round in 1 2; do
  for w in ragged e f dlrm; do
    for d in $BUILDS; do echo -n "round $round $w $d: "; one "$d" --workload $w; done
  done
done
