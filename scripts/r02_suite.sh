#!/bin/bash
# Round 2: the other workloads (DLRM, RAGGED, reference models E / F) on the round-1 build and on the
# current build, through bench.py (single stream, no CPU baseline), device us per request.
cd $GRAFT_REPO_ROOT
one() { # $1 = lib dir ("" = product), rest = bench args
  local d=$1; shift
  if [ -n "$d" ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  python3 bench.py --no-cpu-baseline --steps 600 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), 3 workers %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], o.get('us_per_request', float('nan'))))"
}
for w in dlrm ragged e f; do
  for d in build/r01 ""; do echo -n "$w ${d:-product}: "; one "$d" --workload $w; done
done
for b in 64 128 256; do
  for d in build/r01 ""; do echo -n "s2 batch $b ${d:-product}: "; one "$d" --workload s2 --batch $b; done
done
