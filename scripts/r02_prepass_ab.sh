#!/bin/bash
# RAGGED with SparseTensor indices: builds of the segment-offset pre-pass, interleaved (single stream / overlapped).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
one() { local d=$1; shift; if [ "$d" != product ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  python3 bench.py --workload ragged --no-cpu-baseline --steps 800 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), overlapped %s' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], {k: round(v,2) for k,v in o.get('sweep_us_per_request',{}).items()}))"; }
for round in 1 2 3; do for d in $*; do echo -n "round $round $d: "; one $d --seg indices; done; done
