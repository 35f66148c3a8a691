#!/bin/bash
# RAGGED and the reference's models E / F: requests resident as they arrive (SparseTensor indices) vs as the staging step leaves
# them (--staged: ids int32, row ids -> CSR offsets on the host).  Single stream / overlapped, interleaved.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
one() { python3 bench.py --no-cpu-baseline --steps 800 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us (%.3f aggregate)' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan')), o.get('aggregate_frac_of_peak', float('nan'))))"; }
for round in 1 2; do
  for w in ragged e f; do
    echo -n "round $round $w as delivered: "; one --workload $w
    echo -n "round $round $w as staged   : "; one --workload $w --staged
  done
done
