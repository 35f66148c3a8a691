#!/bin/bash
# GPU box: blob prefetch of the ragged body on / off (FCP_BLOB_PREFETCH_MAX), interleaved, same library.
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --no-cpu-baseline --no-pcie --steps 800 --warmup 100 $* 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for w in "--workload ragged" "--workload ragged --seg csr" "--workload ragged --batch 1024"; do
    for pf in 0 4194304; do
      echo -n "round $round [$w] FCP_BLOB_PREFETCH_MAX=$pf: "; FCP_BLOB_PREFETCH_MAX=$pf one $w
    done
  done
done
