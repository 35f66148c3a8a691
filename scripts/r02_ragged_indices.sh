#!/bin/bash
# RAGGED (BASELINE configs[3]) with SparseTensor indices: pre-pass launch vs in-block search, single stream and overlapped.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
one() { python3 bench.py --workload ragged --no-cpu-baseline --steps 800 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us (%s workers)' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan')), o.get('serve_workers')))"; }
for round in 1 2; do
  echo -n "round $round csr: "; one --seg csr
  echo -n "round $round indices, pre-pass launch: "; one --seg indices
  echo -n "round $round indices, searched in the blocks: "; FCP_SEG_SEARCH_MAX_PAIRS=100000000 one --seg indices
  echo -n "round $round rowids32, pre-pass launch: "; one --seg rowids32
done
