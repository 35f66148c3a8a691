#!/bin/bash
# Round 6, run 15: the staged ConcatInputs places its row-offset arrays as one matrix behind the other inputs — GPU suite, then
# RAGGED staged / as delivered / E / F
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run15"; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-verify --no-overlap --steps 1500 --warmup 200 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); s=r.get('staged') or {}; a=r.get('as_delivered') or {}
print('%.2f us/request (frac %.3f)  staged %.2f  as delivered %.2f' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], s.get('us_per_request', float('nan')), a.get('us_per_request', float('nan'))))"; }
for round in 1 2; do
  echo -n "round $round RAGGED --staged: "; one --workload ragged --staged
  echo -n "round $round RAGGED: "; one --workload ragged
  echo -n "round $round E: "; one --workload e
  echo -n "round $round F: "; one --workload f
done 2>&1 | tee "$O/staged_matrix.txt"
