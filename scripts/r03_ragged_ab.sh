#!/bin/bash
# Round 3: interleaved A/B of builds on the ragged path (bench.py, single stream + overlapped): RAGGED as staged (the
# default: CSR offsets + int32 ids made by ConcatInputs on the host), as delivered (SparseTensor indices on the device),
# with long bags, and the reference's model E.  Usage: r03_ragged_ab.sh [build dirs...]   (default: build/r02 product)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
BUILDS=${*:-build/r02 product}
one() { local d=$1; shift; if [ "$d" != product ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  python3 bench.py --no-cpu-baseline --no-pcie --steps 600 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us/request (frac %.3f), p50 %.2f us, overlapped %.2f us' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], r['p50_latency_ms']*1e3, o.get('us_per_request', float('nan'))))"; }
for round in 1 2; do
  for d in $BUILDS; do
    echo -n "round $round $d ragged (staged)    : "; one $d --workload ragged
    echo -n "round $round $d ragged as delivered: "; one $d --workload ragged --as-delivered
    echo -n "round $round $d long bags (64 cols, U{0..300}, zipf): "; one $d --workload ragged --columns 64 --max-len 300 --ids zipf --no-overlap
    echo -n "round $round $d long bags (512 cols, U{0..100}): "; one $d --workload ragged --max-len 100 --no-overlap
    echo -n "round $round $d model E: "; one $d --workload e
  done
done
