#!/bin/bash
# Round 6, run 14: the guided tail of the ragged launch (FCP_DIAG=ragged_tail=1) — parity, then A/B
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run14"; mkdir -p "$O"
# (tests that set FCP_DIAG themselves override it; the others run with the tail on)
FCP_DIAG=ragged_tail=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "ragged or mixed or fuzz or long_bags or golden" > "$O/tail_parity.log" 2>&1; echo "parity with the tail on rc=$?"; tail -2 "$O/tail_parity.log"
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-verify --no-overlap --steps 1500 --warmup 200 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); s=r.get('staged') or {}
print('%.2f us/request (frac %.3f)  staged %.2f' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], s.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for v in 1 0; do
    echo -n "round $round RAGGED as delivered ragged_tail=$v: "; FCP_DIAG=ragged_tail=$v one --workload ragged
    echo -n "round $round RAGGED csr ragged_tail=$v: "; FCP_DIAG=ragged_tail=$v one --workload ragged --seg csr
    echo -n "round $round RAGGED long bags (max-len 100) ragged_tail=$v: "; FCP_DIAG=ragged_tail=$v one --workload ragged --seg csr --max-len 100
  done
done 2>&1 | tee "$O/ragged_tail_ab.txt"
