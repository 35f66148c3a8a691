#!/bin/bash
# Round 6, run 4: GPU suite on the regular-CSR ragged front; RAGGED A/B (FCP_CSR_BY_POS=0 = packed scratch of round 5);
# plain-store threshold by batch; gather probe by load cache policy (+ PMC request counters).
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run4"; mkdir -p "$O"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-verify --no-overlap --steps 1500 --warmup 200 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); s=r.get('staged') or {}
print('%.2f us/request (frac %.3f)  staged %.2f' % (r['roofline']['kernel_avg_us'], r['roofline']['frac'], s.get('us_per_request', float('nan'))))"; }
for round in 1 2 3; do
  for v in 1 0; do
    echo -n "round $round RAGGED as delivered FCP_CSR_BY_POS=$v: "; FCP_CSR_BY_POS=$v one --workload ragged
    echo -n "round $round RAGGED csr FCP_CSR_BY_POS=$v: "; FCP_CSR_BY_POS=$v one --workload ragged --seg csr
  done
done 2>&1 | tee "$O/ragged_csr_by_pos_ab.txt"
us() { sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; }
for b in 512 640 768 1024 2048; do
  for mode in 0 2; do
    echo -n "S2 batch $b ring 1 FCP_STORE_PLAIN_REUSE=$mode: "; FCP_STORE_PLAIN_REUSE=$mode ./recom_amd/fcp_bench --steps 600 --ring 1 --batch $b --verify 0 | tail -1 | us
  done
done 2>&1 | tee "$O/plain_threshold.txt"
./build/gather_policy_probe 8 10 2>&1 | tee "$O/gather_policy_probe.txt"
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_REQ_sum"; do
  d="$O/pmc_$(echo $c | tr ' ' '_')"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- ./build/gather_policy_probe 8 1 > "$d.log" 2>&1
  f=$(find "$d" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$O/gather_policy_pmc.txt"
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.OrderedDict()
for r in rows:
    k=(r.get("Dispatch_Id"), r.get("Kernel_Name","")[:60], r.get("Counter_Name"))
    acc[k]=acc.get(k,0)+float(r.get("Counter_Value",0))
for (d,k,c),v in acc.items(): print(d,k,c,int(v))
PY
  rm -rf "$d"
done
