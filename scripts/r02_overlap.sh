#!/bin/bash
# Round 2: why does the 3-worker pass of bench.py (25.9 us) trail fcp_bench --threads 3 (23.2 us)?  Hardware
# queues (GPU_MAX_HW_QUEUES), and the sc1-nt store policy across all workloads.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
ov() { python3 bench.py --no-cpu-baseline --steps 600 --warmup 100 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); o=r.get('overlapped_serving') or {}
print('%.2f us single, 3 workers %.2f us' % (r['roofline']['kernel_avg_us'], o.get('us_per_request', float('nan'))))"; }
for q in "" 2 4 8; do
  echo -n "bench.py s2 GPU_MAX_HW_QUEUES=${q:-default}: "; if [ -n "$q" ]; then GPU_MAX_HW_QUEUES=$q ov; else ov; fi
done
for t in 2 3 4; do echo -n "fcp_bench --threads $t: "; ./recom_amd/fcp_bench --steps 600 --threads $t --verify 0 | tail -1 | sed 's/.*"wall_us_per_step": \([0-9.]*\).*/\1/'; done
for q in 2 8; do echo -n "fcp_bench --threads 3 GPU_MAX_HW_QUEUES=$q: "; GPU_MAX_HW_QUEUES=$q ./recom_amd/fcp_bench --steps 600 --threads 3 --verify 0 | tail -1 | sed 's/.*"wall_us_per_step": \([0-9.]*\).*/\1/'; done
echo "== store policy sc1 nt (build/st4) vs product, all workloads"
bash scripts/r02_suite.sh product build/st4
for round in 1 2 3; do for v in recom_amd build/st4; do echo -n "s2 fcp_bench round $round $v: "; ./$v/fcp_bench --steps 1000 --verify $((round==1)) | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done; done
