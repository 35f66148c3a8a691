#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
python3 bench.py --workload e --no-cpu-baseline --no-pcie --no-verify 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('E run $i: median', round(r['ms_per_step']*1e3,2), 'kernel in time order', [round(x,2) for x in r['repeats']['kernel_avg_us_in_time_order']], 'arena_reuse', {k:round(v,2) for k,v in r['arena_reuse'].items() if k.endswith('_us')}, 'p50', round(r['p50_latency_ms']*1e3,2))"
done
FCP_DIAG=install_stats python3 bench.py --workload e --no-cpu-baseline --no-pcie --no-verify --no-overlap 2>&1 >/dev/null | grep -i "install" | tail -5
