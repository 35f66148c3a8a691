#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
 for w in e f; do
python3 bench.py --workload $w --no-cpu-baseline --no-pcie --no-overlap 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$w run $i (hipStreamSynchronize): median', round(r['ms_per_step']*1e3,2), 'kernel in time order', [round(x,2) for x in r['repeats']['kernel_avg_us_in_time_order']], 'p50', round(r['p50_latency_ms']*1e3,2))"
 done
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-overlap 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('S2 driver: median', round(r['ms_per_step']*1e3,2), 'kernel', [round(x,2) for x in r['repeats']['kernel_avg_us_in_time_order']], 'bracket', round(r['bracket_overhead_us'],1))"
