#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run11"; mkdir -p "$O"
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pcie --no-overlap 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('fresh run $i: median', round(r['ms_per_step']*1e3,2), 'kernel in time order', [round(x,2) for x in r['repeats']['kernel_avg_us_in_time_order']], 'extra warmup', r['extra_warmup_requests'])"
done 2>&1 | tee "$O/driver_repeats_in_order.txt"
