#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run18"; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; tail -3 "$O/gputest.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/r06_bench_s2_driver_steps20.json" 2>/dev/null
python3 -c "
import json; r=json.loads(open('$O/r06_bench_s2_driver_steps20.json').readline()); ro=r['roofline']; print('driver:', round(r['ms_per_step']*1e3,2), round(ro['kernel_avg_us'],2), round(ro['frac'],3), ro['traffic'], [round(x,2) for x in r['repeats']['kernel_avg_us_in_time_order']])"
