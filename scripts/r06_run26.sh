#!/bin/bash
# harness on its creator's device in every worker thread: suite + smoke + the driver's invocation
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run26"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$? $(grep -h 'passed\|failed' "$O/gputest.log" | tail -1)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2>/dev/null
python3 -c "
import json; r=json.loads(open('$O/bench_driver.json').readline()); ro=r['roofline']; print('driver:', round(r['ms_per_step']*1e3,2), round(ro['kernel_avg_us'],2), round(ro['frac'],3), ro['traffic'], 'overlapped', round(r['overlapped_serving']['us_per_request'],2))"
