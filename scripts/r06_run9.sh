#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run9"; mkdir -p "$O"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
/usr/bin/time -v python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.time"; grep -E "Elapsed|Maximum resident" "$O/bench_driver.time"
python3 -c "
import json; r=json.loads(open('$O/bench_driver.json').readline()); print(r['ms_per_step']*1e3, r['roofline']['frac'], r['roofline']['traffic'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'])"
python3 scripts/r06_harness_order.py 2>&1 | grep -v amdgpu.ids | tee "$O/harness_order.txt"
