#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run10"; mkdir -p "$O"
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err"; echo "bench rc=$? wall ${SECONDS}s"
python3 -c "
import json; r=json.loads(open('$O/bench_driver.json').readline()); print(r['ms_per_step']*1e3, r['roofline']['frac'], r['roofline']['traffic'], r['cpu_baseline']['value'], r['cpu_baseline']['cores'], r['config'])"
tail -3 "$O/bench_driver.err"
