#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run13"; mkdir -p "$O"
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2> "$O/bench_driver.err"; echo "bench rc=$? wall ${SECONDS}s"
python3 -c "
import json; r=json.loads(open('$O/bench_driver.json').readline()); ro=r['roofline']; print(r['ms_per_step']*1e3, ro['frac'], ro['traffic'], ro['traffic_source'][:160]); print(ro.get('traffic_record'), ro.get('traffic_live_error'))"
tail -3 "$O/bench_driver.err"
