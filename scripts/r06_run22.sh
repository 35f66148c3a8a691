#!/bin/bash
# flake hunt under host load: 8 busy loops (half of the box's 16-CPU quota) beside the wall-clock file x4 and the whole suite x1
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run22"; mkdir -p "$O"
PIDS=""
for i in 1 2 3 4 5 6 7 8; do python3 -c "
import time
t=time.time()
while time.time()-t < 900: pass" & PIDS="$PIDS $!"; done
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_z_gpu_private_streams.py -m gpu -x -q > "$O/z_$i.log" 2>&1; echo "z $i rc=$? $(grep -h 'passed\|failed' "$O/z_$i.log" | tail -1)"
done
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/all.log" 2>&1; echo "all rc=$? $(grep -h 'passed\|failed' "$O/all.log" | tail -1)"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_loaded.json" 2>/dev/null
python3 -c "
import json; r=json.loads(open('$O/bench_loaded.json').readline()); ro=r['roofline']; print('driver under load:', round(r['ms_per_step']*1e3,2), round(ro['kernel_avg_us'],2), round(ro['frac'],3))"
kill $PIDS 2>/dev/null
wait 2>/dev/null
