#!/bin/bash
# final state: suite (skips listed), smoke, bench with no flags and by the driver's invocation
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run27"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q -rs > "$O/gputest.log" 2>&1; echo "gpu tests rc=$? $(grep -h 'passed\|failed' "$O/gputest.log" | tail -1)"; grep -n "SKIPPED" "$O/gputest.log"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; python bench.py > "$O/bench_default.json" 2>/dev/null; echo "bench (no flags) took $SECONDS s"
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2>/dev/null; echo "bench (driver) took $SECONDS s"
python3 -c "
import json
for f in ('bench_default','bench_driver'):
    r=json.loads(open('$O/'+f+'.json').readline()); ro=r['roofline']; print(f, round(r['ms_per_step']*1e3,2), round(ro['kernel_avg_us'],2), round(ro['frac'],3), ro['traffic'], round(r['value']/1e6,2))"
