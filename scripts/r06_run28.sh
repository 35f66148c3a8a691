#!/bin/bash
# bench.py with its side records behind try/except: the line's fields as before; a side record made to fail leaves the line whole
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run28"; mkdir -p "$O"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver.json" 2>"$O/bench_driver.err"; echo "bench rc=$?"
python3 -c "
import json; r=json.loads(open('$O/bench_driver.json').readline()); ro=r['roofline']
print('driver:', round(r['ms_per_step']*1e3,2), round(ro['kernel_avg_us'],2), round(ro['frac'],3), ro['traffic'])
print(sorted(r.keys()))
print(sorted(ro.keys()))
print({k: ('error' in v) for k, v in r.items() if isinstance(v, dict) and k in ('arena_reuse','overlapped_serving','single_caller_stream','cpu_baseline','pcie_inclusive')})
print(r['cpu_baseline']['value'], r['cpu_baseline']['cores'], r['overlapped_serving']['us_per_request'], r['single_caller_stream']['us_per_request'], r['arena_reuse'])"
timeout 900 python -m pytest tests/test_0_gpu_shard_ranks.py -m gpu -x -q 2>&1 | tail -2
