#!/bin/bash
# Round 6, after the staged ConcatInputs began to place its row offsets as one matrix: the records that form touches
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/refresh6b"; rm -rf "${O:?}"; mkdir -p "$O"
export TMPDIR=/tmp
python bench.py --workload ragged > "$O/r06_bench_ragged.json" 2>/dev/null
python bench.py --workload ragged --staged --no-cpu-baseline > "$O/r06_bench_ragged_staged.json" 2>/dev/null
python bench.py --workload e > "$O/r06_bench_ae_model_e.json" 2>/dev/null
python bench.py --workload f > "$O/r06_bench_ae_model_f.json" 2>/dev/null
bash scripts/pmc_py.sh refresh6b_ragged --workload ragged --staged > "$O/r06_ragged_pmc.txt" 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_ragged/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload ragged --staged --steps 300 --warmup 50 --no-cpu-baseline --no-pcie --no-overlap > "$O/trace_ragged.log" 2>&1 )
python3 scripts/summarize_prof.py "$O/t_ragged" > "$O/r06_ragged_kernel_trace_stats.txt" 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/r06_bench_s2_driver_steps20.json" 2>/dev/null
rm -rf "$O"/t_*; find "$O" -name "*.csv" -size +2M -delete
python3 - <<'PY'
import json,os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/refresh6b/"
for f in ("r06_bench_ragged.json","r06_bench_ragged_staged.json","r06_bench_ae_model_e.json","r06_bench_ae_model_f.json","r06_bench_s2_driver_steps20.json"):
    r=json.loads(open(O+f).readline()); ro=r["roofline"]
    print(f, round(r["ms_per_step"]*1e3,2), round(ro["kernel_avg_us"],2), round(ro["frac"],3), ro["traffic"], (r.get("staged") or {}).get("us_per_request"), [round(x,2) for x in r["repeats"]["kernel_avg_us_in_time_order"]])
PY
grep "fcp_" "$O/r06_ragged_kernel_trace_stats.txt" | cut -c1-160
