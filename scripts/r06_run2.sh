#!/bin/bash
# Round 6, run 2: GPU suite on the three-policy stores; the host's auto policy by ring; small-output models by policy;
# the driver's own bench invocation (pre-warm + median of repeats; full-size CPU baseline).
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run2"; mkdir -p "$O"
free -g | head -2 > "$O/host.txt"; nproc >> "$O/host.txt"; cat /proc/sys/kernel/numa_balancing >> "$O/host.txt" 2>&1; lscpu | grep -i "numa\|socket\|model name" >> "$O/host.txt"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"
for w in s2 dlrm f; do
  for mode in 1 0 2; do
    FCP_STORE_PLAIN_REUSE=$mode python3 scripts/r06_arena_reuse.py --workload $w --policy "plain_reuse=$mode" 2>>"$O/err.log"
  done
done | tee "$O/cells2.jsonl"
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/bench_driver_steps20.json" 2> "$O/bench_driver.err"
tail -c 1500 "$O/bench_driver.err"
python3 - <<'PY'
import json,os
r=json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r06_run2/bench_driver_steps20.json")).readline())
print({k:r[k] for k in ("value","ms_per_step","extra_warmup_requests","repeats","arena_reuse") if k in r})
print(r["roofline"]["frac"], r["roofline"]["kernel_avg_us"])
print(json.dumps(r.get("cpu_baseline"))[:3000])
PY
tail -5 "$O/gputest.log"
