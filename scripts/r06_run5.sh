#!/bin/bash
# Round 6, run 5: gather probe by load cache policy (+ PMC request counters); RAGGED with regular CSR inputs in the blob;
# CPU baseline by dataflow; the new multi-rank bench records.
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run5"; mkdir -p "$O"
export TMPDIR=/tmp
timeout 300 ./build/gather_policy_probe 8 10 2>&1 | tee "$O/gather_policy_probe.txt"
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_REQ_sum"; do
  d="$O/pmc_$(echo $c | tr ' ' '_')"
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -o p -- ./build/gather_policy_probe 8 1 > "$d.log" 2>&1
  f=$(find "$d" -name '*counter_collection.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a "$O/gather_policy_pmc.txt"
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.OrderedDict()
for r in rows:
    k=(int(r.get("Dispatch_Id",0)), r.get("Kernel_Name","")[:40], r.get("Counter_Name"))
    acc[k]=acc.get(k,0)+float(r.get("Counter_Value",0))
for (d,k,c),v in acc.items(): print(d,k,c,int(v))
PY
  rm -rf "$d"
done
python3 scripts/r06_ragged_grouped.py 2>&1 | grep -v amdgpu.ids | tee "$O/ragged_grouped.txt"
timeout 900 python -m pytest tests/test_0_gpu_shard_ranks.py tests/test_gpu_shard.py -x -q -k "bench_gpus_2 or side_record" > "$O/shard_tests.log" 2>&1; echo "shard tests rc=$?"; tail -15 "$O/shard_tests.log"
{
python3 scripts/r06_cpu_baseline_probe.py --dataflow 0 --label "fused layout (checker's form), dram tables"
python3 scripts/r06_cpu_baseline_probe.py --dataflow 1 --label "TF-CPU dataflow (column tensors + ConcatV2), dram tables"
python3 scripts/r06_cpu_baseline_probe.py --dataflow 1 --vocab 2000 --label "TF-CPU dataflow, cache-resident tables"
} 2>&1 | tee "$O/cpu_probe_dataflow.jsonl"
