#!/bin/bash
# Round 6, run 3: (a) CPU-baseline collapse probe on the box's host; (b) S2 at ring 1 / 6: product vs streamed table reads
# (-DFCP_NT_LOADS) vs the timing-only builds without table reads (abl1) / without stores (abl2), batch 512 and 4096.
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run3"; mkdir -p "$O"
us() { sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; }
for round in 1 2; do
  for v in recom_amd build/ntl build/abl1 build/abl2; do
    for ring in 1 6; do
      echo -n "round $round $v ring $ring b512: "; ./$v/fcp_bench --steps 1000 --ring $ring --verify 0 | tail -1 | us
    done
  done
done 2>&1 | tee "$O/ring_ablate.txt"
for v in recom_amd build/abl1 build/abl2; do
  for ring in 1 6; do
    echo -n "$v ring $ring b4096: "; ./$v/fcp_bench --steps 200 --ring $ring --batch 4096 --verify 0 | tail -1 | us
  done
done 2>&1 | tee -a "$O/ring_ablate.txt"
{
python3 scripts/r06_cpu_baseline_probe.py --label "dram tables, unpinned"
OMP_PROC_BIND=spread OMP_PLACES=cores python3 scripts/r06_cpu_baseline_probe.py --label "dram tables, OMP_PROC_BIND=spread OMP_PLACES=cores"
python3 scripts/r06_cpu_baseline_probe.py --vocab 2000 --label "cache-resident tables, unpinned"
python3 scripts/r06_cpu_baseline_probe.py --columns 50 --label "50 columns (6 GB), unpinned"
} 2>&1 | tee "$O/cpu_probe.jsonl"
