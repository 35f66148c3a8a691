#!/bin/bash
# GPU box: PMC passes over the torch-free bench binary, counters restricted to our kernels.
# Usage: scripts/pmc.sh <tag> [fcp_bench args]
TAG=${1:-x}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BIN="$GRAFT_REPO_ROOT/recom_amd/fcp_bench --steps 40 --warmup 10 --verify 0 $*"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace --kernel-include-regex "fcp_" --output-format csv -d $OUT/p$i -- $BIN > $OUT/p$i.log 2>&1 || echo "pass $i failed: $line"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
LIST
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        m = re.search(r"fcp_\w+(<[^>]*>)?", name)
        acc[m.group(0) if m else name[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern, d in acc.items():
    print(kern)
    for k, v in d.items():
        print(f"   {k:<36s} n={len(v):>4d} mean={sum(v)/len(v):.6g}")
PY
find $OUT -name "*.csv" -size +1M -delete
