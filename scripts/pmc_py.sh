#!/bin/bash
# GPU box: PMC passes over bench.py (counters restricted to fcp_* kernels). Usage: scripts/pmc_py.sh <tag> <bench args>
TAG=${1:-x}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcpy_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-overlap $*"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace --kernel-include-regex "fcp_" --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1 || echo "pass $i failed: $line"
done <<'LIST'
GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
FETCH_SIZE
WRITE_SIZE
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
