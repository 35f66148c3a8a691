#!/bin/bash
# Round 3, second look at the PCIe-inclusive S2 request: which runtime call of the enqueue phase takes the time, and does it
# follow the spinning pack workers (FCP_PACK_SPINS) or the ring depth?
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_STAGER_STATS=1
run() { echo "== $*"; "$@" 2>&1 | grep -E "pcie_inclusive|fcp_stager"; }
B="./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0"
for round in 1 2; do
  run $B --pack-threads 4
  run $B --pack-threads 8
  FCP_PACK_SPINS=0 run $B --pack-threads 8
  FCP_PACK_SPINS=2000 run $B --pack-threads 8
  run $B --pack-threads 8 --stager-depth 8
  run $B --pack-threads 8 --stager-depth 2
  GPU_MAX_HW_QUEUES=8 run $B --pack-threads 8
  HIP_FORCE_DEV_KERNARG=1 run $B --pack-threads 8
done
