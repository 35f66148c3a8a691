#!/bin/bash
# GPU box: PCIe-inclusive S2 (host int64 ids -> stager (narrow) -> H2D -> kernel), phase timers per pack-thread count
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
B=${FCP_LIB_DIR:-./recom_amd}
for rep in 1 2; do
for pt in 1 4 8 16 32; do
  echo "== pack threads $pt"
  FCP_STAGER_STATS=1 $B/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --pack-threads $pt 2>&1 | grep "pcie_inclusive\|fcp_stager:"
done
done
nproc; lscpu | grep -i "model name\|socket\|numa node" | head -8
