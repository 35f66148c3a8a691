#!/bin/bash
# GPU box: PCIe-inclusive S2 with 1 / 2 / 3 serve workers (each its own stager + stream), pack threads 8 / 16.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for w in 1 2 3; do
  for pt in 8 16; do
    echo "== fcp_bench --h2d 1 --narrow 1 --threads $w --pack-threads $pt"
    ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --threads $w --pack-threads $pt | grep pcie_inclusive
  done
done
echo "== zero copy, 1 and 2 workers"
./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --threads 1 --pack-threads 16 --zero-copy 1 | grep pcie_inclusive
./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --threads 2 --pack-threads 16 --zero-copy 1 | grep pcie_inclusive
done
