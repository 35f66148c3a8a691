#!/bin/bash
# PCIe-inclusive S2 (host id tensors -> stager -> kernel): H2D copy vs the kernel reading the pinned ring itself.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  for pt in 8 16; do
    echo -n "round $round copy      pack-threads $pt: "; ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --pack-threads $pt | tail -1
    echo -n "round $round zero-copy pack-threads $pt: "; ./recom_amd/fcp_bench --h2d 1 --narrow 1 --zero-copy 1 --steps 400 --warmup 50 --verify 0 --pack-threads $pt | tail -1
  done
done
