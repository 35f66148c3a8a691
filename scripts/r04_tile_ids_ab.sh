#!/bin/bash
# GPU box: upper bound of a block-tile-major id layout (timing-only build FCP_ABLATE=6: every block requests its id words at
# t = 0 from an address that needs only the block index) against the product, interleaved, torch-free bench binary.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3 4; do
  for v in build/abl6 recom_amd; do echo -n "$v: "; ./$v/fcp_bench --steps 1200 --warmup 200 --verify 0 | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1 us per request/'; done
done
