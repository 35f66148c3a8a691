#!/bin/bash
# GPU box: product build vs timing-only ablation builds of the same kernel (interleaved rounds).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  echo "round $round"
  echo -n "product : "; ./recom_amd/fcp_bench --steps 500 --verify $((round==1)) | tail -1
  for a in 1 2 3; do
    echo -n "ablate $a: "; ./build/abl$a/fcp_bench --steps 500 --verify 0 | tail -1
  done
done
