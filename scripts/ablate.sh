#!/bin/bash
# GPU box: product build vs timing-only ablation builds of the same kernel (interleaved rounds).
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  echo "round $round"
  echo -n "product : "; ./recom_amd/fcp_bench --steps 500 --verify $((round==1)) | tail -1
  for a in 1 2 3; do
    echo -n "ablate $a: "; ./build/abl$a/fcp_bench --steps 500 --verify 0 | tail -1
  done
done
