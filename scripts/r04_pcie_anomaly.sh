#!/bin/bash
# GPU box: how often does hipMemcpyAsync block inside the stager, and what does it depend on?  (S2, narrow, h2d copy)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
run() { # label, env..., args
  local label=$1; shift
  for rep in 1 2 3 4 5; do
    env "$@" FCP_STAGER_STATS=1 ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 300 --warmup 50 --verify 0 --pack-threads $PT 2>&1 | grep "pcie_inclusive\|fcp_stager:" | tr '\n' ' ' | sed -e 's/.*pack \([0-9.]*\) .*hipMemcpyAsync \([0-9.]*\),.*us_per_request_pipelined": \([0-9.]*\),.*/pack \1 memcpy \2 pipelined \3/'
    echo " | $label threads $PT"
  done
}
for PT in 8 16; do
  run default X=1
  run spins0 FCP_PACK_SPINS=0
  run nopin FCP_STAGER_NO_PIN=1
  run sdma_off HSA_ENABLE_SDMA=0
done
