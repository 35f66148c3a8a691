#!/bin/bash
# Round 3, third look: pack workers that sleep between requests and are woken early (PackPool::expect) against the
# round-2 behaviour (spinning through the gaps) and against plain sleeping workers.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_STAGER_STATS=1
run() { echo "== $*"; "$@" 2>&1 | grep -E "pcie_inclusive|fcp_stager"; }
B="./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0"
for round in 1 2 3; do
  for t in 4 8 16; do
    run $B --pack-threads $t
  done
  FCP_STAGER_NO_EARLY_WAKE=1 FCP_PACK_SPINS=0 run $B --pack-threads 8
  FCP_STAGER_NO_EARLY_WAKE=1 FCP_PACK_SPINS=65536 run $B --pack-threads 8
  FCP_PACK_SPINS=4096 run $B --pack-threads 8
  FCP_PACK_SPINS=256 run $B --pack-threads 8
  run $B --pack-threads 8 --zero-copy 1
done
