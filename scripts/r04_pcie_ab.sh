#!/bin/bash
# GPU box: wake-up style x spin window of the stager's pack pool (S2, narrow, h2d copy), interleaved in one call
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for pt in 8 16; do
for fan in 0 2 4; do
for spins in 1024 8192 65536; do
  echo "== threads $pt fanout $fan spins $spins"
  FCP_PACK_FANOUT=$fan FCP_PACK_SPINS=$spins FCP_STAGER_STATS=1 ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --pack-threads $pt 2>&1 | grep "pcie_inclusive\|fcp_stager:" | sed -e 's/"h2d_copy_alone.*//' -e 's/(wait-event on the copy.*hipMemcpyAsync/(hipMemcpyAsync/' -e 's/, event record.*//' | cut -c1-260
done; done; done; done
