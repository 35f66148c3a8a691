#!/bin/bash
# Stage call of S2 (fcp_bench --h2d 1 --narrow 1) against how long woken pack workers keep spinning for the next job
# (FCP_PACK_SPINS pauses; default 1024 ~ 20-40 us) and the pool size.  FCP_STAGER_STATS=1 prints the phases.
cd $GRAFT_REPO_ROOT
for spins in 1024 100000 3000000; do
  for th in 8 16 32; do
    echo "== FCP_PACK_SPINS=$spins pack threads $th"
    FCP_STAGER_STATS=1 FCP_PACK_SPINS=$spins ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 100 --verify 0 --pack-threads $th 2>&1 | grep -E "fcp_stager:|pcie_inclusive" | cut -c1-330
  done
done
