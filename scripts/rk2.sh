#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 2048 8192; do for k in 1 2 4 8 16; do echo -n "batch $b K=$k: "; FCP_TILES_PER_BLOCK=$k ./recom_amd/fcp_bench --steps 200 --verify 0 --batch $b | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us \2/'; done; done
