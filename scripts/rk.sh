#!/bin/bash
cd $GRAFT_REPO_ROOT
./recom_amd/fcp_bench --steps 200 | grep verify
for cfg in "4 1" "4 2" "4 4" "2 1" "2 2" "2 4" "2 8" "1 4" "1 8" "1 16" "4 1" "4 2" "2 4" "2 8"; do set -- $cfg; echo -n "R=$1 K=$2: "; FCP_ROWS_PER_WAVE=$1 FCP_TILES_PER_BLOCK=$2 ./recom_amd/fcp_bench --steps 400 --verify 0 $EXTRA | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us \2/'; done
