#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 512 2048; do for r in 1 2 4 8; do echo -n "batch $b R=$r: "; FCP_ROWS_PER_WAVE=$r ./recom_amd/fcp_bench --steps 300 --verify 0 --batch $b | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us  frac \2/'; done; done
