#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
for cfg in "4 0" "4 8000" "4 14000" "4 20000" "4 28000" "2 0" "2 10000" "2 20000" "1 0"; do set -- $cfg; echo -n "R=$1 pad=$2: "; FCP_ROWS_PER_WAVE=$1 FCP_LDS_PAD=$2 ./recom_amd/fcp_bench --steps 400 --verify 0 | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done
done
