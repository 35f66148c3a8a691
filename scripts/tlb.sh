#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_ROWS_PER_WAVE=4
for round in 1 2; do
for v in 1000000 300000 100000 30000; do echo -n "vocab $v separate: "; ./recom_amd/fcp_bench --steps 300 --verify 0 --vocab $v | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us  frac \2/'; done
echo -n "vocab 1000000 slab: "; ./recom_amd/fcp_bench --steps 300 --verify 0 --slab 1 | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us  frac \2/'
done
