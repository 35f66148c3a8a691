#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_ROWS_PER_WAVE=4
./recom_amd/fcp_bench --copy-probe 1024; ./recom_amd/fcp_bench --copy-probe 4096; ./recom_amd/fcp_bench --copy-probe 64
for v in 1000000 3000 1000 100; do echo -n "vocab $v: "; ./recom_amd/fcp_bench --steps 300 --verify 0 --vocab $v | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*"frac_of_8TBs": \([0-9.]*\).*/\1 us  frac \2/'; done
