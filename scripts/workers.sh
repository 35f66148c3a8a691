#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_ROWS_PER_WAVE=4
for t in 1 2 3 4; do echo -n "workers $t: "; ./recom_amd/fcp_bench --steps 400 --verify 0 --threads $t | tail -1 | sed 's/.*"wall_us_per_step": \([0-9.]*\), "dev_us_per_step": \([0-9.]*\).*/wall \1 us per request (aggregate), dev(stream0) \2/'; done
