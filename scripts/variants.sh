#!/bin/bash
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_ROWS_PER_WAVE=4
for round in 1 2; do
for v in recom_amd build/nts build/ntl build/ntls; do echo -n "$v ring6: "; ./$v/fcp_bench --steps 500 --verify 0 | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done
done
for ring in 1 2 4; do echo -n "product ring$ring: "; ./recom_amd/fcp_bench --steps 500 --verify 0 --ring $ring | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done
for ring in 1 2; do echo -n "ntls ring$ring: "; ./build/ntls/fcp_bench --steps 500 --verify 0 --ring $ring | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done
