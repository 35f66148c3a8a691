#!/bin/bash
# interleaved A/B of two builds in one session: build/prev vs recom_amd
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2 3; do
  for v in build/prev recom_amd; do echo -n "$v: "; ./$v/fcp_bench --steps 600 --verify $((round==1)) $* | tail -1 | sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; done
done
