#!/bin/bash
# the multi-process file x4 (watchdog test), then the whole suite once
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run21"; mkdir -p "$O"
for i in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_0_gpu_shard_ranks.py -m gpu -x -q > "$O/ranks_$i.log" 2>&1; echo "ranks $i rc=$? $(grep -h 'passed\|failed' "$O/ranks_$i.log" | tail -1)"
done
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/all.log" 2>&1; echo "all rc=$? $(grep -h 'passed\|failed' "$O/all.log" | tail -1)"
