#!/bin/bash
# flake hunt: the timing-dependent file x8, the multi-process file x3, the whole suite x2, on one box
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run20"; mkdir -p "$O"
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python -m pytest tests/test_z_gpu_private_streams.py -m gpu -x -q > "$O/z_$i.log" 2>&1; echo "z $i rc=$? $(grep -h 'passed\|failed' "$O/z_$i.log" | tail -1)"
done
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_0_gpu_shard_ranks.py -m gpu -x -q > "$O/ranks_$i.log" 2>&1; echo "ranks $i rc=$? $(grep -h 'passed\|failed' "$O/ranks_$i.log" | tail -1)"
done
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > "$O/all_$i.log" 2>&1; echo "all $i rc=$? $(grep -h 'passed\|failed' "$O/all_$i.log" | tail -1)"
done
