#!/bin/bash
# the regular-row-offsets fuzz at its suite size, then a soak of every fuzz family on fresh seeds (not -x: every failure is listed)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run23"; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k regular_row_offsets > "$O/regular.log" 2>&1; echo "regular rc=$? $(grep -h 'passed\|failed' "$O/regular.log" | tail -1)"
grep -n "^E \|FAILED" "$O/regular.log" | head -20
FCP_FUZZ_SEED0=100 FCP_FUZZ_SEEDS=200 FCP_FUZZ_SHARD_SEEDS=40 FCP_FUZZ_FINALIZE_SEEDS=30 FCP_FUZZ_STAGER_SEEDS=60 FCP_FUZZ_REGULAR_SEEDS=150 \
  timeout 2000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/soak.log" 2>&1; echo "soak rc=$? $(grep -h 'passed\|failed' "$O/soak.log" | tail -1)"
grep -n "^FAILED" "$O/soak.log" | head -30
