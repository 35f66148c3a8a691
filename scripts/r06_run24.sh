#!/bin/bash
# big soak of every fuzz family on fresh seeds (not -x)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run24"; mkdir -p "$O"
FCP_FUZZ_SEED0=1000 FCP_FUZZ_SEEDS=4000 FCP_FUZZ_SHARD_SEEDS=600 FCP_FUZZ_FINALIZE_SEEDS=400 FCP_FUZZ_STAGER_SEEDS=800 FCP_FUZZ_REGULAR_SEEDS=1500 \
  timeout 2600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/soak.log" 2>&1; echo "soak rc=$? $(grep -h 'passed\|failed' "$O/soak.log" | tail -1)"
grep -n "^FAILED" "$O/soak.log" | head -40
