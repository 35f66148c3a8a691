#!/bin/bash
# fuzz soak under the library's alternative code paths (shipping switches and tuning keys): each variant, every fuzz family, fresh seeds
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run25"; mkdir -p "$O"
run() {   # name, then VAR=value ...
  name=$1; shift
  env "$@" FCP_FUZZ_SEED0=20000 FCP_FUZZ_SEEDS=500 FCP_FUZZ_SHARD_SEEDS=50 FCP_FUZZ_FINALIZE_SEEDS=30 FCP_FUZZ_STAGER_SEEDS=80 FCP_FUZZ_REGULAR_SEEDS=200 \
    timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > "$O/$name.log" 2>&1
  echo "$name rc=$? $(grep -h 'passed\|failed' "$O/$name.log" | tail -1)"
  grep -n "^FAILED" "$O/$name.log" | head -10
}
run prepass FCP_SEG_PREPASS=1
run search_always FCP_SEG_SEARCH_MAX_PAIRS=100000000000
run upload_kernel FCP_DYN_UPLOAD=kernel
run through_all FCP_STORE_THROUGH_BYTES=0
run through_never FCP_STORE_THROUGH_BYTES=1099511627776
run plain_stores FCP_DIAG=store_plain_reuse=2
run wide_rows FCP_DIAG=wide_rows
run rows_per_wave_1 FCP_DIAG=rows_per_wave=1
run rows_per_wave_2 FCP_DIAG=rows_per_wave=2
run dyn_general FCP_DIAG=dyn_general
run packed_scratch FCP_DIAG=csr_by_pos=0
run no_xcd_map FCP_DIAG=no_xcd_map
run stager_sdma_groups FCP_STAGER_COPY=sdma FCP_STAGER_GROUPS=2 FCP_DIAG=stager_groups_always
