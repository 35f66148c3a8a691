#!/bin/bash
# Round 5: PCIe-inclusive S2 (fcp_bench --h2d 1 --narrow 1), fresh process per run: copy engine (kernel = default | hipMemcpyAsync
# = SDMA | HSA_ENABLE_SDMA=0) x groups per request (pack / copy overlap inside one request; default: only when the caller is not
# issuing back to back), stalls counted per process.
cd "$GRAFT_REPO_ROOT"
B=./recom_amd/fcp_bench
run() { # label, env..., then args
  local label="$1"; shift
  local envs=(); while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  env "${envs[@]}" $B --h2d 1 --narrow 1 --steps 700 --warmup 50 --verify 0 --pack-threads 16 "$@" 2>/dev/null | grep pcie_inclusive | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: pipelined %.1f us  lone %.1f us  stage call %.1f  process call %.1f  copy calls %d  >1ms %d  slowest %.0f us  fallbacks %d' % (r['us_per_request_pipelined'], r['us_latency_single'], r['host_us_stage_call'], r['host_us_process_call'], r['copy_calls'], r['copy_calls_over_1ms'], r['max_copy_call_us'], r['zero_copy_fallback_switches']))"
}
for rep in 1 2 3 4 5 6 7 8 9 10; do
  run "default (kernel copies, groups when idle)" A=1
done
for rep in 1 2 3 4; do
  run "kernel, groups always " FCP_STAGER_GROUPS_ALWAYS=1
  run "kernel, one copy      " FCP_STAGER_GROUPS=1
  run "sdma,   one copy      " FCP_STAGER_GROUPS=1 --copy-kernel 0
  run "sdma off (blit), one  " FCP_STAGER_GROUPS=1 HSA_ENABLE_SDMA=0 --copy-kernel 0
done
echo "== serve workers (each its own stager, stream and ring)"
for t in 2 3; do run "threads=$t default" A=1 --threads $t; done
for t in 2 3; do run "threads=$t sdma   " A=1 --threads $t --copy-kernel 0; done
