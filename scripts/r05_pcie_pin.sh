#!/bin/bash
# Round 5: is the stager's pack bound by the memory bandwidth of the ONE NUMA node its workers are pinned to (next to the GPU)?
# Pinned (default) against unpinned workers (FCP_STAGER_NO_PIN=1), 8 / 16 / 32 / 48 threads, kernel copies, one copy per request.
cd "$GRAFT_REPO_ROOT"
B=./recom_amd/fcp_bench
lscpu | grep -E "NUMA node|Socket|Model name" | head -12
for rep in 1 2; do for pin in 0 1; do for th in 8 16 32 48; do
  echo -n "no_pin=$pin threads=$th: "
  if [ $pin = 1 ]; then export FCP_STAGER_NO_PIN=1; else unset FCP_STAGER_NO_PIN; fi
  FCP_STAGER_STATS=1 FCP_STAGER_GROUPS=1 $B --h2d 1 --narrow 1 --steps 700 --warmup 50 --verify 0 --pack-threads $th 2>/tmp/err.txt | grep pcie_inclusive | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pipelined %.1f us  lone %.1f us  stage call %.1f' % (r['us_per_request_pipelined'], r['us_latency_single'], r['host_us_stage_call']), end='  ')"
  grep -o "pack [0-9.]* ([0-9]* threads)" /tmp/err.txt | head -1
done; done; done
