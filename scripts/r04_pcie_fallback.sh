#!/bin/bash
# GPU box: the stager's zero-copy fallback against the blocking-hipMemcpyAsync anomaly (provoked with unpinned pack workers)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3 4 5 6 7 8; do
  for fb in on off; do
    if [ $fb = off ]; then export FCP_STAGER_NO_FALLBACK=1; else unset FCP_STAGER_NO_FALLBACK; fi
    FCP_STAGER_NO_PIN=1 FCP_STAGER_STATS=1 ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 600 --warmup 50 --verify 0 --pack-threads 16 2>&1 | grep "pcie_inclusive\|fcp_stager:" | python3 -c "
import sys, re, json
txt = sys.stdin.read()
st = re.findall(r'fcp_stager: (\d+) calls.*?pack ([0-9.]+) .*?hipMemcpyAsync ([0-9.]+),.*?blocked > 20 us: (\d+), zero-copy fallbacks (\d+)', txt)
rec = json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print(' ; '.join(f'{c} calls pack {p} memcpy {m} blocked {b} fallbacks {f}' for c, p, m, b, f in st), '| pipelined', rec['us_per_request_pipelined'], '| fallback $fb')"
  done
done
