#!/bin/bash
# Round 3: where the PCIe-inclusive S2 request spends its time (fcp_bench --h2d 1 --narrow 1): stager phase timers
# (FCP_STAGER_STATS), host time of the two calls, by pack threads, pinned / unpinned workers, copy vs zero copy.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
export FCP_STAGER_STATS=1
for round in 1 2; do
  for t in 1 4 8 16 32; do
    echo "== round $round copy, $t pack threads"; ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --pack-threads $t 2>&1 | grep -E "pcie_inclusive|fcp_stager"
  done
  echo "== round $round copy, 8 threads, workers not pinned"; FCP_STAGER_NO_PIN=1 ./recom_amd/fcp_bench --h2d 1 --narrow 1 --steps 400 --warmup 50 --verify 0 --pack-threads 8 2>&1 | grep -E "pcie_inclusive|fcp_stager"
  echo "== round $round zero copy, 8 threads"; ./recom_amd/fcp_bench --h2d 1 --narrow 1 --zero-copy 1 --steps 400 --warmup 50 --verify 0 --pack-threads 8 2>&1 | grep -E "pcie_inclusive|fcp_stager"
done
echo "== host pack probe (SparseTensor indices -> row offsets, three forms; short bags, long bags)"
g++ -O3 -std=c++17 scripts/probes/pack_probe.cc -o /tmp/pack_probe && /tmp/pack_probe 10 | tail -2 && /tmp/pack_probe 150 | tail -1
