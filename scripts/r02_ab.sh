#!/bin/bash
# Round 2: interleaved A/B of builds on S2 (fcp_bench, single stream), then the block timeline of the
# current build (diagnostic build/stamps).  Usage: r02_ab.sh [build dirs...]   (default: build/r01 recom_amd)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
BUILDS=${*:-build/r01 recom_amd}
us() { sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; }
for round in 1 2 3; do
  for v in $BUILDS; do
    echo -n "round $round $v b512: "; ./$v/fcp_bench --steps 1000 --verify $((round==1)) | tail -1 | us
  done
done
for v in $BUILDS; do
  echo -n "$v b2048: "; ./$v/fcp_bench --steps 300 --batch 2048 --verify 0 | tail -1 | us
  echo -n "$v b128: "; ./$v/fcp_bench --steps 1000 --batch 128 --verify 0 | tail -1 | us
  echo -n "$v E-like one-hot (980 bucketize columns dim 8, vocab 101): "; ./$v/fcp_bench --steps 1000 --columns 980 --dim 8 --vocab 101 --bucketize-every 1 --verify 0 | tail -1 | us
  echo -n "$v 3 threads: "; ./$v/fcp_bench --steps 600 --threads 3 --verify 0 | tail -1 | sed 's/.*"wall_us_per_step": \([0-9.]*\).*/\1/'
done
echo "== stamps build"
./build/stamps/fcp_bench --steps 200 --verify 1 | tail -19
