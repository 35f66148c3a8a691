#!/bin/bash
# Round 3: the long-bag line VERDICT r02 asked for, before (build/r02) and after (product):
#   bench.py --workload ragged --max-len 300 --ids zipf      (512 columns, bags U{0..300}, Zipf ids)
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for d in build/r02 product; do
  if [ "$d" != product ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/$d; else unset FCP_LIB_DIR; fi
  echo "== $d: python bench.py --workload ragged --max-len 300 --ids zipf --requests 4 --steps 200 --warmup 20 --no-cpu-baseline --no-overlap"
  python3 bench.py --workload ragged --max-len 300 --ids zipf --requests 4 --steps 200 --warmup 20 --no-cpu-baseline --no-overlap 2>/dev/null | tail -1
done
