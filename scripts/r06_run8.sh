#!/bin/bash
# Round 6, run 8: plain vs nt stores by ring for the outputs that FIT the L2s (model F 19.8 MB, RAGGED 15.7 MB, DLRM 3.5 MB)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run8"; mkdir -p "$O"
for round in 1 2; do
  for w in f ragged dlrm; do
    for mode in 0 2; do
      FCP_DIAG=store_plain_reuse=$mode python3 scripts/r06_arena_reuse.py --workload $w --rings 1,6 --rounds 2 --policy "store_plain_reuse=$mode" 2>>"$O/err.log"
    done
  done
done | tee "$O/small_outputs_plain_vs_nt.jsonl" | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['policy'], {k:(r[k]['plain_us'], r[k]['consumer_us']) for k in r if k.startswith('ring_')})"
