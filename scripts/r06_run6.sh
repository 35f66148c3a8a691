#!/bin/bash
# Round 6, run 6: GPU suite on the carved library + streamed table reads; A/B of FCP_STREAM_TABLE_BYTES (0 = never stream =
# round 5's loads; default 4 MiB) on S2 / RAGGED / RAGGED staged / DLRM / model F, rings 1 and 6, interleaved.
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run6"; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
for round in 1 2; do
  for w in s2 ragged ragged-staged dlrm f; do
    for sb in 0 4194304; do
      FCP_STREAM_TABLE_BYTES=$sb python3 scripts/r06_arena_reuse.py --workload $w --rings 1,6 --rounds 2 --policy "stream_table_bytes=$sb" 2>>"$O/err.log"
    done
  done
done | tee "$O/stream_ab.jsonl" | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['policy'], 'ring1', r['ring_1']['plain_us'], 'ring6', r['ring_6']['plain_us'])"
