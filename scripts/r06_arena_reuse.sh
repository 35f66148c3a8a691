#!/bin/bash
# Round 6: arena ring {1,2,3,6} x output-store policy {plain, nt, sc1 nt}, S2 and RAGGED, stream order,
# with and without the consumer kernel; then a rocprofv3 kernel trace per S2 cell (native fcp_bench right after `--`).
# Needs build/plain (make -C recom_amd/csrc OUT=../../build/plain DEFS=-DFCP_NO_NT), built in the container.
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_arena"
mkdir -p "$O"
export TMPDIR=/tmp
HUGE=4611686018427387904
run() { # label  FCP_STORE_THROUGH_BYTES  lib-dir-or-""  workload
  if [ -n "$3" ]; then
    FCP_STORE_THROUGH_BYTES=$2 FCP_LIB_DIR="$GRAFT_REPO_ROOT/$3" python3 scripts/r06_arena_reuse.py --workload "$4" --policy "$1" 2>>"$O/err.log"
  else
    FCP_STORE_THROUGH_BYTES=$2 python3 scripts/r06_arena_reuse.py --workload "$4" --policy "$1" 2>>"$O/err.log"
  fi
}
for w in s2 ragged ragged-staged; do
  run plain  $HUGE build/plain $w
  run nt     $HUGE ""          $w
  run sc1nt  0     ""          $w
done | tee "$O/cells.jsonl"
# kernel traces, S2, one per cell: the program itself follows `--`
for pol in plain nt sc1nt; do
  for ring in 1 2 3 6; do
    d="$O/trace_s2_${pol}_ring${ring}"
    case $pol in
      plain) exe=./build/plain/fcp_bench; thr=$HUGE;;
      nt)    exe=./recom_amd/fcp_bench;   thr=$HUGE;;
      sc1nt) exe=./recom_amd/fcp_bench;   thr=0;;
    esac
    export FCP_STORE_THROUGH_BYTES=$thr
    rocprofv3 --kernel-trace --stats -d "$d" -o t -- $exe --ring $ring --steps 600 --warmup 60 --verify 0 > "$d.log" 2>&1
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    echo "== S2 $pol ring $ring: $(tail -1 "$d.log")"
    [ -n "$f" ] && head -4 "$f"
    rm -rf "$d"
  done
done | tee "$O/traces.txt"
