#!/bin/bash
# GPU box: default build (plain table loads) vs build/ntl (-DFCP_NT_LOADS), interleaved.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
one() { python bench.py --no-cpu-baseline --steps 1000 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step']*1e3,2), 'overlapped', round((r.get('overlapped') or {}).get('us_per_request',0),2))"; }
for round in 1 2; do
  for v in plain ntl; do
    if [ $v = ntl ]; then export FCP_LIB_DIR=$GRAFT_REPO_ROOT/build/ntl; else unset FCP_LIB_DIR; fi
    echo -n "$v S2 uniform : "; one
    echo -n "$v S2 zipf    : "; one --ids zipf
    echo -n "$v S2 b2048   : "; one --batch 2048
    echo -n "$v RAGGED     : "; one --workload ragged
    echo -n "$v DLRM       : "; one --workload dlrm
    echo -n "$v E          : "; one --workload e --requests 8
    echo -n "$v F          : "; one --workload f --requests 8
  done
done
