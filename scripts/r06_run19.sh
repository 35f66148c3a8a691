#!/bin/bash
# full GPU suite (skips listed) + the watchdog test of the sharded side record by itself
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run19"; mkdir -p "$O"
timeout 1500 python -m pytest tests -m gpu -x -q -rs > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed\|SKIPPED" "$O/gputest.log" | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
