#!/bin/bash
# GPU box: kernel timelines of the serving modes (S2): who overlaps with whom
cd /tmp && export TMPDIR=/tmp
for m in stream inputs_ready workers3 workers4 lanes2 lanes3 lanes4; do
  rm -rf /tmp/tr_$m
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$m -- python3 $GRAFT_REPO_ROOT/scripts/r04_private_trace.py $m > /tmp/tr_$m.log 2>&1
  python3 $GRAFT_REPO_ROOT/scripts/r04_private_trace_summarize.py /tmp/tr_$m $m
done
