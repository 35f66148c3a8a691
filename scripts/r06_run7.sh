#!/bin/bash
# Round 6, run 7: after the nt-store fix — GPU suite; arena ring x workload (auto policy); why was model F's main timed loop
# slower than the arena_reuse loops of the same process (with / without the closed-form check of the resident requests)?
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run7"; mkdir -p "$O"
cat /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8 > "$O/cgroup_cpu.txt"; nproc >> "$O/cgroup_cpu.txt"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
for w in s2 ragged ragged-staged dlrm f; do
  python3 scripts/r06_arena_reuse.py --workload $w --rounds 2 --policy "product after the nt fix" 2>>"$O/err.log"
done | tee "$O/cells_after_nt_fix.jsonl" | python3 -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], {k:r[k]['plain_us'] for k in r if k.startswith('ring_')})"
one() { python3 bench.py --no-cpu-baseline --no-pcie --no-overlap "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline())
print('ms_per_step %.2f us kernel %.2f p50 %.2f repeats %s' % (r['ms_per_step']*1e3, r['roofline']['kernel_avg_us'], r['p50_latency_ms']*1e3, [round(x*1e3,2) for x in r['repeats']['ms_per_step_all']]))"; }
for i in 1 2; do
  echo -n "F verify: "; one --workload f
  echo -n "F --no-verify: "; one --workload f --no-verify
  echo -n "E verify: "; one --workload e
  echo -n "E --no-verify: "; one --workload e --no-verify
done 2>&1 | tee "$O/ef_main_loop.txt"
cat "$O/cgroup_cpu.txt"
