#!/bin/bash
# GPU box: regenerate the round-6 records under gpurun_out/refresh6/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/refresh6"
rm -rf "${O:?}"; mkdir -p "$O"
fail() { echo "refresh_r06: $*" >&2; exit 1; }
export TMPDIR=/tmp
# the GPU suite first: nothing below is worth keeping from a library that fails it
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; echo "gpu tests rc=$?" | tee -a "$O/gputest.log"; tail -3 "$O/gputest.log"
# PMC passes (their own runs: --pmc with --kernel-trace only).  S2 with ONE arena (what `value` times since round 6) and with six
bash scripts/pmc.sh refresh6 --ring 1 > "$O/r06_s2_pmc_fcp_bench.txt" 2>&1
bash scripts/pmc.sh refresh6_ring6 --ring 6 > "$O/r06_s2_ring6_pmc_fcp_bench.txt" 2>&1
bash scripts/pmc_py.sh refresh6_ragged --workload ragged --staged > "$O/r06_ragged_pmc.txt" 2>&1
bash scripts/pmc_py.sh refresh6_ragged_ad --workload ragged > "$O/r06_ragged_as_delivered_pmc.txt" 2>&1
bash scripts/pmc_py.sh refresh6_e --workload e > "$O/r06_ae_model_e_pmc.txt" 2>&1
python3 scripts/traffic_from_pmc.py "$O" r06 > /dev/null || fail "traffic_from_pmc.py"
cp profiles/traffic.json "$O/traffic.json"
# the driver's invocation (few steps) and the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > "$O/r06_bench_s2_driver_steps20.json" 2> "$O/bench_s2_driver.err"
python bench.py > "$O/r06_bench_s2.json" 2> "$O/bench_s2.err"
python bench.py --arena-ring 6 --no-cpu-baseline --no-pcie --no-overlap > "$O/r06_bench_s2_ring6.json" 2>/dev/null
python bench.py --ids zipf --no-cpu-baseline > "$O/r06_bench_s2_zipf.json" 2>/dev/null
python bench.py --workload ragged > "$O/r06_bench_ragged.json" 2>/dev/null
python bench.py --workload ragged --staged --no-cpu-baseline > "$O/r06_bench_ragged_staged.json" 2>/dev/null
python bench.py --workload ragged --seg csr --no-cpu-baseline > "$O/r06_bench_ragged_csr.json" 2>/dev/null
python bench.py --workload dlrm > "$O/r06_bench_dlrm.json" 2>/dev/null
python bench.py --workload e > "$O/r06_bench_ae_model_e.json" 2>/dev/null
python bench.py --workload f > "$O/r06_bench_ae_model_f.json" 2>/dev/null
# kernel traces (single stream: --no-overlap keeps every traced kernel alone on its stream)
trace() { # trace <tag> <bench args...>
  local tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/t_$tag/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" --steps 300 --warmup 50 --no-cpu-baseline --no-pcie --no-overlap > "$O/trace_$tag.log" 2>&1 )
  python3 scripts/summarize_prof.py "$O/t_$tag" > "$O/r06_${tag}_kernel_trace_stats.txt" 2>&1 || fail "summarize_prof.py $tag"
  grep -q "fcp_" "$O/r06_${tag}_kernel_trace_stats.txt" || fail "no fcp_ kernel in the trace summary of $tag"
}
trace s2 --workload s2
trace ragged --workload ragged --staged
trace ragged_as_delivered --workload ragged
trace e --workload e
# S2 by arena ring x store policy under a kernel trace: the native binary right after `--`
HUGE=4611686018427387904
for pol in plain_always nt sc1nt product; do
  for ring in 1 2 3 6; do
    d="$O/t_s2_${pol}_ring${ring}"
    case $pol in
      plain_always) export FCP_DIAG=store_plain_reuse=2; export FCP_STORE_THROUGH_BYTES=$HUGE;;
      nt)           export FCP_DIAG=store_plain_reuse=0; export FCP_STORE_THROUGH_BYTES=$HUGE;;
      sc1nt)        export FCP_DIAG=store_plain_reuse=0; export FCP_STORE_THROUGH_BYTES=0;;
      product)      unset FCP_DIAG; unset FCP_STORE_THROUGH_BYTES;;
    esac
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- "$GRAFT_REPO_ROOT/recom_amd/fcp_bench" --ring $ring --steps 600 --warmup 60 --verify 0 > "$d.log" 2>&1 )
    f=$(find "$d" -name '*kernel_stats.csv' | head -1)
    echo "== S2 $pol ring $ring: $(tail -1 "$d.log")"
    [ -n "$f" ] && grep "fcp_dense" "$f" | head -1
    rm -rf "$d" "$d.log"
  done
done > "$O/r06_arena_reuse_kernel_traces_fixed.txt" 2>&1
unset FCP_DIAG; unset FCP_STORE_THROUGH_BYTES
# the host's memory system and the CPU baseline by dataflow / worker count
{
./build/host_stream
python3 scripts/r06_cpu_baseline_probe.py --dataflow 0 --label "fused layout (checker's form), dram tables"
python3 scripts/r06_cpu_baseline_probe.py --dataflow 1 --label "TF-CPU dataflow (column tensors + ConcatV2), dram tables"
} > "$O/r06_cpu_baseline_collapse_raw.txt" 2>&1
find "$O" -name "*.csv" -size +2M -delete
rm -rf "${O:?}"/t_*
ls -la "$O"
