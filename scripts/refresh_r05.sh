#!/bin/bash
# GPU box: regenerate the round-5 records under gpurun_out/refresh5/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/refresh5
rm -rf "${O:?}"; mkdir -p "$O"
fail() { echo "refresh_r05: $*" >&2; exit 1; }
# PMC passes first (their own runs: --pmc with --kernel-trace only): the bench lines below report
# roofline.traffic from profiles/traffic.json, which must describe the kernels of this build
bash scripts/pmc.sh refresh5 > $O/r05_s2_pmc_fcp_bench.txt 2>&1
bash scripts/pmc_py.sh refresh5_ragged --workload ragged --staged > $O/r05_ragged_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh5_ragged_ad --workload ragged > $O/r05_ragged_as_delivered_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh5_e --workload e > $O/r05_ae_model_e_pmc.txt 2>&1
python3 scripts/traffic_from_pmc.py $O r05 > /dev/null || fail "traffic_from_pmc.py"
cp profiles/traffic.json $O/traffic.json
# the driver's invocation (few steps) and the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_s2_driver_steps20.json 2> $O/bench_s2_driver.err
python bench.py > $O/r05_bench_s2.json 2> $O/bench_s2.err
python bench.py --ids zipf --no-cpu-baseline > $O/r05_bench_s2_zipf.json 2>/dev/null
python bench.py --workload ragged > $O/r05_bench_ragged.json 2>/dev/null               # (r5) headline = as delivered, `staged` + its host cost beside it
python bench.py --workload ragged --staged --no-cpu-baseline > $O/r05_bench_ragged_staged.json 2>/dev/null
python bench.py --workload ragged --seg csr --no-cpu-baseline > $O/r05_bench_ragged_csr.json 2>/dev/null
python bench.py --workload dlrm > $O/r05_bench_dlrm.json 2>/dev/null
python bench.py --workload e > $O/r05_bench_ae_model_e.json 2>/dev/null
python bench.py --workload f > $O/r05_bench_ae_model_f.json 2>/dev/null
python bench.py --workload e --requests 64 --no-cpu-baseline --no-overlap > $O/r05_bench_ae_model_e_64_shapes.json 2>/dev/null
# kernel traces (single stream: --no-overlap keeps every traced kernel alone on its stream); tag = file name, explicit
trace() { # trace <tag> <bench args...>
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag/trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 300 --warmup 50 --no-cpu-baseline --no-pcie --no-overlap > $O/trace_$tag.log 2>&1 )
  python3 scripts/summarize_prof.py $O/t_$tag > $O/r05_${tag}_kernel_trace_stats.txt 2>&1 || fail "summarize_prof.py $tag"
  grep -q "fcp_" $O/r05_${tag}_kernel_trace_stats.txt || fail "no fcp_ kernel in the trace summary of $tag"
}
trace s2 --workload s2
trace ragged --workload ragged --staged
trace ragged_as_delivered --workload ragged
trace e --workload e
# the reference's serving protocol over the private streams: T host threads x depth on ONE caller stream
python scripts/probes/caller_threads_grid.py s2 > $O/grid_s2.json 2>/dev/null
python scripts/probes/caller_threads_grid.py ragged > $O/grid_ragged.json 2>/dev/null
find $O -name "*.csv" -size +2M -delete
rm -rf "${O:?}"/t_*
ls -la $O
