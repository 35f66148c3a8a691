#!/bin/bash
# GPU box: regenerate the round-4 records under gpurun_out/refresh4/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/refresh4
rm -rf "${O:?}"; mkdir -p "$O"
fail() { echo "refresh_r04: $*" >&2; exit 1; }
# PMC passes first (their own runs: --pmc with --kernel-trace only): the bench lines below report
# roofline.traffic from profiles/traffic.json, which must describe the kernels of this build
bash scripts/pmc.sh refresh4 > $O/r04_s2_pmc_fcp_bench.txt 2>&1
bash scripts/pmc_py.sh refresh4_ragged --workload ragged > $O/r04_ragged_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh4_ragged_ad --workload ragged --as-delivered > $O/r04_ragged_as_delivered_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh4_e --workload e > $O/r04_ae_model_e_pmc.txt 2>&1
python3 scripts/traffic_from_pmc.py $O r04 > /dev/null || fail "traffic_from_pmc.py"
cp profiles/traffic.json $O/traffic.json
# the driver's invocation (few steps) and the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_s2_driver_steps20.json 2> $O/bench_s2_driver.err
python bench.py > $O/r04_bench_s2.json 2> $O/bench_s2.err
python bench.py --ids zipf --no-cpu-baseline > $O/r04_bench_s2_zipf.json 2>/dev/null
python bench.py --workload ragged > $O/r04_bench_ragged.json 2>/dev/null
python bench.py --workload ragged --as-delivered --no-cpu-baseline > $O/r04_bench_ragged_as_delivered.json 2>/dev/null
python bench.py --workload ragged --seg csr --no-cpu-baseline > $O/r04_bench_ragged_csr.json 2>/dev/null
python bench.py --workload dlrm > $O/r04_bench_dlrm.json 2>/dev/null
python bench.py --workload e > $O/r04_bench_ae_model_e.json 2>/dev/null
python bench.py --workload f > $O/r04_bench_ae_model_f.json 2>/dev/null
python bench.py --workload e --requests 64 --no-cpu-baseline --no-overlap > $O/r04_bench_ae_model_e_64_shapes.json 2>/dev/null
# kernel traces (single stream: --no-overlap keeps every traced kernel alone on its stream); tag = file name, explicit
trace() { # trace <tag> <bench args...>
  local tag=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag/trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 300 --warmup 50 --no-cpu-baseline --no-pcie --no-overlap > $O/trace_$tag.log 2>&1 )
  python3 scripts/summarize_prof.py $O/t_$tag > $O/r04_${tag}_kernel_trace_stats.txt 2>&1 || fail "summarize_prof.py $tag"
  grep -q "fcp_" $O/r04_${tag}_kernel_trace_stats.txt || fail "no fcp_ kernel in the trace summary of $tag"
}
trace s2 --workload s2
trace ragged --workload ragged
trace ragged_as_delivered --workload ragged --as-delivered
trace e --workload e
# one host thread, one caller stream, the plan's private streams: lanes x depth, next to the one-stream and 3-worker figures
python scripts/r04_private_sweep.py --combos 2x2,2x3,3x3,3x4 --nowait 0,1 > $O/r04_private_streams_sweep.txt 2>&1
python scripts/r04_private_sweep.py --workload ragged --combos 2x3,3x3,3x4 --nowait 0 >> $O/r04_private_streams_sweep.txt 2>&1
python scripts/r04_private_sweep.py --workload e --combos 2x3,3x3 --nowait 0 >> $O/r04_private_streams_sweep.txt 2>&1
find $O -name "*.csv" -size +2M -delete
rm -rf "${O:?}"/t_*
ls -la $O
