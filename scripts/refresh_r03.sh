#!/bin/bash
# GPU box: regenerate the round-3 records under gpurun_out/refresh3/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/refresh3
rm -rf "${O:?}"; mkdir -p "$O"
# PMC passes first (their own runs: --pmc with --kernel-trace only): the bench lines below report
# roofline.traffic from profiles/traffic.json, which must describe the kernels of this build
bash scripts/pmc.sh refresh3 > $O/r03_s2_pmc_fcp_bench.txt 2>&1
bash scripts/pmc_py.sh refresh3_ragged --workload ragged > $O/r03_ragged_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh3_ragged_ad --workload ragged --as-delivered > $O/r03_ragged_as_delivered_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh3_e --workload e > $O/r03_ae_model_e_pmc.txt 2>&1
python3 scripts/traffic_from_pmc.py $O r03 > /dev/null && cp profiles/traffic.json $O/traffic.json
# the driver's invocation (few steps) and the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_s2_driver_steps20.json 2> $O/bench_s2_driver.err
python bench.py > $O/r03_bench_s2.json 2> $O/bench_s2.err
python bench.py --ids zipf --no-cpu-baseline > $O/r03_bench_s2_zipf.json 2>/dev/null
python bench.py --workload ragged > $O/r03_bench_ragged.json 2>/dev/null
python bench.py --workload ragged --as-delivered --no-cpu-baseline > $O/r03_bench_ragged_as_delivered.json 2>/dev/null
python bench.py --workload ragged --seg csr --no-cpu-baseline > $O/r03_bench_ragged_csr.json 2>/dev/null
python bench.py --workload dlrm > $O/r03_bench_dlrm.json 2>/dev/null
python bench.py --workload e > $O/r03_bench_ae_model_e.json 2>/dev/null
python bench.py --workload f > $O/r03_bench_ae_model_f.json 2>/dev/null
# kernel traces (single stream: --no-overlap keeps every traced kernel alone on its stream)
for w in s2 ragged "ragged --as-delivered" e; do
  tag=$(echo $w | tr -d ' -')
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$tag/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline --no-pcie --no-overlap > $O/trace_$tag.log 2>&1 )
  python3 scripts/summarize_prof.py $O/t_$tag > $O/r03_${tag}_kernel_trace_stats.txt  # (ragged --as-delivered is renamed r03_ragged_as_delivered_… when copied) 2>&1
done
find $O -name "*.csv" -size +2M -delete
rm -rf "${O:?}"/t_*
ls -la $O
