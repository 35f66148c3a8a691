#!/bin/bash
# GPU box: regenerate the round-2 records under gpurun_out/refresh2/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/refresh2
rm -rf "${O:?}"; mkdir -p "$O"
# PMC passes first (their own runs: --pmc with --kernel-trace only): the bench lines below report
# roofline.traffic from profiles/traffic.json, which must describe the kernels of this build
bash scripts/pmc.sh refresh2 > $O/r02_s2_pmc_fcp_bench.txt 2>&1
bash scripts/pmc_py.sh refresh2_ragged --workload ragged > $O/r02_ragged_pmc.txt 2>&1
bash scripts/pmc_py.sh refresh2_e --workload e > $O/r02_ae_model_e_pmc.txt 2>&1
python3 scripts/traffic_from_pmc.py $O > /dev/null && cp profiles/traffic.json $O/traffic.json
# the driver's invocation (few steps) and the default one
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench_s2_driver_steps20.json 2> $O/bench_s2_driver.err
python bench.py > $O/r02_bench_s2.json 2> $O/bench_s2.err
python bench.py --ids zipf --no-cpu-baseline > $O/r02_bench_s2_zipf.json 2>/dev/null
python bench.py --workload ragged > $O/r02_bench_ragged.json 2>/dev/null
python bench.py --workload ragged --seg csr --no-cpu-baseline > $O/r02_bench_ragged_csr.json 2>/dev/null
python bench.py --workload dlrm > $O/r02_bench_dlrm.json 2>/dev/null
python bench.py --workload e > $O/r02_bench_ae_model_e.json 2>/dev/null
python bench.py --workload f > $O/r02_bench_ae_model_f.json 2>/dev/null
./build/stamps/fcp_bench --steps 200 | tail -24 > $O/stamps_s2.txt
./build/stamps/fcp_bench --columns 980 --vocab 101 --bucketize-every 1 --dim 8 --steps 200 | tail -12 > $O/stamps_elike.txt
./build/ramp_probe 3776 300 > $O/ramp_probe.txt 2>&1
bash scripts/r02_suite.sh product > $O/r02_other_workloads.txt 2>&1
bash scripts/r02_ragged_indices.sh > $O/r02_ragged_segment_encodings.txt 2>&1
python3 scripts/shard_rank_share.py 300 2>&1 | grep -v amdgpu.ids > $O/r02_shard_rank_share.txt
bash scripts/r02_staged.sh > $O/r02_delivered_vs_staged.txt 2>&1
# kernel traces (single stream: --no-overlap keeps every traced kernel alone on its stream)
bash scripts/profile_s2.sh refresh2 > $O/profile_s2.log 2>&1
cp gpurun_out/prof_refresh2/summary.txt $O/r02_s2_kernel_trace_stats.txt 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload ragged --steps 300 --warmup 50 --no-cpu-baseline --no-overlap > $O/trace_r.log 2>&1 )
python3 scripts/summarize_prof.py $O/r > $O/r02_ragged_kernel_trace_stats.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/e/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload e --steps 300 --warmup 50 --no-cpu-baseline --no-overlap > $O/trace_e.log 2>&1 )
python3 scripts/summarize_prof.py $O/e > $O/r02_ae_model_e_kernel_trace_stats.txt 2>&1
find $O -name "*.csv" -size +2M -delete
rm -rf "${O:?}"/r $O/e
ls -la $O
