#!/bin/bash
# GPU box: regenerate every round-1 record under gpurun_out/refresh/ (copied into profiles/ afterwards).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/refresh
mkdir -p $O
python bench.py > $O/r01_bench_s2.json 2> $O/bench_s2.err
python bench.py --ids zipf --no-cpu-baseline > $O/r01_bench_s2_zipf.json 2>/dev/null
python bench.py --workload ragged > $O/r01_bench_ragged.json 2>/dev/null
python bench.py --workload dlrm > $O/r01_bench_dlrm.json 2>/dev/null
python bench.py --workload e > $O/r01_bench_ae_model_e.json 2>/dev/null
python bench.py --workload f > $O/r01_bench_ae_model_f.json 2>/dev/null
python scripts/ae_split.py e 1000 > $O/r01_ae_split.txt 2>/dev/null
python scripts/ae_split.py f 1000 >> $O/r01_ae_split.txt 2>/dev/null
python scripts/seg_encoding.py > $O/r01_ragged_segment_encodings.txt 2>/dev/null
./build/stamps/fcp_bench --steps 200 | tail -24 > $O/stamps_s2.txt
./build/stamps/fcp_bench --columns 980 --vocab 101 --bucketize-every 1 --dim 8 --steps 200 | tail -12 > $O/stamps_elike.txt
bash scripts/profile_s2.sh refresh > $O/profile_s2.log 2>&1
cp gpurun_out/prof_refresh/summary.txt $O/r01_s2_kernel_trace_stats.txt 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/e/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload e --steps 300 --warmup 50 --no-cpu-baseline --no-overlap > $O/trace_e.log 2>&1 )
python3 scripts/summarize_prof.py $O/e > $O/r01_ae_model_e_kernel_trace_stats.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r/trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload ragged --steps 300 --warmup 50 --no-cpu-baseline --no-overlap > $O/trace_r.log 2>&1 )
python3 scripts/summarize_prof.py $O/r > $O/r01_ragged_kernel_trace_stats.txt 2>&1
bash scripts/pmc.sh refresh > $O/r01_s2_pmc_fcp_bench.txt 2>&1
find $O -name "*.csv" -size +2M -delete
ls -la $O
