show() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sc=d.get('single_caller_stream') or {}
print(round(d['ms_per_step']*1e3,2), 'lanes', {k:round(v,2) for k,v in sc.get('sweep_us_per_request',{}).items()}, 'workers', {k:round(v,2) for k,v in (d.get('overlapped_serving') or {}).get('sweep_us_per_request',{}).items()})"; }
for pr in normal low; do
 echo "== csr $pr"; FCP_LANE_PRIORITY=$pr python bench.py --workload ragged --seg csr --no-cpu-baseline 2>/dev/null | show
 echo "== csr $pr --no-pcie"; FCP_LANE_PRIORITY=$pr python bench.py --workload ragged --seg csr --no-cpu-baseline --no-pcie 2>/dev/null | show
done
echo "== staged low"; FCP_LANE_PRIORITY=low python bench.py --workload ragged --no-cpu-baseline 2>/dev/null | show
