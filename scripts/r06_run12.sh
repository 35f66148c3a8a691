#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT" || exit 1
O="$GRAFT_REPO_ROOT/gpurun_out/r06_run12"; mkdir -p "$O"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sla_bounded or regular_csr" 2>&1 | tail -3
python bench.py --sla-ms 1.0 > "$O/r06_sla_throughput_s2_1ms.json" 2> "$O/sla_s2_1ms.log"; tail -4 "$O/sla_s2_1ms.log"
python bench.py --sla-ms 1.0 --threads 3 > "$O/r06_sla_throughput_s2_1ms_3workers.json" 2> "$O/sla_s2_1ms_3w.log"; tail -3 "$O/sla_s2_1ms_3w.log"
python bench.py --sla-ms 0.1 > "$O/r06_sla_throughput_s2_100us.json" 2> "$O/sla_s2_100us.log"; tail -3 "$O/sla_s2_100us.log"
python3 -c "
import json
for f in ('r06_sla_throughput_s2_1ms','r06_sla_throughput_s2_1ms_3workers','r06_sla_throughput_s2_100us'):
    r=json.loads(open('$O/'+f+'.json').readline()); print(f, r['max_batch_size'], round(r['max_throughput']/1e6,2), 'M inf/s', r['ended_by'], len(r['search']))"
