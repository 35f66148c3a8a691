#!/bin/bash
# A/B of the private-stream request path inside ONE gpurun call (figures from different calls are not compared)
C="--combos 2x2,2x3,3x3,3x4 --nowait 0"
for rep in 1 2; do
echo "== marker event"; FCP_LANE_STOP_EVENT=0 python scripts/r04_private_sweep.py $C 2>&1 | grep -v amdgpu.ids
echo "== stop event on the kernel"; FCP_LANE_STOP_EVENT=1 python scripts/r04_private_sweep.py $C 2>&1 | grep -v amdgpu.ids
done
echo "== stop event, no reader kernel"; FCP_HARNESS_NO_READER=1 python scripts/r04_private_sweep.py $C 2>&1 | grep -v amdgpu.ids
echo "== stop event, 4 hw queues"; GPU_MAX_HW_QUEUES=4 python scripts/r04_private_sweep.py $C 2>&1 | grep -v amdgpu.ids
echo "== stop event, 16 hw queues"; GPU_MAX_HW_QUEUES=16 python scripts/r04_private_sweep.py --combos 2x2,3x3,4x4,5x6 --nowait 0 2>&1 | grep -v amdgpu.ids
