#!/bin/bash
# Round 2, experiment 2: cost of the front of a launch (kernel arguments in host vs device memory,
# dependent-load hops at launch start), and the sc1-nt store policy re-checked after the hazard fix.
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
us() { sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; }
for k in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$k"
  HIP_FORCE_DEV_KERNARG=$k ./build/ramp_probe 3776 300
  HIP_FORCE_DEV_KERNARG=$k ./build/ramp_probe 2048 300
  for round in 1 2; do
    echo -n "fcp_bench S2 b512 (HIP_FORCE_DEV_KERNARG=$k): "; HIP_FORCE_DEV_KERNARG=$k ./recom_amd/fcp_bench --steps 1000 --verify 0 | tail -1 | us
  done
done
echo "== default env"
./build/ramp_probe 3776 300
for round in 1 2 3; do
  for v in recom_amd build/st4; do
    echo -n "round $round $v b512: "; ./$v/fcp_bench --steps 1000 --verify $((round==1)) | tail -2 | tr '\n' ' ' | sed 's/{.*"dev_us_per_step": \([0-9.]*\).*/\1/'; echo
  done
done
for v in recom_amd build/st4; do
  for t in 2 3; do echo -n "$v $t threads: "; ./$v/fcp_bench --steps 600 --threads $t --verify 0 | tail -1 | sed 's/.*"wall_us_per_step": \([0-9.]*\).*/\1/'; done
done
echo "== stamps build"
./build/stamps/fcp_bench --steps 200 --verify 0 | tail -20
HIP_FORCE_DEV_KERNARG=1 ./build/stamps/fcp_bench --steps 200 --verify 0 | tail -20
