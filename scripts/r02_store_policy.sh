#!/bin/bash
# Round 2, experiment 1: what does the output-store cache policy cost at the kernel boundary?
# (MI355X_MICROARCH.md "boundary": + B / 6 TB/s when the predecessor leaves B bytes dirty in L2.)
# Interleaved rounds of the product build (nt stores) against plain / sc1 / sc0 sc1 / sc1 nt / sc0 sc1 nt
# builds, S2 at batch 512 and 2048, single stream; plus FCP_DYN_UPLOAD=kernel (descriptors in ordinary
# instead of fine-grained device memory).
: "${GRAFT_REPO_ROOT:?run on a gpurun box (or export GRAFT_REPO_ROOT)}"
cd "$GRAFT_REPO_ROOT" || exit 1
us() { sed 's/.*"dev_us_per_step": \([0-9.]*\).*/\1/'; }
for round in 1 2; do
  for v in recom_amd build/stplain build/st2 build/st3 build/st4 build/st5; do
    echo -n "round $round $v b512: "; ./$v/fcp_bench --steps 1000 --verify $((round==1)) | tail -1 | us
  done
  echo -n "round $round recom_amd b512 FCP_DYN_UPLOAD=kernel: "; FCP_DYN_UPLOAD=kernel ./recom_amd/fcp_bench --steps 1000 --verify 0 | tail -1 | us
done
for v in recom_amd build/st2 build/st3 build/st4; do
  echo -n "$v b2048: "; ./$v/fcp_bench --steps 300 --batch 2048 --verify 0 | tail -1 | us
  echo -n "$v b512 3 threads: "; ./$v/fcp_bench --steps 600 --threads 3 --verify 0 | tail -1
done
