#!/bin/bash
# CPU container (no GPU needed): the HOST side of libfcp_hip.so under AddressSanitizer + UBSan (device code is compiled
# without instrumentation: GPU sanitizers are unavailable on this pool), then the host-only tests against that build:
# descriptor validation, shape evaluation, plan files, placement gate, external slots, gloo sharding.  Also the C oracle
# under gcc's ASan + UBSan through its own tests.
set -e
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/fcp_asan}
mkdir -p $OUT
CLANG=/opt/rocm/lib/llvm/bin/clang++
SAN="-std=c++17 -fPIC -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
( cd recom_amd/csrc && $CLANG $SAN -c fcp_graph.cc -o $OUT/fcp_graph.o && $CLANG $SAN -c fcp_pack.cc -o $OUT/fcp_pack.o &&
  for f in fcp_kernels fcp_plan fcp_process fcp_lanes fcp_concat fcp_stager fcp_shard; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -Xarch_device -fno-sanitize=address,undefined -c $f.hip -o $OUT/$f.o || exit 1
  done &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -fsanitize=address,undefined -shared $OUT/fcp_kernels.o $OUT/fcp_plan.o $OUT/fcp_process.o $OUT/fcp_lanes.o $OUT/fcp_concat.o $OUT/fcp_stager.o $OUT/fcp_shard.o \
    $OUT/fcp_graph.o $OUT/fcp_pack.o -o $OUT/libfcp_hip.so -ldl )
RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 FCP_LIB_DIR=$OUT python -m pytest tests/test_host.py tests/test_graph_plan.py \
    tests/test_shard_gloo.py -x -q -k "not occupancy and not sanitizer and not plain_c and not tf_shim"
cp oracle/libfcp_oracle.so $OUT/libfcp_oracle.so.keep 2>/dev/null || true
gcc -O1 -g -fPIC -fno-fast-math -ffp-contract=off -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared oracle/fcp_oracle.c -o oracle/libfcp_oracle.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle.py -x -q || rc=$?
make -C oracle -B libfcp_oracle.so > /dev/null   # back to the ordinary build
exit ${rc:-0}
