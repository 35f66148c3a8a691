// ref_device_wrap.hip — the REFERENCE's own CUB-free device templates, compiled for gfx950 from the text of their string
// literals and run on the GPU as a parity pin.  Test infrastructure only (tests/ load oracle/_ref/libref_device.so).
//
// oracle/ref_extract.py reads, at build time, from /root/reference/tensorflow_addons/graph_optimizers/cuda_emitter.cc
//   :250-293   GatherRowsToGlbMem        (form 1)
//   :296-345   GatherScatterRows         (form 3, a9)
//   :664-765   AlignedVector             (ported by the reference from TensorFlow 2.6)
//   :768-962   experiment::ComputeSegmentOffsets / SparseSegmentReduce (form 2, dim > 20, a8)
// un-escapes the literals into a temporary directory, compiles THIS file against them and deletes them again: the text
// is never part of the repository and does not travel to the GPU box — only the built library does.
//
// What is ours here, and what it stands for: the reference has no kernels as source either — `FusedKnl`
// (:2097-2134) is generated text whose per-column body is [pre_glb_area] + [loop_body] of the column's SubgraphCode.  The
// `__global__` wrappers below restate exactly those generated fragments, one 64-thread block per column as FusedKnl
// runs them (block_threads = 64, passes/fc_optimize_pass.cc:71):
//   ref_gather_rows_kernel          loop_body of EmitGatherRows                    :1305-1327
//   ref_gather_scatter_rows_kernel  zero fill :1351-1359 + loop_body of EmitGatherScatterRows :1415-1439
//   ref_ssr_kernel                  zero fill :1659-1672 + the call EmitSparseSegmentReduceExperiment emits :1734-1753
//                                   with its hard-coded parameters :1630-1635 (vector 4, vblock 8 x 8, 1 segment id per
//                                   thread, 1 segment per block, unroll 4)
//   ref_segment_offsets_kernel      experiment::ComputeSegmentOffsets alone (the first half of that call)
// Index operands are read as EmitInputInline would inline them: ids `(int) ids[idx]` (Cast, :1788), segment ids
// `(int) seg[idx * seg_stride]` (StridedSlice [:, 0:1] of an [nnz, k] matrix, :1836-1873).
// This is FUNCTION-LEVEL pinning: the templates are the reference's, the dozen lines of driver around them are restated.
// SparseSegmentSum / SparseSegmentMean for dim <= 20 (:402-661) need cub::BlockScan (CUB 1.8) and stay restated only.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define int32 int32_t
#define int64 int64_t
#define float32 float
#include "ref_gather.inc"          // GatherRowsToGlbMem, GatherScatterRows
#include "ref_experiment.inc"      // AlignedVector, namespace experiment

namespace {

constexpr int kBlockThreads = 64; // CudaEmitter::block_threads as the pass constructs it (passes/fc_optimize_pass.cc:71)

template <int EmbedDim>
__global__ void __launch_bounds__(kBlockThreads)
    ref_gather_rows_kernel(const float *params, const long long *ids, int num_input, float *out) {
  __shared__ int s_indices[kBlockThreads];
  for (int i = 0; i < num_input; i += kBlockThreads) {
    const bool full_block = (i + kBlockThreads) <= num_input;
    const int item_idx = i + threadIdx.x;
    const bool execute_flag = item_idx < num_input;
    GatherRowsToGlbMem<EmbedDim>(s_indices, params, execute_flag ? (int)ids[item_idx] : 0, out + i * EmbedDim,
                                 (num_input - i) * EmbedDim, full_block, execute_flag);
    __syncthreads();
  }
}

template <int EmbedDim>
__global__ void __launch_bounds__(kBlockThreads)
    ref_gather_scatter_rows_kernel(const float *params, const long long *ids, const long long *rows, int row_stride,
                                   int num_input, int num_output, float *out) {
  __shared__ int s_indices[kBlockThreads];
  __shared__ int s_row_ids[kBlockThreads];
  for (int i = threadIdx.x; i < num_output; i += kBlockThreads) out[i] = 0;
  __syncthreads(); // (ConstructSubgraphCode puts a barrier between pre_glb_area and the loop body, :1969-1971)
  for (int i = 0; i < num_input; i += kBlockThreads) {
    const bool full_block = (i + kBlockThreads) <= num_input;
    const int item_idx = i + threadIdx.x;
    const bool execute_flag = item_idx < num_input;
    GatherScatterRows<EmbedDim>(s_indices, s_row_ids, params, execute_flag ? (int)ids[item_idx] : 0,
                                execute_flag ? (int)rows[(long long)item_idx * row_stride] : 0, out,
                                (num_input - i) * EmbedDim, full_block, execute_flag);
    __syncthreads();
  }
}

template <int EmbedDim, bool Mean>
__global__ void __launch_bounds__(kBlockThreads)
    ref_ssr_kernel(const float *params, const long long *ids, const long long *seg, int seg_stride, int input_elenum,
                   int num_segments, int *segment_offsets, float *out) {
  __shared__ experiment::SparseSegmentReduceTempStorage<float, 4, 8, 8> s_ssr[1];
  const int output_elenum = num_segments * EmbedDim;
  for (int i = threadIdx.x; i < output_elenum; i += kBlockThreads) out[i] = 0;
  __syncthreads(); // :1969-1971
  experiment::SparseSegmentReduce<EmbedDim, 4, 1, 1, 8, 8, 4, Mean>(
      s_ssr[0], [&](int idx) { return (int)seg[(long long)idx * seg_stride]; }, [&](int idx) { return (int)ids[idx]; }, params,
      segment_offsets, out, input_elenum, num_segments);
}

__global__ void __launch_bounds__(kBlockThreads)
    ref_segment_offsets_kernel(const long long *seg, int seg_stride, int num_inputs, int num_segments, int *segment_offsets) {
  experiment::ComputeSegmentOffsets<1, kBlockThreads>([&](int idx) { return (int)seg[(long long)idx * seg_stride]; }, segment_offsets,
                                                      num_inputs, num_segments);
}

struct DeviceBuf {
  void *p = nullptr;
  hipError_t err = hipSuccess;
  DeviceBuf(const void *host, size_t bytes) {
    err = hipMalloc(&p, bytes ? bytes : 16);
    if (err == hipSuccess && host && bytes) err = hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
  }
  ~DeviceBuf() {
    if (p) (void)hipFree(p);
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

#define REF_TRY(e)                          \
  do {                                      \
    const hipError_t e_ = (e);              \
    if (e_ != hipSuccess) return (int)e_;   \
  } while (0)

// the embedding widths the tests use (EmbedDim is a template argument of the reference's code)
#define REF_FOR_DIMS(X) X(1) X(2) X(3) X(4) X(8) X(12) X(16) X(20) X(24) X(32) X(48) X(64) X(128)
#define REF_FOR_DIMS4(X) X(4) X(8) X(12) X(16) X(20) X(24) X(32) X(48) X(64) X(128)

} // namespace

extern "C" {

// out[n, dim] = table[ids, :]; ids must lie in [0, vocab): the reference does not check (it reads out of bounds)
int ref_dev_gather_rows(const float *table, int64_t vocab, int dim, const int64_t *ids, int n, float *out) {
  DeviceBuf d_t(table, (size_t)vocab * dim * 4), d_i(ids, (size_t)n * 8), d_o(nullptr, (size_t)n * dim * 4);
  REF_TRY(d_t.err);
  REF_TRY(d_i.err);
  REF_TRY(d_o.err);
  switch (dim) {
#define X(D)                                                                                                             \
  case D:                                                                                                                \
    hipLaunchKernelGGL(ref_gather_rows_kernel<D>, dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(), d_i.as<long long>(), n, \
                       d_o.as<float>());                                                                                 \
    break;
    REF_FOR_DIMS(X)
#undef X
  default: return -1;
  }
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(out, d_o.p, (size_t)n * dim * 4, hipMemcpyDeviceToHost));
  return 0;
}

// out[num_rows, dim] = 0; out[rows[i * row_stride], :] = table[ids[i], :] (rows must lie in [0, num_rows): unchecked there)
int ref_dev_gather_scatter_rows(const float *table, int64_t vocab, int dim, const int64_t *ids, const int64_t *rows,
                                int row_stride, int n, int num_rows, float *out) {
  DeviceBuf d_t(table, (size_t)vocab * dim * 4), d_i(ids, (size_t)n * 8), d_r(rows, (size_t)n * row_stride * 8),
      d_o(nullptr, (size_t)num_rows * dim * 4);
  REF_TRY(d_t.err);
  REF_TRY(d_i.err);
  REF_TRY(d_r.err);
  REF_TRY(d_o.err);
  switch (dim) {
#define X(D)                                                                                                             \
  case D:                                                                                                                \
    hipLaunchKernelGGL(ref_gather_scatter_rows_kernel<D>, dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(),            \
                       d_i.as<long long>(), d_r.as<long long>(), row_stride, n, num_rows * dim, d_o.as<float>());         \
    break;
    REF_FOR_DIMS(X)
#undef X
  default: return -1;
  }
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(out, d_o.p, (size_t)num_rows * dim * 4, hipMemcpyDeviceToHost));
  return 0;
}

// experiment::SparseSegmentReduce as the emitter calls it: offsets_out[num_segments + 1] (ComputeSegmentOffsets) and
// out[num_segments, dim]; seg = sorted segment ids read with element stride seg_stride
int ref_dev_sparse_segment_reduce(const float *table, int64_t vocab, int dim, const int64_t *ids, const int64_t *seg,
                                  int seg_stride, int n, int num_segments, int mean, float *out, int32_t *offsets_out) {
  DeviceBuf d_t(table, (size_t)vocab * dim * 4), d_i(ids, (size_t)n * 8), d_s(seg, (size_t)n * seg_stride * 8),
      d_off(nullptr, (size_t)(num_segments + 1) * 4), d_o(nullptr, (size_t)num_segments * dim * 4);
  REF_TRY(d_t.err);
  REF_TRY(d_i.err);
  REF_TRY(d_s.err);
  REF_TRY(d_off.err);
  REF_TRY(d_o.err);
  REF_TRY(hipMemset(d_off.p, 0xff, (size_t)(num_segments + 1) * 4)); // every entry must be WRITTEN by ComputeSegmentOffsets
  switch (dim) {
#define X(D)                                                                                                             \
  case D:                                                                                                                \
    if (mean)                                                                                                            \
      hipLaunchKernelGGL((ref_ssr_kernel<D, true>), dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(), d_i.as<long long>(), \
                         d_s.as<long long>(), seg_stride, n, num_segments, d_off.as<int>(), d_o.as<float>());             \
    else                                                                                                                 \
      hipLaunchKernelGGL((ref_ssr_kernel<D, false>), dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(), d_i.as<long long>(), \
                         d_s.as<long long>(), seg_stride, n, num_segments, d_off.as<int>(), d_o.as<float>());             \
    break;
    REF_FOR_DIMS4(X)
#undef X
  default: return -1;
  }
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(out, d_o.p, (size_t)num_segments * dim * 4, hipMemcpyDeviceToHost));
  if (offsets_out) REF_TRY(hipMemcpy(offsets_out, d_off.p, (size_t)(num_segments + 1) * 4, hipMemcpyDeviceToHost));
  return 0;
}

int ref_dev_segment_offsets(const int64_t *seg, int seg_stride, int n, int num_segments, int32_t *offsets_out) {
  DeviceBuf d_s(seg, (size_t)n * seg_stride * 8), d_off(nullptr, (size_t)(num_segments + 1) * 4);
  REF_TRY(d_s.err);
  REF_TRY(d_off.err);
  REF_TRY(hipMemset(d_off.p, 0xff, (size_t)(num_segments + 1) * 4));
  hipLaunchKernelGGL(ref_segment_offsets_kernel, dim3(1), dim3(kBlockThreads), 0, 0, d_s.as<long long>(), seg_stride, n, num_segments,
                     d_off.as<int>());
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(offsets_out, d_off.p, (size_t)(num_segments + 1) * 4, hipMemcpyDeviceToHost));
  return 0;
}

int ref_dev_block_threads(void) { return kBlockThreads; }

} // extern "C"
