// ref_device_scan_wrap.hip — the REFERENCE's dim <= 20 pooling templates (SparseSegmentSum / SparseSegmentMean,
// tensorflow_addons/graph_optimizers/cuda_emitter.cc:348-661) compiled for gfx950 from the text of their string literals and
// run on the GPU.  Test infrastructure only (oracle/_ref/libref_device_scan.so, built by oracle/ref_extract.py like
// libref_device.so: literals extracted at build time into a temporary directory, never stored).
//
// READ THIS BEFORE TRUSTING IT AS A PIN.  Those templates sum with `cub::BlockScan<..., BLOCK_SCAN_WARP_SCANS>` of CUB 1.8.0
// (WORKSPACE:5-14), which is NOT in this image.  What IS in the image is hipCUB (/opt/rocm/include/hipcub), AMD's port of
// the CUB interface over rocPRIM: same class, same template parameters, same `TempStorage` / `InclusiveScan(in, out, op)`.
// `namespace cub = hipcub;` below lets the unmodified text compile against it.  Consequences:
//   * everything that is the TEMPLATE's own logic runs as the reference wrote it: head / tail flags from neighbouring
//     row ids, the segmented scan operator, the carry (`last_aggregate`) across 64-id tiles, which rows are written
//     and which stay zero, mean = sum / counter (integer counter, :625), the 8-float slabs of the driver loop;
//   * the ORDER in which the fp32 partial sums of one bag meet is rocPRIM's block scan, not CUB 1.8's: this library does
//     not pin the reference's fp32 association (orc_sparse_segment_reduce_refscan restates CUB's and remains the only
//     statement of it).  Results are therefore compared at the north star's tolerance, exactly only where no
//     association is involved (bags of one or two ids, empty rows).
// The `__global__` wrapper restates the generated code of EmitSparseSegmentReduce (:1444-1622): zero fill of the output
// (:1460-1475, the WithNumSegments case), per 8-float slab the reset of `last_aggregate` (:1548-1563) and the tile loop
// (:1565-1598); SCAN_DIM = 8 (:230); 64 threads (passes/fc_optimize_pass.cc:71).  The only liberty: the LEFT_DIM tail is
// under `if constexpr` so that dims that are multiples of 8 do not instantiate a zero-length scan.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <limits.h>
#include <stdint.h>

namespace cub = hipcub;
constexpr int SCAN_DIM = 8; // cuda_emitter.cc:230
#include "ref_segment_scan.inc" // ScanVecPair .. SparseSegmentSum, ScanVecCntTuple .. SparseSegmentMean

namespace {

constexpr int kBlockThreads = 64;

template <int EmbedDim, bool Mean> struct Pick;
template <int EmbedDim> struct Pick<EmbedDim, false> {
  using Wrapper = SparseSegmentSumTempStorageWrapper<EmbedDim, SCAN_DIM, kBlockThreads, float>;
  template <int ScanDim, typename S>
  static __device__ __forceinline__ void run(S &s, const float *params, int indice, int row_id, int embed_offset, float *out, int n,
                                             bool full_block, bool execute_flag) {
    SparseSegmentSum<EmbedDim, ScanDim, kBlockThreads, float>(s, params, indice, row_id, embed_offset, out, n, full_block, execute_flag);
  }
  template <typename S> static __device__ __forceinline__ void reset_counter(S &) {}
};
template <int EmbedDim> struct Pick<EmbedDim, true> {
  using Wrapper = SparseSegmentMeanTempStorageWrapper<EmbedDim, SCAN_DIM, kBlockThreads, float>;
  template <int ScanDim, typename S>
  static __device__ __forceinline__ void run(S &s, const float *params, int indice, int row_id, int embed_offset, float *out, int n,
                                             bool full_block, bool execute_flag) {
    SparseSegmentMean<EmbedDim, ScanDim, kBlockThreads, float>(s, params, indice, row_id, embed_offset, out, n, full_block, execute_flag);
  }
  template <typename S> static __device__ __forceinline__ void reset_counter(S &s) { s.last_aggregate.counter = 0; }
};

// get_process_str (:1545-1600) for one slab of ScanDim floats starting at embed_offset
template <int EmbedDim, int ScanDim, bool Mean, typename S>
__device__ __forceinline__ void process_slab(S &s, const float *params, const long long *ids, const long long *seg, int seg_stride,
                                             int num_input, int embed_offset, float *out) {
  for (int i = threadIdx.x; i < ScanDim; i += kBlockThreads) s.last_aggregate.scan_vec[i] = 0;
  if (threadIdx.x == 0) {
    s.last_aggregate.scan_key = 0;
    Pick<EmbedDim, Mean>::reset_counter(s);
  }
  __syncthreads();
  int last_row_id = 0;
  for (int offset = 0; offset < num_input; offset += kBlockThreads) {
    const bool full_block = (offset + kBlockThreads) <= num_input;
    const int item_idx = offset + threadIdx.x;
    const bool execute_flag = item_idx < num_input;
    if (threadIdx.x + 1 == kBlockThreads) {
      s.row_ids[0] = last_row_id;
      s.row_ids[kBlockThreads + 1] = (item_idx + 1) < num_input ? (int)seg[(long long)(item_idx + 1) * seg_stride] : INT_MAX;
    }
    last_row_id = execute_flag ? (int)seg[(long long)item_idx * seg_stride] : INT_MAX;
    Pick<EmbedDim, Mean>::template run<ScanDim>(s, params, execute_flag ? (int)ids[item_idx] : 0, last_row_id, embed_offset, out,
                                                num_input - offset, full_block, execute_flag);
    __syncthreads();
  }
}

template <int EmbedDim, bool Mean>
__global__ void __launch_bounds__(kBlockThreads)
    ref_scan_kernel(const float *params, const long long *ids, const long long *seg, int seg_stride, int num_input, int num_output,
                    float *out) {
  __shared__ typename Pick<EmbedDim, Mean>::Wrapper s_ssr[1];
  for (int i = threadIdx.x; i < num_output; i += kBlockThreads) out[i] = 0;
  __syncthreads(); // ConstructSubgraphCode :1969-1971
  constexpr int LEFT_DIM = EmbedDim % SCAN_DIM;
  for (int embed_offset = 0; embed_offset < EmbedDim - LEFT_DIM; embed_offset += SCAN_DIM)
    process_slab<EmbedDim, SCAN_DIM, Mean>(s_ssr[0].normal, params, ids, seg, seg_stride, num_input, embed_offset, out);
  if constexpr (LEFT_DIM != 0)
    process_slab<EmbedDim, LEFT_DIM, Mean>(s_ssr[0].left, params, ids, seg, seg_stride, num_input, EmbedDim - LEFT_DIM, out);
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
#define REF_FOR_DIMS(X) X(1) X(2) X(3) X(4) X(8) X(12) X(16) X(20)

} // namespace

extern "C" {

// SparseSegment{Sum,Mean}WithNumSegments as the dim <= 20 emitter generates it: out[num_segments, dim]; seg = sorted segment
// ids read with element stride seg_stride; ids / segment ids must be in range (the reference does not check)
int ref_dev_scan_segment_reduce(const float *table, int64_t vocab, int dim, const int64_t *ids, const int64_t *seg, int seg_stride,
                                int n, int num_segments, int mean, float *out) {
  DeviceBuf d_t(table, (size_t)vocab * dim * 4), d_i(ids, (size_t)n * 8), d_s(seg, (size_t)n * seg_stride * 8),
      d_o(nullptr, (size_t)num_segments * dim * 4);
  REF_TRY(d_t.err);
  REF_TRY(d_i.err);
  REF_TRY(d_s.err);
  REF_TRY(d_o.err);
  switch (dim) {
#define X(D)                                                                                                              \
  case D:                                                                                                                 \
    if (mean)                                                                                                             \
      hipLaunchKernelGGL((ref_scan_kernel<D, true>), dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(), d_i.as<long long>(), \
                         d_s.as<long long>(), seg_stride, n, num_segments * dim, d_o.as<float>());                         \
    else                                                                                                                  \
      hipLaunchKernelGGL((ref_scan_kernel<D, false>), dim3(1), dim3(kBlockThreads), 0, 0, d_t.as<float>(), d_i.as<long long>(), \
                         d_s.as<long long>(), seg_stride, n, num_segments * dim, d_o.as<float>());                         \
    break;
    REF_FOR_DIMS(X)
#undef X
  default: return -1;
  }
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(out, d_o.p, (size_t)num_segments * dim * 4, hipMemcpyDeviceToHost));
  return 0;
}

} // extern "C"
