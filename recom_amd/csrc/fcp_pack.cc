// fcp_pack.cc — the host loops of the staged Addons>ConcatInputs / request stager (fcp_concat_inputs_ex,
// fcp_stager_stage_ex): int64 -> int32 narrowing and sorted row ids -> row offsets.  Plain C++ (no HIP), in a file of
// its own so that the two loops can be built once per instruction set and picked at run time (function
// multi-versioning): they read 16 bytes per id of a request and are what the CPU op spends its time in.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__x86_64__) && defined(__has_attribute)
#if __has_attribute(target_clones)
#define FCP_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#endif
#endif
#ifndef FCP_CLONES
#define FCP_CLONES
#endif

// int64 -> int32; values that do not fit are not valid ids / rows anyway: -1 (reads as zeros on the device)
FCP_CLONES static void narrow_plain(const int64_t *src, int32_t *dst, int64_t n) {
  for (int64_t k = 0; k < n; ++k) {
    const int64_t v = src[k];
    dst[k] = (uint64_t)v <= 0x7fffffffull ? (int32_t)v : -1;
  }
}

// (Streaming stores for the destination — it is only read by the DMA engine next — were measured and dropped: the same
// at 8-16 pack threads, 15-40 % slower at 1-4, profiles/r04_pcie_staging.txt.)
extern "C" void fcp_pack_narrow_i64(const int64_t *src, int32_t *dst, int64_t n) { narrow_plain(src, dst, n); }

// Sorted segment / row ids (element i at index i * stride, int32 or int64) -> CSR offsets[0..rows]: offsets[r] = number
// of ids below r — what the device pre-pass (fcp_segment_offsets_kernel) and ComputeSegmentOffsets
// (cuda_emitter.cc:768-818) produce.  Run lengths, then their prefix sum.  The pass over the ids has no branch and
// no load from `out`: every id stores the length of its row's run so far (the last store of a run is the row's count),
// so neither a row boundary (one per ~5 ids, unpredictable) nor a store-to-load dependency on a counter stalls it;
// rows beyond the last one go to a dummy, all rows below 0 count as "before row 0".
template <typename T> static inline int seg_to_csr_t(const T *p, int64_t stride, int64_t nnz, int64_t rows, int32_t *out) {
  memset(out, 0, sizeof(int32_t) * (size_t)(rows + 1));
  // blocks of ids: first the strided row words are gathered into a small dense int32 buffer, clamped to [-1, rows]
  // (a loop the compiler vectorises: this is the pass that touches the request's memory), then the run lengths
  // are taken from that buffer
  constexpr int kBlock = 1024;
  int32_t tmp[kBlock];
  const int32_t hi = (int32_t)(rows < 0x7fffffff ? rows : 0x7fffffff);
  int32_t dummy = 0, cur = INT32_MIN, run = 0, descending = 0;
  for (int64_t i0 = 0; i0 < nnz; i0 += kBlock) {
    const int n = (int)(nnz - i0 < kBlock ? nnz - i0 : kBlock);
    const T *q = p + i0 * stride;
    if (stride == 2) {
      for (int i = 0; i < n; ++i) {
        const int64_t r = (int64_t)q[2 * i];
        tmp[i] = (int32_t)(r < -1 ? -1 : (r > hi ? hi : r));
      }
    } else {
      for (int i = 0; i < n; ++i) {
        const int64_t r = (int64_t)q[(int64_t)i * stride];
        tmp[i] = (int32_t)(r < -1 ? -1 : (r > hi ? hi : r));
      }
    }
    for (int i = 0; i < n; ++i) {
      const int32_t r = tmp[i];
      run = (run & -(int32_t)(r == cur)) + 1; // arithmetic, not a branch: one row in ~5 ends here, unpredictably
      descending |= (int32_t)(r < cur);       // the ids must be sorted (TF: "segment ids are not increasing")
      cur = r;
      // counts[r + 1] = ids of row r; counts[0] = ids before row 0; rows >= `rows` go to a dummy
      const uintptr_t in = (uintptr_t)0 - (uintptr_t)(r < hi);
      int32_t *dst = reinterpret_cast<int32_t *>((reinterpret_cast<uintptr_t>(out + (r + 1)) & in) | (reinterpret_cast<uintptr_t>(&dummy) & ~in));
      *dst = run;
    }
  }
  int32_t acc = 0;
  for (int64_t r = 0; r <= rows; ++r) {
    acc += out[r];
    out[r] = acc;
  }
  return descending;
}

// returns 1 if the ids were not sorted (the offsets are then meaningless), else 0
extern "C" FCP_CLONES int fcp_pack_seg_to_csr(const void *seg, int elem_size, int64_t stride, int64_t nnz, int64_t rows, int32_t *out) {
  if (elem_size == 8) return seg_to_csr_t(static_cast<const int64_t *>(seg), stride, nnz, rows, out);
  return seg_to_csr_t(static_cast<const int32_t *>(seg), stride, nnz, rows, out);
}
