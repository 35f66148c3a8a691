// fcp_pack.cc — the host loops of the staged Addons>ConcatInputs / request stager (fcp_concat_inputs_ex,
// fcp_stager_stage_ex): int64 -> int32 narrowing and sorted row ids -> row offsets.  Plain C++ (no HIP), in a file of
// its own so that the two loops can be built once per instruction set and picked at run time (function
// multi-versioning): they read 16 bytes per id of a request and are what the CPU op spends its time in.
#include "fcp_env.h"
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
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

// the run-length form for any element size / stride, built per instruction set
FCP_CLONES static int seg_to_csr_any(const void *seg, int elem_size, int64_t stride, int64_t nnz, int64_t rows, int32_t *out) {
  if (elem_size == 8) return seg_to_csr_t(static_cast<const int64_t *>(seg), stride, nnz, rows, out);
  return seg_to_csr_t(static_cast<const int32_t *>(seg), stride, nnz, rows, out);
}

#if defined(__x86_64__)
#include <immintrin.h>
// The same offsets from the BOUNDARIES of the sorted ids, 16 ids per step (AVX-512): offsets[r] = the first position whose
// id is >= r.  One pass: the row words of 16 ids are taken out of the index matrix with two permutes (int64, stride 2: the
// SparseTensor indices [nnz, 2] of BASELINE configs[3]) or loaded as they are (stride 1), clamped to [-1, rows] and
// narrowed; a lane-shifted copy gives every id its predecessor; one compare marks the positions where a new row starts
// (one id in ~5: the only scalar work left), another the descents.  The run-length form above spends ~6 cycles on every
// id in a dependent chain and runs at half the speed of a plain read of the same bytes
// (profiles/r03_pcie_staging_phase_timers.txt: 1108 vs 530 us per RAGGED request on one core); this one follows the read.
enum { kSegI64x2 = 0, kSegI64 = 1, kSegI32 = 2 };

__attribute__((target("avx512f"))) static inline __m512i seg_rows16(int kind, const void *seg, int64_t i, __m512i lo64, __m512i hi64) {
  if (kind == kSegI64x2) {
    const int64_t *q = static_cast<const int64_t *>(seg) + 2 * i;
    const __m512i even = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14);
    const __m512i a = _mm512_permutex2var_epi64(_mm512_loadu_si512(q), even, _mm512_loadu_si512(q + 8));
    const __m512i b = _mm512_permutex2var_epi64(_mm512_loadu_si512(q + 16), even, _mm512_loadu_si512(q + 24));
    return _mm512_inserti64x4(_mm512_castsi256_si512(_mm512_cvtepi64_epi32(_mm512_min_epi64(_mm512_max_epi64(a, lo64), hi64))),
                              _mm512_cvtepi64_epi32(_mm512_min_epi64(_mm512_max_epi64(b, lo64), hi64)), 1);
  }
  if (kind == kSegI64) {
    const int64_t *q = static_cast<const int64_t *>(seg) + i;
    const __m512i a = _mm512_loadu_si512(q), b = _mm512_loadu_si512(q + 8);
    return _mm512_inserti64x4(_mm512_castsi256_si512(_mm512_cvtepi64_epi32(_mm512_min_epi64(_mm512_max_epi64(a, lo64), hi64))),
                              _mm512_cvtepi64_epi32(_mm512_min_epi64(_mm512_max_epi64(b, lo64), hi64)), 1);
  }
  const __m512i x = _mm512_loadu_si512(static_cast<const int32_t *>(seg) + i);
  const __m512i hi32 = _mm512_broadcastd_epi32(_mm512_castsi512_si128(hi64)); // (the low dword of the 64-bit limit: rows <= 2^31 - 1)
  return _mm512_min_epi32(_mm512_max_epi32(x, _mm512_set1_epi32(-1)), hi32);
}

__attribute__((target("avx512f"))) static int seg_to_csr_avx512(int kind, const void *seg, int64_t nnz, int64_t rows, int32_t *out) {
  const int32_t hi = (int32_t)(rows < 0x7fffffff ? rows : 0x7fffffff);
  const __m512i lo64 = _mm512_set1_epi64(-1), hi64 = _mm512_set1_epi64(hi);
  auto row_at = [&](int64_t i) -> int32_t { // the clamped row of one id (the last < 16 of them)
    const int64_t r = kind == kSegI64x2 ? static_cast<const int64_t *>(seg)[2 * i]
                                        : kind == kSegI64 ? static_cast<const int64_t *>(seg)[i] : (int64_t) static_cast<const int32_t *>(seg)[i];
    return (int32_t)(r < -1 ? -1 : (r > hi ? hi : r));
  };
  int64_t next_row = 0; // offsets[0 .. next_row) are written
  int descending = 0;
  alignas(64) int32_t cur_a[16];
  auto boundary = [&](int32_t r, int64_t pos) { // the first id of row r (> every row before it) sits at position pos
    const int64_t top = r < hi ? r : hi;
    for (int64_t rr = next_row; rr <= top; ++rr) out[rr] = (int32_t)pos;
    if (top + 1 > next_row) next_row = top + 1;
  };
  int64_t i = 0;
  __m512i last = _mm512_set1_epi32(-1); // "the row before the first id": ids of rows < 0 (clamped to -1) never start a row
  for (; i + 16 <= nnz; i += 16) {
    const __m512i cur = seg_rows16(kind, seg, i, lo64, hi64);
    const __m512i prv = _mm512_alignr_epi32(cur, last, 15); // lane j = row of id i + j - 1
    last = cur;
    const __mmask16 up = _mm512_cmpgt_epi32_mask(cur, prv);
    descending |= (int)_mm512_cmplt_epi32_mask(cur, prv);
    if (up) {
      _mm512_store_si512(cur_a, cur);
      unsigned m = up;
      do {
        const int j = __builtin_ctz(m);
        m &= m - 1;
        boundary(cur_a[j], i + j);
      } while (m);
    }
  }
  int32_t prev = -1;
  if (i > 0) {
    _mm512_store_si512(cur_a, last);
    prev = cur_a[15];
  }
  for (; i < nnz; ++i) {
    const int32_t r = row_at(i);
    descending |= (int)(r < prev);
    if (r > prev) boundary(r, i);
    prev = r;
  }
  for (int64_t rr = next_row; rr <= rows; ++rr) out[rr] = (int32_t)nnz; // the rows behind the last id's
  return descending != 0;
}
#endif

// returns 1 if the ids were not sorted (the offsets are then meaningless), else 0
extern "C" int fcp_pack_seg_to_csr(const void *seg, int elem_size, int64_t stride, int64_t nnz, int64_t rows, int32_t *out) {
#if defined(__x86_64__)
  static const bool avx512 = __builtin_cpu_supports("avx512f") && !fcp::diag_on("pack_no_avx512"); // (tuning / test aid: FCP_DIAG=pack_no_avx512)
  if (avx512 && rows >= 0 && nnz < 0x7fffffff) {
    if (elem_size == 8 && stride == 2) return seg_to_csr_avx512(kSegI64x2, seg, nnz, rows, out);
    if (elem_size == 8 && stride == 1) return seg_to_csr_avx512(kSegI64, seg, nnz, rows, out);
    if (elem_size == 4 && stride == 1) return seg_to_csr_avx512(kSegI32, seg, nnz, rows, out);
  }
#endif
  return seg_to_csr_any(seg, elem_size, stride, nnz, rows, out);
}
