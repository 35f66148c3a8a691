// fcp_kernels.hip — hand-written gfx950 (CDNA4, MI355X) kernels of the fused
// feature-column path.  HBM-bound gather / pool / concat: no MFMA on purpose.
//
// Replaces, for every model, what the reference generates as CUDA text:
//   FusedKnl + struct FCi            graph_optimizers/cuda_emitter.cc:1976-2134
//   Bucketize                        :233-247
//   GatherRowsToGlbMem               :250-293
//   GatherScatterRows                :296-345
//   SparseSegmentSum / Mean          :402-501 / :564-661
//   experiment::ComputeSegmentOffsets / SparseSegmentReduce   :768-962
//   BatchColReduction                :1216-1241
//   ConcatOutputsKnl / ScatterBlock  custom_ops/concat_outputs/concat_outputs_op_gpu.cu.cc:85-131
//
// Work decomposition (the MI355X-first part).  The reference launches ONE
// 64-thread block per column (`FusedKnl<<<num_fc, 64>>>`, :2234) which walks
// the whole batch serially — at most #columns waves on the chip.  Here the unit
// of work is a *slot* of the concatenated output row: the output matrix
// [rows, sum(dim)] of a concat group is cut into V-float slots (V = 4 when all
// dims are multiples of 4: one 16-byte access per lane); a wave owns 64
// consecutive slots (1 KiB of one output row — spanning as many neighbouring
// columns as fit) for R consecutive batch rows, a 256-thread block owns that
// span for 4*R rows.  Consequences:
//   * every store instruction of a wave writes 1 KiB contiguous bytes of the
//     final concat layout — the concat pass and its intermediate arena vanish;
//   * a table row of dim floats is read by dim/4 adjacent lanes as one
//     contiguous run (coalesced into whole 64/128-byte requests);
//   * per block, the column records, the CSR row ranges and the ids of the
//     span are fetched ONCE (coalesced, one request per cache line) and staged in
//     LDS; the lanes then read them as LDS broadcasts;
//   * 1000 columns x batch 512 give 3840 blocks / 15360 waves instead of 1000
//     waves, enough to keep >16 MB of loads in flight (HBM latency x bandwidth);
//   * blocks that share a span (the same columns / tables) are given the same
//     `blockIdx % 8`, i.e. the same XCD and L2 under round-robin dispatch, so
//     skewed (Zipf) ids and small bucketize tables are served from one L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/fcp_hip.h"
#include <hip/hip_ext.h>
#include "fcp_internal.h"
#include "fcp_env.h"

namespace {

// Pointers that travel through LDS records (or are computed from them) lose their
// address space: hipcc then emits FLAT loads, which probe the LDS aperture as well
// and complete out of order.  Every such access is cast back to the global address
// space so that it becomes a plain global_load / global_store.
#define FCP_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ const FCP_GLOBAL T *as_global(const T *p) {
  return (const FCP_GLOBAL T *)(p);
}
template <typename T> __device__ __forceinline__ FCP_GLOBAL T *as_global(T *p) { return (FCP_GLOBAL T *)(p); }
// Plan data and the request's descriptors are read-only for the lifetime of a launch: the constant
// address space lets the compiler use scalar loads for wave-uniform addresses (slot map, span list).
#define FCP_CONST __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const FCP_CONST T *as_const(const T *p) { return (const FCP_CONST T *)(p); }

template <int V> struct VecType;
template <> struct VecType<4> { typedef float __attribute__((ext_vector_type(4))) T; };
template <> struct VecType<2> { typedef float __attribute__((ext_vector_type(2))) T; };
template <> struct VecType<1> { typedef float T; };

template <int V> struct alignas(4 * V) VF { float v[V]; };

template <int V> __device__ __forceinline__ VF<V> vzero() {
  VF<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = 0.0f;
  return r;
}

// Output rows are written once and consumed by a later kernel: non-temporal stores
// (measured on S2: 34.3 -> 30.5 us per request).  Table rows are read with the DEFAULT
// cache policy: streaming them (-DFCP_NT_LOADS) changes nothing on S2 (1M-row tables,
// uniform ids) but costs the reference's models E / F 4.5 us per request — their ~1000
// bucketize / hash tables of ~100 rows are re-read by every row and belong in L2.
// (r6) A per-column choice — non-temporal reads for tables far beyond an XCD's L2, default policy for the small hot ones —
// was built and measured too, because a bare gather probe reads 11 % faster with `nt` at every row size (53.7 against 48.5
// G rows/s): inside the fused kernels it moved nothing (S2 27.39 against 27.35 us, model F 12.9 / 12.9, RAGGED -0.5 us in
// one encoding, +0 in the other: profiles/r06_streamed_table_reads_negative.txt) and was taken out again.
// -DFCP_NO_NT restores the default policy for stores as well (tuning builds).
// Write-through form (`sc1 nt`: the line leaves the XCD's L2 at once instead of at the kernel boundary) for
// outputs larger than the L2s can hold — S2's 61 MB: 28.4 vs 29.0 us per request; outputs that FIT the
// caches (DLRM 3.5 MB, models E / F 16 MB) lose with it (DLRM 4.7 -> 5.9 us, F 14.7 -> 17.7 us): the host
// picks per launch (FcpLaunch::store_through, profiles/r02_store_policy.txt).  Inline asm: the compiler has
// no builtin for the sc1 bit; the s_nop covers the hazard "VMEM store of more than 64 bits followed by a
// VALU write of its data registers", which the hazard recogniser cannot see inside asm.
__device__ __forceinline__ void st_through(FCP_GLOBAL VecType<4>::T *p, VecType<4>::T t) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_through(FCP_GLOBAL VecType<2>::T *p, VecType<2>::T t) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_through(FCP_GLOBAL VecType<1>::T *p, VecType<1>::T t) {
  asm volatile("global_store_dword %0, %1, off sc1 nt" ::"v"(p), "v"(t) : "memory");
}

// (r6) Third policy, PLAIN stores (default cache policy), for an arena that is the one the plan's previous request (or the
// one before it) wrote: TF's allocate_output(2) hands a serving loop the block it just freed
// (feature_column_process_op_gpu.cu.cc:107-111), so an output line is rewritten one request later — 61 MB of output + 86 MB
// of table lines in between stay within the 256-MiB Infinity Cache, and rewriting a resident line is cheaper than streaming
// it past the caches: S2 27.2 us against 27.9 (nt) / 28.0 (sc1 nt) with one arena, 28.0 / 28.2 / 28.2 with two; with three
// or more arenas plain stores LOSE (31.1 against 28.1): profiles/r06_arena_reuse_store_policy.txt.  The host decides per
// request (FcpLaunch::store_through bit 2, fill_launch).
#define FCP_ST_THROUGH 1
#define FCP_ST_PLAIN 4
// The non-temporal form in inline asm as well: written as `if (plain) *p = t; else __builtin_nontemporal_store(t, p);` the
// compiler MERGED the two stores into one and dropped the nontemporal hint with it (the first build of the three-policy
// st_out had exactly two store instructions per row: `sc1 nt` and plain — every "nt" request wrote with plain stores,
// RAGGED with six arenas 27.5 -> 29.3 us, profiles/r06_arena_reuse_kernel_traces.txt).
__device__ __forceinline__ void st_nt(FCP_GLOBAL VecType<4>::T *p, VecType<4>::T t) {
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_nt(FCP_GLOBAL VecType<2>::T *p, VecType<2>::T t) {
  asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_nt(FCP_GLOBAL VecType<1>::T *p, VecType<1>::T t) {
  asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(t) : "memory");
}

template <int V> __device__ __forceinline__ void st_out(float *p, const VF<V> &v, int policy = 0) {
  typedef typename VecType<V>::T T;
  T t;
  __builtin_memcpy(&t, &v, sizeof(T));
  if (policy & FCP_ST_THROUGH) {
    st_through(as_global(reinterpret_cast<T *>(p)), t);
    return;
  }
#if !defined(FCP_NO_NT)
  if (!(policy & FCP_ST_PLAIN)) {
    st_nt(as_global(reinterpret_cast<T *>(p)), t);
    return;
  }
#endif
  *as_global(reinterpret_cast<T *>(p)) = t;
}

// Row `off` of a table of `spr` slots (of V floats) per row whose lane-specific base is `tb`: one v_mad_u64_u32
// (row x slots per row, 64-bit: a table may be of any size, rows < 2^32 - 3) + one global_load.
template <int V> __device__ __forceinline__ VF<V> ld_slot32(const float *tb, uint32_t off);
template <int V> __device__ __forceinline__ VF<V> ld_slot(const float *tb, uint32_t off, uint32_t spr) {
  typedef typename VecType<V>::T T;
#if defined(FCP_ABLATE) && FCP_ABLATE == 4 // timing-only build 4: no table reads at all (ragged kernel too)
  VF<V> z = vzero<V>();
  z.v[0] = (float)off;
  return z;
#endif
  const FCP_GLOBAL T *g = as_global(reinterpret_cast<const T *>(tb)) + (uint64_t)off * spr;
#if defined(FCP_NT_LOADS) // tuning build: stream table rows too (see the comment above st_out)
  T t = __builtin_nontemporal_load(g);
#else
  T t = *g;
#endif
  VF<V> r;
  __builtin_memcpy(&r, &t, sizeof(T));
  return r;
}
// The same for a PRE-SCALED slot offset (row x slots per row < 2^32 - 3, known for the whole plan: FcpLaunch::store_through
// bit 1 clear): one v_lshl_add_u64 + one global_load — what the dense kernel uses whenever every table allows it
// (S2: 0.15-0.25 us per request against the 64-bit multiply-add, profiles/r03_row_index_ab.txt).
template <int V> __device__ __forceinline__ VF<V> ld_slot32(const float *tb, uint32_t off) {
  typedef typename VecType<V>::T T;
#if defined(FCP_ABLATE) && FCP_ABLATE == 4 // timing-only build 4: no table reads at all (ragged kernel too)
  VF<V> z = vzero<V>();
  z.v[0] = (float)off;
  return z;
#endif
  const FCP_GLOBAL T *g = as_global(reinterpret_cast<const T *>(tb)) + off;
#if defined(FCP_NT_LOADS) // tuning build: stream table rows too (see the comment above st_out)
  T t = __builtin_nontemporal_load(g);
#else
  T t = *g;
#endif
  VF<V> r;
  __builtin_memcpy(&r, &t, sizeof(T));
  return r;
}

// Blob tensors are only guaranteed 4-byte aligned (ConcatInputs packs bytes
// back to back, concat_inputs_ops.cc:52-60): payloads are read dword by dword,
// 8-byte ids as two dwords.
template <int V> __device__ __forceinline__ VF<V> ld_blob_f32(const char *p) {
  VF<V> r;
  const FCP_GLOBAL float *q = as_global(reinterpret_cast<const float *>(p));
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = q[i];
  return r;
}

__device__ __forceinline__ int64_t ld_i64_a4(const char *p) {
  const FCP_GLOBAL uint32_t *q = as_global(reinterpret_cast<const uint32_t *>(p));
  const uint32_t lo = q[0], hi = q[1];
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// cuda_emitter.cc:233-247 — r+1 = number of boundaries <= value.
template <typename P> __device__ __forceinline__ int bucketize(P b, int n, float value) {
#if defined(FCP_ABLATE) && FCP_ABLATE == 5 // timing-only build 5: no boundary search
  return (int)value & 63;
#endif
  int l = 0, r = n - 1;
  while (l <= r) {
    const int mid = (l + r) >> 1;
    if (value < b[mid]) {
      r = mid - 1;
    } else {
      l = mid + 1;
    }
  }
  return r + 1;
}

// The same count for (nearly) evenly spaced boundaries — the reference's bucketized columns use
// 0, 5, ..., 495 (examples/python/microbenchmark.py:46): guess the bucket from the spacing, read the two
// boundaries that bracket it (independent reads, one round trip) and accept the guess only if
// b[g-1] <= value < b[g]; anything else (rounding at a boundary, NaN, uneven spacing) runs the search.
// Exact for every input by construction; ~2 reads instead of log2(n) dependent ones.
template <typename P> __device__ __forceinline__ int bucketize_fast(P b, int n, float b0, float inv, float value) {
  float t = (value - b0) * inv;
  t = fminf(fmaxf(t, -1.0f), (float)n); // NaN -> -1
  const int g = min(max((int)floorf(t) + 1, 0), n);
  const float below = b[max(g - 1, 0)], above = b[min(g, n - 1)];
  const bool ok = (g == 0 || !(value < below)) && (g == n || value < above);
  if (ok) return g;
  return bucketize(b, n, value);
}

// Boundaries that are REPRODUCIBLE as fma(i, step, b0) (checked bit for bit when the plan is created):
// the same guess-and-verify, and the fallback search, on computed boundaries — the array is never read.
// (Reading it made every block of a launch hit the same few cache lines at once: 2.3 us of queueing on
// one L2 channel at the head of an S2 launch.)
__device__ __forceinline__ int bucketize_arith(int n, float b0, float inv, float step, float value) {
  float t = (value - b0) * inv;
  t = fminf(fmaxf(t, -1.0f), (float)n); // NaN -> -1
  const int g = min(max((int)floorf(t) + 1, 0), n);
  const float below = __builtin_fmaf((float)(g - 1), step, b0), above = __builtin_fmaf((float)g, step, b0);
  if ((g == 0 || !(value < below)) && (g == n || value < above)) return g;
  int l = 0, r = n - 1;
  while (l <= r) {
    const int mid = (l + r) >> 1;
    if (value < __builtin_fmaf((float)mid, step, b0)) {
      r = mid - 1;
    } else {
      l = mid + 1;
    }
  }
  return r + 1;
}

constexpr uint32_t kNoRow = 0xFFFFFFFFu;    // "this id contributes nothing" (another rank's row, or past the end of a bag)
constexpr uint32_t kFiltered = 0xFFFFFFFEu; // dropped by the column's id filter: contributes nothing AND does not count in a mean
constexpr uint32_t kBadRow = 0xFFFFFFFDu;   // an id outside [0, vocab): reads as zeros (and can be told from kNoRow when it is counted late)
__device__ __forceinline__ bool is_row(uint32_t off) { return off < kBadRow; }

// One column of the span, staged in LDS by the block: the static record VERBATIM (its six 16-byte words go
// from the load straight to LDS: nothing is re-packed, few registers live) plus what the request's dynamic
// record turns into.
struct alignas(16) LdsCol : FcpColStatic { // 64 + 32 = 96 bytes
  const char *ids;            // id / value stream of this request
  const int32_t *csr;         // CSR offsets of this request (blob or arena scratch), or — L.seg_search — the segment ids
  int64_t out_base;           // byte offset in the arena of element (0,0)
  int32_t out_stride;
  union {
    int32_t nnz;              // lookup forms: number of ids
    int32_t inner;            // BatchColReduction: rows reduced per output row
  };
};
static_assert(sizeof(LdsCol) == 96 && sizeof(FcpColStatic) == 64, "column records: 64 static + 32 derived bytes");

// The scalars of the argument block a body uses, fetched up front in ONE batch of scalar loads and
// pinned there (the empty asm keeps the compiler from sinking each load next to its first use, which
// made a string of separate scalar-cache round trips at the head of every block).
struct Hot {
  const FCP_CONST uint32_t *slot_map, *span_list;
  const FCP_CONST FcpColStatic *cols;
  const FCP_CONST FcpColDyn *dyn;
  const char *blob;
  char *arena;
  unsigned long long *bad_ids;
  const float *zeros;
  int64_t csr_arena_off;
  int32_t n_groups, rank, world, seg_search, store_through;
  FcpGroupLaunch g0;
};

__device__ __forceinline__ Hot load_hot(const FcpLaunch &L) {
  Hot h;
  h.slot_map = as_const(L.slot_map);
  h.span_list = as_const(L.span_list);
  h.cols = as_const(L.cols);
  h.dyn = as_const(L.dyn);
  h.blob = L.blob;
  h.arena = L.arena;
  h.bad_ids = L.bad_ids;
  h.zeros = L.zeros;
  h.csr_arena_off = L.csr_arena_off;
  h.n_groups = L.n_groups;
  h.rank = L.shard_rank;
  h.world = L.shard_world;
  h.seg_search = L.seg_search;
  h.store_through = L.store_through;
  h.g0 = L.groups[0];
  asm volatile("" : "+s"(h.slot_map), "+s"(h.span_list), "+s"(h.cols), "+s"(h.dyn), "+s"(h.blob), "+s"(h.arena),
               "+s"(h.bad_ids), "+s"(h.csr_arena_off), "+s"(h.zeros));
  asm volatile("" : "+s"(h.n_groups), "+s"(h.rank), "+s"(h.world), "+s"(h.seg_search), "+s"(h.store_through), "+s"(h.g0.rows), "+s"(h.g0.nslots),
               "+s"(h.g0.nsp8), "+s"(h.g0.block_begin), "+s"(h.g0.slot_map_off), "+s"(h.g0.span_list_off), "+s"(h.g0.nlist));
  return h;
}

// A column record as unconditional 16-byte loads (field-by-field access let the compiler wait
// for `flags` before it asked for the rest: two or three dependent round trips in phase 0).
template <typename T> __device__ __forceinline__ T ld_rec(const FCP_CONST T *p) {
  static_assert(sizeof(T) % 16 == 0, "column records are whole 16-byte words");
  typedef uint32_t __attribute__((ext_vector_type(4))) U4;
  const FCP_CONST U4 *g = reinterpret_cast<const FCP_CONST U4 *>(p);
  U4 w[sizeof(T) / 16];
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 16; ++i) w[i] = g[i];
  T r;
  __builtin_memcpy(&r, w, sizeof(T));
  return r;
}

// Phase 0 of both bodies for one column: static record -> LDS word by word, then the derived part.
// (For a PASSTHROUGH column the "table" is its payload in the blob.)
__device__ __forceinline__ void stage_col(const Hot &L, LdsCol *dst, const FCP_CONST FcpColStatic *gs,
                                          const FCP_CONST FcpColDyn *gd) {
  typedef uint32_t __attribute__((ext_vector_type(4))) U4;
  const FCP_CONST U4 *ws = reinterpret_cast<const FCP_CONST U4 *>(gs);
  const FCP_CONST U4 *wd = reinterpret_cast<const FCP_CONST U4 *>(gd);
  U4 s0 = ws[0], s1 = ws[1], s2 = ws[2], s3 = ws[3];
  const U4 d0 = wd[0], d1 = wd[1], d2 = wd[2];
  FcpColDyn cd;
  {
    U4 w[3] = {d0, d1, d2};
    __builtin_memcpy(&cd, w, sizeof(cd));
  }
  const uint32_t flags = s2.x; // word 2: flags, n_boundaries, seg_stride, bnd_b0 (offset 32)
  const char *ids = L.blob + cd.ids_off;
  if (FCP_F_FORM(flags) == FCP_FORM_PASSTHROUGH) { // word 0: table, boundaries
    const uint64_t t = (uint64_t)reinterpret_cast<uintptr_t>(ids);
    s0.x = (uint32_t)t;
    s0.y = (uint32_t)(t >> 32);
  }
  U4 *out = reinterpret_cast<U4 *>(dst);
  out[0] = s0;
  out[1] = s1;
  out[2] = s2;
  out[3] = s3;
  const unsigned segkind = FCP_F_SEGKIND(flags);
  dst->ids = ids;
  dst->csr = (segkind == FCP_SEG_CSR_I32 || (segkind != FCP_SEG_NONE && L.seg_search))
                 ? reinterpret_cast<const int32_t *>(L.blob + cd.seg_off)
             : segkind != FCP_SEG_NONE ? reinterpret_cast<const int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base
                                       : nullptr;
  dst->out_base = cd.out_base;
  dst->out_stride = cd.out_stride;
  dst->nnz = FCP_F_FORM(flags) == FCP_FORM_BATCH_COL_REDUCTION ? cd.inner : cd.nnz;
}
static_assert(offsetof(FcpColStatic, flags) == 32 && offsetof(FcpColStatic, table) == 0, "stage_col reads the record by word");

// the interval test of Addons>SelectValue / Addons>GatherIndiceValue /
// Addons>GatherValueGenIndice (select_value_ops.cc:33-56 and siblings), fused: closed intervals,
// `lo <= id && id <= hi` (the reference's `||` accepts everything).  Returns the id the
// lookup sees, or kDroppedId: the filter removed it.  Out of line and fed from the per-column side table
// (FcpLaunch::xforms) on purpose: columns without a transform — nearly all — pay one compare, no registers
// and no record bytes for it (inlined with the intervals in the column record it cost S2 2 us of 29: 76 VGPRs).
constexpr int64_t kDroppedId = INT64_MIN;

// ---- Fingerprint64 of a short byte string (FarmHash farmhashna::Hash64, lengths 1..32; TensorFlow's
// StringToHashBucketFast, core/kernels/string_to_hash_bucket_fast_op.h) -----------------------------------
// The string is the decimal form of an int64 (at most 20 bytes), kept in three little-endian 64-bit words
// held in registers (no arrays: nothing may end up in scratch memory).
struct Str24 {
  uint64_t w0, w1, w2;
};
__device__ __forceinline__ uint64_t fetch64(const Str24 &s, int o) { // unaligned little-endian read at byte o (o <= 15)
  const int i = o >> 3, sh = (o & 7) * 8;
  const uint64_t lo = (i == 0 ? s.w0 : s.w1) >> sh;
  return sh ? lo | ((i == 0 ? s.w1 : s.w2) << (64 - sh)) : lo;
}
__device__ __forceinline__ uint64_t rot64(uint64_t v, int sh) { return sh ? (v >> sh) | (v << (64 - sh)) : v; }
__device__ __forceinline__ uint64_t hash_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= a >> 47;
  uint64_t b = (v ^ a) * mul;
  b ^= b >> 47;
  return b * mul;
}
__device__ uint64_t fingerprint64_decimal(int64_t value) {
  constexpr uint64_t k0 = 0xc3a5c85c97cb3127ull, k1 = 0xb492b66fbe98f273ull, k2 = 0x9ae16a3b2f90404full;
  // AsString of an integer: decimal digits, most significant first, '-' for negatives, no padding
  const uint64_t mag = value < 0 ? 0ull - (uint64_t)value : (uint64_t)value;
  int n = 1;
  for (uint64_t p = 10; n < 20 && mag >= p; p *= 10) ++n; // 10^19 < 2^64: p never overflows before n reaches 20
  const int neg = value < 0 ? 1 : 0;
  Str24 s = {neg ? (uint64_t)'-' : 0ull, 0ull, 0ull};
  uint64_t m = mag;
  for (int k = 0; k < n; ++k) { // least significant digit first, written at its final position
    const int pos = neg + n - 1 - k;
    const uint64_t v = (uint64_t)('0' + (int)(m % 10)) << ((pos & 7) * 8);
    m /= 10;
    if (pos < 8) s.w0 |= v;
    else if (pos < 16) s.w1 |= v;
    else s.w2 |= v;
  }
  n += neg;
  const uint64_t len = (uint64_t)n;
  if (n <= 16) {
    if (n >= 8) {
      const uint64_t mul = k2 + len * 2, a = fetch64(s, 0) + k2, b = fetch64(s, n - 8);
      return hash_len16(rot64(b, 37) * mul + a, (rot64(a, 25) + b) * mul, mul);
    }
    if (n >= 4) {
      const uint64_t mul = k2 + len * 2, a = (uint32_t)s.w0;
      return hash_len16(len + (a << 3), (uint32_t)(s.w0 >> ((n - 4) * 8)), mul);
    }
    const uint8_t a = (uint8_t)s.w0, b = (uint8_t)(s.w0 >> ((n >> 1) * 8)), c = (uint8_t)(s.w0 >> ((n - 1) * 8));
    const uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)n + ((uint32_t)c << 2);
    uint64_t h = (uint64_t)y * k2 ^ (uint64_t)z * k0;
    h ^= h >> 47;
    return h * k2;
  }
  const uint64_t mul = k2 + len * 2, a = fetch64(s, 0) * k1, b = fetch64(s, 8), c = fetch64(s, n - 8) * mul,
                 e = fetch64(s, n - 16) * k2;
  return hash_len16(rot64(a + b, 43) + rot64(c, 30) + e, a + rot64(b + k2, 18) + c, mul);
}

__device__ __attribute__((noinline)) int64_t apply_xform(uint32_t xform, const FcpXform *xf, int64_t id) {
  const FCP_GLOBAL FcpXform *x = as_global(xf);
  if (xform & FCP_XFORM_HASH_BIT) id = (int64_t)(fingerprint64_decimal(id) % (uint64_t)x->hash_buckets);
  const unsigned mode = xform & 3u;
  if (mode == FCP_XFORM_NONE) return id;
  bool in = id >= x->lo0 && id <= x->hi0;
  const int n = (int)((xform & ~FCP_XFORM_HASH_BIT) >> 2);
  for (int i = 1; i < n && !in; ++i) {
    const FCP_GLOBAL int64_t *e = as_global(x->extra) + 2 * (i - 1);
    in = id >= e[0] && id <= e[1];
  }
  if (in) return id;
  return mode == FCP_XFORM_FILTER ? kDroppedId : x->sub;
}

// The index expression the reference inlines per column (EmitInputInline,
// cuda_emitter.cc:1769-1949: raw int32 / int64 ids, or Bucketize(float value)),
// the range check and the row shard, folded into ONE number per id: the (local) row
// of the table (< 2^32 - 3 rows per table or shard is checked when the plan is created;
// the byte offset is formed in 64 bits where the row is read), or kNoRow.  Ids outside [0, vocab) read as zeros (the
// reference reads out of bounds, TF-GPU GatherV2 returns zeros); under row
// sharding an id owned by another rank contributes nothing here.
template <int V, bool SHARDED>
__device__ __forceinline__ uint32_t slot_offset_from_raw(const LdsCol &c, const FcpXform *xf, uint32_t lo, uint32_t hi,
                                                         const float *lds_bnd, int rank, int world, bool &bad) {
  const unsigned idsrc = FCP_F_IDSRC(c.flags);
  int64_t id;
  if (idsrc == FCP_IDS_F32_BUCKETIZE) {
    const float x = __uint_as_float(lo);
    if (c.bnd_step != 0.0f) // reproducible boundaries: no reads at all
      id = bucketize_arith(c.n_boundaries, c.bnd_b0, c.bnd_inv, c.bnd_step, x);
    else if (lds_bnd)       // staged in LDS by the block; evenly spaced ones start from the guess
      id = c.bnd_inv != 0.0f ? bucketize_fast(lds_bnd, c.n_boundaries, c.bnd_b0, c.bnd_inv, x)
                             : bucketize(lds_bnd, c.n_boundaries, x);
    else
      id = c.bnd_inv != 0.0f ? bucketize_fast(as_global(c.boundaries), c.n_boundaries, c.bnd_b0, c.bnd_inv, x)
                             : bucketize(as_global(c.boundaries), c.n_boundaries, x);
  } else {
    id = idsrc == FCP_IDS_I64 ? (int64_t)(((uint64_t)hi << 32) | lo) : (int64_t)(int32_t)lo;
  }
  bad = false;
  if (c.xform) { // rare
    id = apply_xform(c.xform, xf, id);
    if (id == kDroppedId) return kFiltered;
  }
  bad = (uint64_t)id >= (uint64_t)c.vocab;
  if (bad) return kBadRow;
  if (SHARDED) {
    const int64_t q = id < 0x7fffffffLL ? (int64_t)((uint32_t)id / (uint32_t)world) : id / world;
    if (id - q * world != rank) return kNoRow;
    id = q;
  }
  return (uint32_t)id;
}

template <int V, bool SHARDED>
__device__ __forceinline__ uint32_t fetch_slot_offset(const LdsCol &c, const FcpXform *xf, int64_t pos, const float *lds_bnd,
                                                      int rank, int world, bool &bad) {
  const bool is64 = FCP_F_IDSRC(c.flags) == FCP_IDS_I64;
  // branch-free fetch: one code path for every id source
  const char *a = c.ids + (is64 ? 8 : 4) * pos;
  const uint32_t lo = *as_global(reinterpret_cast<const uint32_t *>(a));
  const uint32_t hi = *as_global(reinterpret_cast<const uint32_t *>(a + (is64 ? 4 : 0)));
  return slot_offset_from_raw<V, SHARDED>(c, xf, lo, hi, lds_bnd, rank, world, bad);
}

// Common block header: which group / span / row tile this block owns.
struct BlockPos {
  int rows, nslots, q0, row_blk, ncols;
  uint32_t first_col;
  const FCP_CONST uint32_t *map;
};

template <int RB> __device__ __forceinline__ bool locate_block(const FcpLaunch &L, const Hot &H, int bid, BlockPos &B) {
  // Every scalar a block of a one-group plan needs sits at a FIXED offset of the argument block: the
  // loads are issued together and waited for once.  (Indexing groups[g] with a searched g made a chain
  // of five dependent scalar loads, each a scalar-cache miss at launch start: 1.9 us before the first
  // column record was requested, profiles/r01_s2_block_timeline_stamps.txt "desc".)
  FcpGroupLaunch G = H.g0;
  if (H.n_groups > 1) {
    for (int k = 1; k < H.n_groups; ++k)
      if (bid >= L.groups[k].block_begin) G = L.groups[k];
  }
  B.rows = G.rows;
  B.nslots = G.nslots;
  const int nsp8 = G.nsp8;
  B.map = H.slot_map + G.slot_map_off;
  bid -= G.block_begin;
  // XCD-aware mapping: blocks with equal (bid & 7) share an XCD under the
  // round-robin dispatch; give them the same spans (same columns / tables).
  int idx, tile; // idx: position in the list of spans this launch covers
  if (nsp8 > 0) {
    const int xcd = bid & 7, j8 = bid >> 3;
    idx = (j8 % nsp8) * 8 + xcd;
    tile = j8 / nsp8;
  } else { // fewer than 8 spans: no padding to 8 (nsp8 = -nlist)
    idx = bid % (-nsp8);
    tile = bid / (-nsp8);
  }
  if (idx >= G.nlist) return false; // uniform: whole block leaves
  const int lo = G.span_list_off;
  const int span = lo >= 0 ? (int)H.span_list[lo + idx] : idx;
  B.q0 = span * FCP_WAVE;
  B.row_blk = tile * RB;
  if (B.q0 >= B.nslots || B.row_blk >= B.rows) return false;
  B.first_col = B.map[B.q0];
  B.ncols = (int)(B.map[min(B.q0 + FCP_WAVE - 1, B.nslots - 1)] - B.first_col) + 1;
  return true;
}

// ---------------------------------------------------------------------------
// Dense kernel: every column of the plan is GATHER or PASSTHROUGH (exactly one
// source row per output row) — BASELINE.json's S2 and DLRM shapes.
//
// A block owns one span (64 slots = 1 KiB of the output row) for RB = 4*R rows.
//   phase 0  the span's column records (static + dynamic, contiguous because
//            the device arrays are kept in concat order) are copied to LDS,
//            one thread per column;
//   phase 0b bucketize boundaries -> LDS (the reference stages them per block
//            too, cuda_emitter.cc:1818-1825); wave 0 assigns LDS offsets with a
//            shuffle prefix sum, columns that do not fit keep searching in L2;
//   phase 1  the block's (column, row) id pairs are fetched with one thread
//            per pair — consecutive threads take consecutive rows of one
//            column, so every id cache line is requested exactly once — turned
//            into table slot offsets and parked in LDS;
//   phase 2  every lane reads its column record and its R slot offsets from
//            LDS (broadcast reads), issues its R 16-byte table reads back to
//            back, then its R stores: 1 KiB contiguous per wave instruction,
//            straight into the concat layout.
// Without the LDS staging every lane fetched its own copy of the id and of the
// 96-byte column record: ~80 vector-memory instructions per wave and — measured
// with rocprofv3 — about half of the kernel time queueing on the same in-flight
// cache lines (profiles/r01_s2_pmc_before_lds_staging.txt).
// ---------------------------------------------------------------------------
template <int R> struct DenseLds {
  static constexpr int RB = FCP_WAVES_PER_BLOCK * R; // rows per block
  static constexpr int IDS = RB + 1;                 // padded row of the offset tile (LDS banks)
  static constexpr int BND = 1024;                   // floats of bucketize boundaries staged per block
  LdsCol col[FCP_WAVE];
  uint32_t off[FCP_WAVE * IDS];
  float bnd[BND];
};

template <int V, int R, bool SHARDED>
__device__ __forceinline__ void dense_body(const FcpLaunch &L, int bid, char *smem) {
  constexpr int RB = DenseLds<R>::RB, IDS = DenseLds<R>::IDS, BND = DenseLds<R>::BND;
  DenseLds<R> &S = *reinterpret_cast<DenseLds<R> *>(smem);
  LdsCol *s_col = S.col;
  uint32_t *s_off = S.off;
  float *s_bnd = S.bnd;

  BlockPos B;
#if defined(FCP_STAMPS) // diagnostic build: where does a block spend its time (never shipped)
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
  const Hot H = load_hot(L);
#if defined(FCP_ABLATE) && FCP_ABLATE == 6
  // timing-only build 6 (S2's blob layout only; results are NOT the request's): what would a block-tile-major id layout buy?
  // Every thread requests the 8-byte id words of its pairs at t = 0 — one contiguous 1152-byte chunk of valid int64 ids per
  // block, at an address that needs nothing but the block index — in parallel with the slot map and the column records,
  // instead of after them (the upper bound of "ids stored block-tile-major by ConcatInputs").
  constexpr int PT4 = (FCP_WAVE * RB + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS;
  uint32_t early_lo[PT4], early_hi[PT4];
  {
    const int k = bid % 3000;
    const char *chunk = H.blob + (size_t)(k % 100) * 38912 + 2048 + (size_t)(k / 100) * 1152;
#pragma unroll
    for (int h = 0; h < PT4; ++h) {
      const int p = threadIdx.x + h * FCP_BLOCK_THREADS;
      early_lo[h] = early_hi[h] = 0;
      if (p < 160) { // a span of S2 holds 8-10 columns x 16 rows: the pairs a block really has
        early_lo[h] = *as_global(reinterpret_cast<const uint32_t *>(chunk + 8 * (p % 144)));
        early_hi[h] = *as_global(reinterpret_cast<const uint32_t *>(chunk + 8 * (p % 144) + 4));
      }
    }
  }
#endif
  if (!locate_block<RB>(L, H, bid, B)) return;
  const int tid = threadIdx.x;
  const int lane = tid & (FCP_WAVE - 1);
  const int wave = tid >> 6;
  const int q = B.q0 + lane;
  const uint32_t my_col = B.map[min(q, B.nslots - 1)];
  const int world = H.world, rank = H.rank;
  const bool wide = (H.store_through & 2) != 0; // some table has 2^32 - 3 slots or more: rows are parked, not slot offsets

  // ---- phase 0 ----------------------------------------------------------------------
  if (tid < B.ncols) stage_col(H, &s_col[tid], H.cols + B.first_col + tid, H.dyn + B.first_col + tid);
  __syncthreads();
#if defined(FCP_STAMPS)
  const unsigned long long t_desc = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- phase 1a: raw id words of this thread's (column, row) pairs ------------------------
  // issued before the boundary staging so that the two memory round trips overlap.  The column facts of
  // all the thread's pairs are read from LDS in one batch (unconditional reads, one wait), then the id
  // loads are issued back to back: interleaving "LDS read, wait, load" per pair put ~0.3 us of LDS
  // round trips in front of the last id load.
  constexpr int PT = (FCP_WAVE * RB + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS; // pairs per thread, at most
  uint32_t raw_lo[PT], raw_hi[PT], pflags[PT];
  const char *pids[PT];
  const int npairs = B.ncols * RB;
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    const int p = tid + h * FCP_BLOCK_THREADS;
    const int j = min(p / RB, B.ncols - 1);
    pflags[h] = s_col[j].flags;
    pids[h] = s_col[j].ids;
  }
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    const int p = tid + h * FCP_BLOCK_THREADS;
    const int b = B.row_blk + p % RB;
    const unsigned form = FCP_F_FORM(pflags[h]);
    raw_lo[h] = raw_hi[h] = 0;
#if defined(FCP_ABLATE) && FCP_ABLATE == 6
    if (p < npairs && b < B.rows && form == FCP_FORM_GATHER) {
      raw_lo[h] = early_lo[h];
      raw_hi[h] = early_hi[h];
    }
#else
    if (p < npairs && b < B.rows && form == FCP_FORM_GATHER) {
      const bool is64 = FCP_F_IDSRC(pflags[h]) == FCP_IDS_I64;
      const char *a = pids[h] + (is64 ? 8 : 4) * (int64_t)b;
      raw_lo[h] = *as_global(reinterpret_cast<const uint32_t *>(a));
      raw_hi[h] = *as_global(reinterpret_cast<const uint32_t *>(a + (is64 ? 4 : 0)));
    }
#endif
  }

  // ---- phase 0b: bucketize boundaries -> LDS (skipped when the span has none) --------------
  // Every wave derives the same staging plan from the column records (lane l looks at column
  // l): which columns bucketize, which of them lead a run of neighbours sharing one boundary
  // array (deduplicated at plan creation), and where each run's copy goes (wave-shuffle prefix
  // sum).  No block-wide vote, no per-column LDS round trips: the leaders' (pointer, length,
  // offset) triples travel by lane broadcast, all 256 threads copy, one barrier publishes.
  // (arrays reproducible as fma(i, step, b0) are not staged: their boundaries are computed, never read)
  const bool my_bkt = lane < B.ncols && FCP_F_IDSRC(s_col[lane].flags) == FCP_IDS_F32_BUCKETIZE &&
                      FCP_F_FORM(s_col[lane].flags) == FCP_FORM_GATHER && s_col[lane].bnd_step == 0.0f;
  const unsigned long long any_bkt = __ballot(my_bkt);
#if defined(FCP_STAMPS)
  const unsigned long long t_or = __builtin_amdgcn_s_memrealtime();
#endif
  if (any_bkt) {
    const float *mine = my_bkt ? s_col[lane].boundaries : nullptr;
    const float *prev = reinterpret_cast<const float *>(__shfl_up((unsigned long long)mine, 1));
    const bool leader = my_bkt && (lane == 0 || prev != mine);
    const int nb_all = my_bkt ? s_col[lane].n_boundaries : 0;
    const int nb = leader ? nb_all : 0;
    int incl = nb;
#pragma unroll
    for (int d = 1; d < FCP_WAVE; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    // followers take their leader's slice; arrays that do not fit stay in global memory (-1)
    const int boff = (nb_all > 0 && incl <= BND && incl >= nb_all) ? incl - nb_all : -1;
    if (wave == 0 && boff >= 0) s_col[lane].bnd_off = boff;
    unsigned long long todo = __ballot(leader && boff >= 0);
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const FCP_GLOBAL float *src = as_global(reinterpret_cast<const float *>(__shfl((unsigned long long)mine, j)));
      const int n = __shfl(nb_all, j), off = __shfl(boff, j);
      for (int i = tid; i < n; i += FCP_BLOCK_THREADS) s_bnd[off + i] = src[i];
    }
    __syncthreads();
  }

#if defined(FCP_STAMPS)
  const unsigned long long t_stage = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // diagnostic: when have the raw ids landed?
  const unsigned long long t_raw = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- phase 1b: raw ids -> table slot offsets in LDS -----------------------------------------
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    const int p = tid + h * FCP_BLOCK_THREADS;
    if (p >= npairs) break;
    const int j = p / RB, r = p % RB;
    const int b = B.row_blk + r;
    uint32_t off = kNoRow;
    if (b < B.rows) {
      const LdsCol &c = s_col[j];
      const unsigned form = FCP_F_FORM(pflags[h]);
      if (form == FCP_FORM_PASSTHROUGH) {
        // a tensor of the blob copied into its concat slot; table-free columns
        // belong to shard rank 0
        if (rank == 0) off = wide ? (uint32_t)b : (uint32_t)b * (uint32_t)(c.dim / V);
      } else if (form == FCP_FORM_GATHER) {
        bool bad;
        off = slot_offset_from_raw<V, SHARDED>(c, L.xforms + B.first_col + j, raw_lo[h], raw_hi[h],
                                               c.bnd_off >= 0 ? s_bnd + c.bnd_off : nullptr,
                                               rank, world, bad);
        if (!wide && is_row(off)) off *= (uint32_t)(c.dim / V); // every table of the plan has < 2^32 - 3 slots: pre-scaled
        // a column that straddles two spans is staged by two blocks: the one holding its first slot counts
        if (bad && H.bad_ids && c.out_off >= B.q0 * V) atomicAdd(H.bad_ids, 1ull);
      } // FCP_FORM_EXTERNAL: nothing to fetch, nothing to write
    }
    s_off[j * IDS + r] = off;
  }
  __syncthreads();
#if defined(FCP_STAMPS)
  const unsigned long long t_ids = __builtin_amdgcn_s_memrealtime();
#endif
  if (q >= B.nslots) return;

  // ---- phase 2: R table reads in flight per lane, then R coalesced stores -------------------
  const int j = (int)(my_col - B.first_col);
  const int e = q * V - s_col[j].out_off;
  const float *tb = s_col[j].table + e;
  const int64_t ostride = s_col[j].out_stride;
  if (FCP_F_FORM(s_col[j].flags) == FCP_FORM_EXTERNAL) return; // somebody else's slot (ConcatOutputs host input)
  float *outp = reinterpret_cast<float *>(H.arena + s_col[j].out_base) + e;
  const int r0 = wave * R;
  const uint32_t spr = (uint32_t)(s_col[j].dim / V); // slots per table row
  uint32_t off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = s_off[j * IDS + r0 + r];
#if defined(FCP_ABLATE) && FCP_ABLATE == 3 // timing-only build: sequential instead of random rows
#pragma unroll
  for (int r = 0; r < R; ++r)
    off[r] = (uint32_t)(((int64_t)(B.row_blk + r0 + r) * 131 + my_col * 977) % s_col[j].vocab);
#endif
  VF<V> v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    v[r] = vzero<V>();
#if !(defined(FCP_ABLATE) && FCP_ABLATE == 1) // timing-only build 1: no table reads
    if (is_row(off[r])) v[r] = wide ? ld_slot<V>(tb, off[r], spr) : ld_slot32<V>(tb, off[r]);
#else
    v[r].v[0] = (float)off[r];
#endif
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = B.row_blk + r0 + r;
#if defined(FCP_ABLATE) && FCP_ABLATE == 2 // timing-only build 2: no output stores
    asm volatile("" ::"v"(v[r].v[0]), "v"(v[r].v[V - 1]));
    if (b < B.rows && v[r].v[0] == 1234.5f) st_out<V>(outp + (int64_t)b * ostride, v[r]);
#else
    if (b < B.rows) st_out<V>(outp + (int64_t)b * ostride, v[r], H.store_through);
#endif
  }
#if defined(FCP_STAMPS)
  if (L.stamps && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's row reads have landed, its stores are issued and acknowledged
    unsigned long long *o = L.stamps + 8ull * bid;
    o[4] = t_or;
    o[5] = t_stage;
    o[6] = t_raw;
    o[0] = t_begin;
    o[1] = t_desc;
    o[2] = t_ids;
    o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int V, int R, bool SHARDED>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_dense_kernel(const FcpLaunch L) {
  __shared__ __attribute__((aligned(16))) char smem[sizeof(DenseLds<R>)];
  dense_body<V, R, SHARDED>(L, blockIdx.x, smem);
}

// Table reads a lane of the ragged kernel keeps in flight while it walks a bag (tuning builds:
// 12 / 16 need 71 / 87 VGPRs and lose more to occupancy than they gain, profiles/HISTORY.md, round 1).
#if !defined(FCP_WALK)
#define FCP_WALK 8
#endif
#if !defined(FCP_WALK_FIRST) // widest first batch of a bag walk (10: 64 VGPRs, the most that keeps 8 waves per SIMD without scratch)
#define FCP_WALK_FIRST 10
#endif
#if !defined(FCP_WALK_LONG) // the same in the rounds after the first (rows whose bags exceed the wave's tile)
#define FCP_WALK_LONG 6 // (8 would need 66 VGPRs in the loop around the rounds)
#endif

// ---------------------------------------------------------------------------
// Ragged kernel: any mix of column forms (dynamic shapes: multi-hot bags of
// variable length, scatter columns, passthrough, Sum(axis=1)).
//
// Same block shape as the dense kernel, one output row per wave (RB = 4).
//   phase 0  (block) the span's column records -> LDS; the only barrier
//            (plans whose few segment-id columns are searched in the blocks add a
//            second one around the search);
//   phase 1  (wave) a wave owns ONE output row, so it stages its own row's bags:
//            lane j < ncols reads the CSR range [lo, lo+cnt) of (column j, row) —
//            the row-offset buffer — a wave prefix sum (shuffles) assigns the bag a
//            slice of the wave's LDS offset tile and the lane marks the slice in the
//            wave's owner table; then one lane per *id* (owner table -> column,
//            position) fetches it and stores the table slot offset (Bucketize, range
//            check, row shard: once per id instead of once per lane, all ids of the
//            row in one memory round trip);
//   phase 2  (wave) lane q walks its column's bag — range and slice come from the
//            owner lane by cross-lane reads: 8 (then 4) slot offsets -> as many
//            independent 16-byte table reads in flight -> adds in id order
//            (sequential fp32 order: deterministic, the oracle's; the additions TF-CPU
//            performs for bags of up to 9 ids — from 10 on TF sums every further 8 rows among
//            themselves first, orc_sparse_segment_reduce_tfcpu, within 1e-5), divides for mean (sum / count, cuda_emitter.cc:625,
//            :903); the wave stores 1 KiB contiguous of the concat row.
// Bags longer than 64 ids, or bags that do not fit the wave's 384-entry tile, are
// walked from global memory by the lanes themselves (same arithmetic order).
// Round 1 staged at block scope (ranges, a block-wide scan and the ids of all four
// rows behind four barriers, 19.7 KB of LDS); measured against this form on RAGGED,
// E and F the two are equal within noise (30.3-30.7 us RAGGED): a launch is bounded
// by its ramp, tail and the ~2.4 us kernel boundary, not by the barriers
// (profiles/HISTORY.md, round 2).  The wave-scope form stays: 15.1 KB of LDS, one barrier.
// The kernel is instruction-issue bound rather than HBM bound (rocprofv3: ~490
// VALU per wave before this layout), hence the pre-scaled 32-bit slot offsets:
// a table read costs one LDS read, one compare, one 64-bit shift-add, one load.
// ---------------------------------------------------------------------------
// First position i in [0, n] whose segment id is >= target, in the sorted id stream of one
// column (int32, or int64 read as two dwords; element i lives at index i * stride) — the
// CSR offset ComputeSegmentOffsets (cuda_emitter.cc:768-818) would store for row `target`.
// 16-ary search: every level issues 16 independent probes, so a column with nnz ids costs
// ceil(log16 nnz) memory round trips (3 for nnz <= 4096) instead of log2 nnz.
__device__ __forceinline__ int64_t seg_at(const char *seg, bool is64, int stride, int p) {
  const int64_t e = (int64_t)p * stride;
  return is64 ? ld_i64_a4(seg + 8 * e) : (int64_t)*as_global(reinterpret_cast<const int32_t *>(seg + 4 * e));
}

// One level of the 16-ary search on [a, z]: 16 independent probes, then the interval shrinks to
// less than a 16th.
__device__ __forceinline__ void seg_narrow(const char *seg, bool is64, int stride, int target, int &a, int &z) {
  const int step = (z - a + 15) >> 4;
  int64_t v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int p = a + k * step;
    v[k] = INT64_MAX;
    if (p < z) v[k] = seg_at(seg, is64, stride, p);
  }
  int c = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) c += v[k] < (int64_t)target ? 1 : 0;
  const int valid = (z - a + step - 1) / step; // probes that were inside [a, z)
  const int na = c ? a + (c - 1) * step + 1 : a;
  if (c < valid) z = a + c * step; // probe c is >= target
  a = na;
}

__device__ __forceinline__ int seg_lower_bound(const char *seg, bool is64, int stride, int n, int target, int rows) {
  int a = 0, z = n; // the answer lies in [a, z]; everything before a is < target, everything from z on is >= target
  if (n > 256 && rows > 0) {
    // Rows hold about nnz / rows ids each, so row `target` starts near target * nnz / rows.  Search a
    // 256-wide window around that guess speculatively, together with the two probes that tell whether
    // the window brackets the answer (same round trip); if it does, one level is saved, if not the
    // full search starts over.
    const int g = (int)((int64_t)target * n / rows);
    const int lo = max(g - 128, 0), hi = min(lo + 256, n);
    const int64_t below = lo > 0 ? seg_at(seg, is64, stride, lo - 1) : INT64_MIN;
    const int64_t above = hi < n ? seg_at(seg, is64, stride, hi) : INT64_MAX;
    int wa = lo, wz = hi;
    seg_narrow(seg, is64, stride, target, wa, wz);
    if (below < (int64_t)target && above >= (int64_t)target) {
      a = wa;
      z = wz;
    }
  }
  while (z > a) seg_narrow(seg, is64, stride, target, a, z);
  return a;
}

// LDS accesses of ONE wave execute in program order; the compiler only has to keep that order.
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Where slot offset `off` of a bag is read from: the table row, or — an id that contributes nothing (out of
// range, another rank's row, dropped by the filter, past the end of the bag) — the plan's zero line: one hot
// cache line instead of a predicated read, so the walk is branch-free (no exec-mask bookkeeping around every
// read); adding its +0.0 is exact (acc is never -0.0: it starts at +0.0).
template <int V> __device__ __forceinline__ VF<V> ld_slot_or_zero(const float *tb, const float *zeros, uint32_t off, uint32_t spr) {
  typedef typename VecType<V>::T T;
  const FCP_GLOBAL T *g = is_row(off) ? as_global(reinterpret_cast<const T *>(tb)) + (uint64_t)off * spr : as_global(reinterpret_cast<const T *>(zeros));
#if defined(FCP_ABLATE) && FCP_ABLATE == 4 // timing-only build 4: no table reads at all
  g = as_global(reinterpret_cast<const T *>(zeros));
#endif
  const T t = *g;
  VF<V> r;
  __builtin_memcpy(&r, &t, sizeof(T));
  return r;
}

// The walk of one bag slice for one output slot: the n table slot offsets staged at s[0..n) are added to `acc`
// in id order (sequential fp32 adds: the order of the oracle; TF-CPU's up to 9 ids per bag), FCP_WALK table reads in flight
// per lane.  EVERY lane issues its first FCP_WALK reads at once, whatever its bag length; further batches
// only while some bag of the wave goes on.  (Round 2 walked "8, then 4" behind per-lane conditions: lanes with
// up to 4 ids sat out the first pass and issued their reads only after it.)
template <int V, int N>
__device__ __forceinline__ void bag_walk_batch(const float *tb, const float *zeros, uint32_t spr, const uint32_t *s, int base, int n, VF<V> &acc) {
  uint32_t off[N];
  VF<V> w[N];
#pragma unroll
  for (int k = 0; k < N; ++k) off[k] = base + k < n ? s[base + k] : kNoRow;
#pragma unroll
  for (int k = 0; k < N; ++k) w[k] = ld_slot_or_zero<V>(tb, zeros, off[k], spr);
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] + w[k].v[t]; // id order
}

template <int V, int WALK>
__device__ __forceinline__ void bag_walk_sum(const float *tb, const float *zeros, uint32_t spr, const uint32_t *s, int n, VF<V> &acc) {
  // The first batch is as wide as the wave's longest bag needs, up to FCP_WALK_FIRST reads per lane: every bag of
  // the wave in ONE round of reads whenever none is longer than that (BASELINE's RAGGED and the reference's models
  // E / F draw 0..10 / 1..10 ids per row: with 8-wide batches nearly every wave ran a second round for its one or two
  // 9- and 10-id bags; RAGGED 30.2 -> 28.9 us, profiles/r03_ragged_walk_width_ab.txt).  Wave-uniform choices.
  int base = WALK;
  if (WALK >= 8 && !__any(n > 4)) {
    bag_walk_batch<V, 4>(tb, zeros, spr, s, 0, n, acc);
    return;
  } else if (WALK >= 8 && FCP_WALK_FIRST > WALK && __any(n > WALK)) {
    bag_walk_batch<V, FCP_WALK_FIRST>(tb, zeros, spr, s, 0, n, acc);
    base = FCP_WALK_FIRST;
  } else {
    bag_walk_batch<V, WALK>(tb, zeros, spr, s, 0, n, acc);
  }
  for (; __any(n > base);) { // wave-uniform trip count
    if (WALK > 4 && !__any(n > base + 4)) { // a short tail (bags of 9..12 ids): half a batch
      bag_walk_batch<V, 4>(tb, zeros, spr, s, base, n, acc);
      base += 4;
    } else {
      bag_walk_batch<V, WALK>(tb, zeros, spr, s, base, n, acc);
      base += WALK;
    }
  }
}

// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive moves: four shifts inside the rows of
// 16 lanes, then the row totals broadcast to the rows after them (row_bcast:15 / row_bcast:31) — six VALU instructions
// and no LDS traffic, where six __shfl_up steps cost six ds_bpermute round trips plus their index arithmetic.
__device__ __forceinline__ int wave_inclusive_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); // row_shr:1 (lanes without a source read 0)
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2 and 3
  return x;
}

struct RaggedLds {
  static constexpr int RB = FCP_WAVES_PER_BLOCK; // rows per block, one per wave
  static constexpr int CAPW = 384;               // staged slot offsets per wave (row) and round
  LdsCol col[FCP_WAVE];
  uint32_t ids[RB][CAPW];
  uint8_t owner[RB][CAPW];                       // staged id slot -> owner lane (= column within the span)
  int32_t bound[FCP_WAVE * (FCP_WAVES_PER_BLOCK + 1)]; // seg_search: row offsets r0..r0+RB of every column
};

template <int V, bool SHARDED>
__device__ __forceinline__ void ragged_body(const FcpLaunch &L, int bid, char *smem) {
  constexpr int RB = RaggedLds::RB, CAPW = RaggedLds::CAPW;
  RaggedLds &S = *reinterpret_cast<RaggedLds *>(smem);
  LdsCol *s_col = S.col;

  BlockPos B;
#if defined(FCP_STAMPS)
  const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
#endif
  // The front of a block (records -> CSR ranges -> ids) is a chain of dependent round trips with a handful of
  // instructions between them; issued at a higher wave priority those instructions do not queue behind the long
  // walk loops of the CU's other waves (back to 0 before the walk): RAGGED -0.35 us, batch 1024 -0.5 us
  // (profiles/r03_ragged_front_priority_ab.txt).  The dense body showed no difference.
  __builtin_amdgcn_s_setprio(3);
  const Hot H = load_hot(L);
  if (!locate_block<RB>(L, H, bid, B)) return;
  const int tid = threadIdx.x;
  const int lane = tid & (FCP_WAVE - 1);
  const int wave = tid >> 6;
  const int q = B.q0 + lane;
  const uint32_t my_col = B.map[min(q, B.nslots - 1)];
  const int world = H.world, rank = H.rank;

  // ---- regular CSR (FcpLaunch::csr_reg): the wave's row ranges are requested NOW, next to the column records, instead
  // of behind them — their address needs the span's first column position only
  int pre0 = 0, pre1 = 0;
  const int csr_reg_stride = H.g0.csr_reg_stride;
  if (csr_reg_stride && lane < B.ncols && B.row_blk + wave < B.rows) {
    const FCP_GLOBAL int32_t *cr = as_global(L.csr_reg) + (int64_t)(B.first_col + lane) * csr_reg_stride + (B.row_blk + wave);
    pre0 = cr[0];
    pre1 = cr[1];
  }
  // ---- phase 0 (block) --------------------------------------------------------------------
  if (tid < B.ncols) stage_col(H, &s_col[tid], H.cols + B.first_col + tid, H.dyn + B.first_col + tid);
  __syncthreads();
#if defined(FCP_STAMPS)
  const unsigned long long t_desc = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- phase 1a' (block): segment-id columns without a pre-pass: RB+1 row offsets per column ---
  if (H.seg_search) {
    for (int u = tid; u < B.ncols * (RB + 1); u += FCP_BLOCK_THREADS) {
      const LdsCol &c = s_col[u / (RB + 1)];
      const unsigned sk = FCP_F_SEGKIND(c.flags), f = FCP_F_FORM(c.flags);
      int v = 0;
      if ((sk == FCP_SEG_IDS_I32 || sk == FCP_SEG_IDS_I64) && (f == FCP_FORM_SEGMENT_REDUCE || f == FCP_FORM_GATHER_SCATTER))
        v = seg_lower_bound(reinterpret_cast<const char *>(c.csr), sk == FCP_SEG_IDS_I64, c.seg_stride, c.nnz,
                            min(B.row_blk + u % (RB + 1), B.rows), B.rows);
      S.bound[u] = v;
    }
    __syncthreads();
  }
#if defined(FCP_STAMPS)
  const unsigned long long t_seg = __builtin_amdgcn_s_memrealtime();
#endif
  const int b = B.row_blk + wave;
  if (b >= B.rows) return; // wave-uniform; no block barrier follows

  // ---- phase 1a (wave): range [lo, lo + cnt) of (column lane, row b) ------------------------------------
  int lo = 0, cnt = 0;
  if (lane < B.ncols) {
    const unsigned form = FCP_F_FORM(s_col[lane].flags);
    if (form == FCP_FORM_GATHER) {
      lo = b;
      cnt = 1;
    } else if (form == FCP_FORM_SEGMENT_REDUCE || form == FCP_FORM_GATHER_SCATTER) {
      const int nnz = s_col[lane].nnz;
      int o0, o1;
      const unsigned sk = FCP_F_SEGKIND(s_col[lane].flags);
      if (form == FCP_FORM_GATHER_SCATTER && sk != FCP_SEG_CSR_I32) {
        // ScatterNd with its row ids as delivered, in ANY order (GatherScatterRows, cuda_emitter.cc:296-345):
        // the pre-pass left "1 + position of the last id written to row b" (0: none) in the column's scratch
        const int t = as_global(s_col[lane].csr)[b];
        o0 = t - 1;
        o1 = t > 0 ? t : -1;
      } else if (H.seg_search && sk != FCP_SEG_CSR_I32) {
        o0 = S.bound[lane * (RB + 1) + wave];
        o1 = S.bound[lane * (RB + 1) + wave + 1];
      } else if (csr_reg_stride) { // requested in front of the records (above)
        o0 = pre0;
        o1 = pre1;
      } else {
        const FCP_GLOBAL int32_t *csr = as_global(s_col[lane].csr);
        o0 = csr[b];
        o1 = csr[b + 1];
      }
      lo = min(max(o0, 0), nnz);
      const int hi = min(max(o1, lo), nnz);
      cnt = hi - lo;
      // GATHER_SCATTER: the last id of the row wins; a row with several ids (duplicate row ids) is walked
      // whole only when an id filter may drop its last ones
      if (form == FCP_FORM_GATHER_SCATTER && cnt > 1 && (s_col[lane].xform & 3u) != FCP_XFORM_FILTER) {
        lo = hi - 1;
        cnt = 1;
      }
    }
  }

  // my slot: which column (its facts are re-read from LDS where they are used: few registers live across the staging)
  const int j = (int)(my_col - B.first_col);
  const bool live = q < B.nslots;
  VF<V> acc = vzero<V>();
  int dropped = 0; // ids the column's filter removed: they do not count in a mean

  uint8_t *ow = S.owner[wave];
  uint32_t *wi = S.ids[wave];
  // ---- phase 1b (wave): slices of the wave's offset tile — a prefix sum over the lanes' bag lengths; a bag gets
  // what is left of the tile after the bags of the lanes before it (`take` of its `cnt` ids; all of them unless the
  // row holds more than CAPW ids, see "long bags" below)
  const int want = min(cnt, CAPW);
  const int incl = wave_inclusive_sum(want);
  const int offx = incl - want;
  const int take = max(min(want, CAPW - offx), 0);
  const int limit = min(__shfl(incl, FCP_WAVE - 1), CAPW);
  if (take <= 16)
    for (int i = 0; i < take; ++i) ow[offx + i] = (uint8_t)lane; // fire-and-forget LDS writes
  for (unsigned long long big = __ballot(take > 16); big; big &= big - 1) { // long slices are marked by the whole wave
    const int p = __ffsll((long long)big) - 1;
    const int po = __shfl(offx, p), pt = __shfl(take, p);
    for (int i = lane; i < pt; i += FCP_WAVE) ow[po + i] = (uint8_t)p;
  }
  wave_lds_order();
#if defined(FCP_STAMPS)
  const unsigned long long t_scan = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- one lane per staged id -> table slot offset in the wave's tile ---------------------------------------
  for (int base = 0; base < limit; base += FCP_WAVE) { // uniform trip count: the cross-lane reads need every lane
    const int k = base + lane;
    const int p = k < limit ? (int)ow[k] : 0;
    const int px = __shfl(offx, p), pl = __shfl(lo, p);
    if (k < limit) {
      bool bad;
      wi[k] = fetch_slot_offset<V, SHARDED>(s_col[p], L.xforms + B.first_col + p, pl + (k - px), nullptr, rank, world, bad);
      // a column that straddles two spans is staged by two blocks: the one holding its first slot counts.  (ScatterNd
      // columns count where the row's winner is known: an id that a later write replaces never reached the output.)
      if (bad && H.bad_ids && s_col[p].out_off >= B.q0 * V && FCP_F_FORM(s_col[p].flags) != FCP_FORM_GATHER_SCATTER)
        atomicAdd(H.bad_ids, 1ull);
    }
  }
  wave_lds_order();
#if defined(FCP_STAMPS)
  const unsigned long long t_ids = __builtin_amdgcn_s_memrealtime();
#endif

  __builtin_amdgcn_s_setprio(0);
  // ---- phase 2 (wave): the owning lanes consume their column's slice ------------------------------------------
  // (a slot's column facts are re-read from LDS where they are used: few registers live across the staging)
  auto consume = [&](auto walk_width, const uint32_t *s, int n) __attribute__((always_inline)) {
    constexpr int WALK = decltype(walk_width)::value;
    const unsigned form = FCP_F_FORM(s_col[j].flags);
    const float *tb = s_col[j].table + (q * V - s_col[j].out_off);
    const uint32_t spr = (uint32_t)(s_col[j].dim / V); // slots per table row
    if (form == FCP_FORM_SEGMENT_REDUCE) {
      if ((s_col[j].xform & 3u) == FCP_XFORM_FILTER && FCP_F_COMBINER(s_col[j].flags) == FCP_COMBINER_MEAN) {
        // ids the filter dropped do not count in the mean: a separate pass over the staged offsets, only for
        // such columns (counting inside the walk cost every column 12 registers)
#pragma unroll 1
        for (int k = 0; k < n; ++k) dropped += s[k] == kFiltered;
      }
      bag_walk_sum<V, WALK>(tb, H.zeros, spr, s, n, acc);
    } else if (form == FCP_FORM_GATHER || form == FCP_FORM_GATHER_SCATTER) {
      // a pure copy of one row (rows without ids stay zero); of several ids the last one the filter kept wins
      // (TF: ScatterNd after the filter op; the oracle compacts first)
      int k = n - 1;
      while (k > 0 && s[k] == kFiltered) --k;
      const uint32_t off = s[k];
      if (off != kFiltered) {
        acc = vzero<V>();
        if (is_row(off)) acc = ld_slot<V>(tb, off, spr);
        // the winner of a ScatterNd row is out of the vocabulary: counted once, by the lane of the column's first slot
        if (form == FCP_FORM_GATHER_SCATTER && off == kBadRow && H.bad_ids && q * V == s_col[j].out_off) atomicAdd(H.bad_ids, 1ull);
      }
    }
  };
  {
    const int ptake = __shfl(take, j), poff = __shfl(offx, j);
    if (live && ptake > 0) consume(std::integral_constant<int, FCP_WALK>(), wi + poff, ptake);
  }

  // ---- long bags (rare: a row whose bags hold more than CAPW ids, e.g. multi-hot history features of hundreds
  // of ids): what the tile could not take in goes through it in further rounds.  Every bag that has ids left gets
  // an EQUAL share of the tile per round (a power of two: position -> (bag, index) is a shift and a mask; no
  // prefix sum), so all the wave's lanes keep walking their own bags at once, and the owning lanes go on adding
  // to their running sums — the order of the adds is the order of the ids, however they are chunked.  (Round 2
  // walked such bags from global memory, one dependent id read -> row read pair at a time.)
  for (int done = take;;) {
    const int rem = cnt - done;
    const unsigned long long act = __ballot(rem > 0);
    if (!act) break; // wave-uniform
    const int nact = __popcll(act);
    const int sh_log2 = 31 - __clz(CAPW / nact); // nact <= 64: at least 4 ids per bag and round
    const int share = 1 << sh_log2;
    const int my_rank = __popcll(act & ((1ull << lane) - 1ull));
    const int tk = rem > 0 ? min(rem, share) : 0;
    wave_lds_order(); // the tile's previous contents have been consumed
    if (rem > 0) ow[my_rank] = (uint8_t)lane;
    wave_lds_order();
    const int limit = nact << sh_log2;
    const int from = lo + done;
    for (int base = 0; base < limit; base += FCP_WAVE) { // uniform trip count
      const int k = base + lane;
      const int p = k < limit ? (int)ow[k >> sh_log2] : 0;
      const int i = k & (share - 1);
      const int ptk = __shfl(tk, p), pf = __shfl(from, p);
      if (k < limit && i < ptk) {
        bool bad;
        wi[k] = fetch_slot_offset<V, SHARDED>(s_col[p], L.xforms + B.first_col + p, pf + i, nullptr, rank, world, bad);
        if (bad && H.bad_ids && s_col[p].out_off >= B.q0 * V && FCP_F_FORM(s_col[p].flags) != FCP_FORM_GATHER_SCATTER)
          atomicAdd(H.bad_ids, 1ull);
      }
    }
    wave_lds_order();
    const int ptake = __shfl(tk, j), prank = __shfl(my_rank, j);
    if (live && ptake > 0) consume(std::integral_constant<int, FCP_WALK_LONG>(), wi + (prank << sh_log2), ptake);
    done += tk;
  }
  const int pcnt = __shfl(cnt, j);
  const LdsCol &C = s_col[j];
  const unsigned form = FCP_F_FORM(C.flags);
  if (!live || form == FCP_FORM_EXTERNAL) return; // EXTERNAL: somebody else's slot (ConcatOutputs host input), never written here

  const int dim = C.dim;
  const int e = q * V - C.out_off;
  if (form == FCP_FORM_PASSTHROUGH) {
    if (rank == 0) acc = ld_blob_f32<V>(C.ids + 4 * ((int64_t)b * dim + e)); // table-free: shard rank 0
  } else if (form == FCP_FORM_BATCH_COL_REDUCTION) {
    // cuda_emitter.cc:1231-1236: r ascending, sequential fp32 adds
    const int inner = rank == 0 ? C.inner : 0;
    for (int rr = 0; rr < inner; ++rr) {
      const VF<V> x = ld_blob_f32<V>(C.ids + 4 * (((int64_t)b * inner + rr) * dim + e));
#pragma unroll
      for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] + x.v[t];
    }
  } else if (form == FCP_FORM_SEGMENT_REDUCE && !SHARDED && FCP_F_COMBINER(C.flags) == FCP_COMBINER_MEAN && pcnt > dropped) {
    const float fc = (float)(pcnt - dropped); // sum / count of the ids that reached the lookup
#pragma unroll
    for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] / fc;
  }
  st_out<V>(reinterpret_cast<float *>(H.arena + C.out_base) + e + (int64_t)b * C.out_stride, acc, H.store_through);
#if defined(FCP_STAMPS)
  if (L.stamps && tid == 0) { // wave 0 = first row of the block
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long *o = L.stamps + 8ull * bid;
    o[0] = t_begin;
    o[1] = t_desc;
    o[4] = t_seg;
    o[5] = t_scan;
    o[6] = t_scan;
    o[2] = t_ids;
    o[3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <int V, bool SHARDED>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) fcp_ragged_kernel(const FcpLaunch L) {
  __shared__ __attribute__((aligned(16))) char smem[sizeof(RaggedLds)];
  ragged_body<V, SHARDED>(L, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------
// Hybrid launch: plans that mix one-hot and pooled columns (the reference's models
// E / F: ~98 % bucketize / hash one-hot columns plus a few multi-hot ones).  Spans
// whose columns are all GATHER / PASSTHROUGH run the dense body, the other spans
// the ragged body — in ONE launch (block-uniform branch, one LDS buffer carved by
// either body), because these models are launch-latency bound: as separate
// dependent launches they cost 27.6 us per request, see DESIGN.md.
// ---------------------------------------------------------------------------
struct FcpHybridLaunch {
  FcpLaunch ragged; // blocks [0, ragged_blocks)
  FcpLaunch dense;  // blocks [ragged_blocks, grid)
  int32_t ragged_blocks;
};

template <int V, int R, bool SHARDED>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) fcp_hybrid_kernel(const FcpHybridLaunch H) {
  constexpr size_t kSmem = sizeof(RaggedLds) > sizeof(DenseLds<R>) ? sizeof(RaggedLds) : sizeof(DenseLds<R>);
  __shared__ __attribute__((aligned(16))) char smem[kSmem];
  const int bid = blockIdx.x;
  if (bid < H.ragged_blocks) {
    ragged_body<V, SHARDED>(H.ragged, bid, smem); // the longer-running blocks are dispatched first
  } else {
    dense_body<V, R, SHARDED>(H.dense, bid - H.ragged_blocks, smem);
  }
}

// ---------------------------------------------------------------------------
// Segment-offset pre-pass: sorted segment ids -> CSR offsets[0..rows]
// (experiment::ComputeSegmentOffsets, cuda_emitter.cc:768-818: position idx
// writes offsets[id] = idx for id in (seg[idx-1], seg[idx]], seg[-1] = -1,
// seg[nnz] = rows).  The reference runs this serially inside one block per
// column; here one thread per position, all columns in one launch; the
// predecessor's id comes from the neighbouring lane (wave shuffle) and a
// wave without any segment boundary retires on one ballot.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t load_seg(const char *seg, unsigned segkind, int stride, int64_t i) {
  if (segkind == FCP_SEG_IDS_I32) return *reinterpret_cast<const int32_t *>(seg + 4 * i * stride);
  return ld_i64_a4(seg + 8 * i * stride);
}

// Segment id of element i when it is a function of several index coordinates (FcpSegMap: a SparseReshape folded into
// the index expression, cuda_emitter.cc:1874-1916): (sum_k idx[i*stride + k] * mul[k]) / div, one factor scaled by the
// request's symbol.  Rare columns, cold code: out of line, 64-bit division and all.
__device__ __noinline__ int64_t load_seg_mapped(const char *seg, unsigned segkind, int stride, int64_t i, const FcpSegMap *mp,
                                                int64_t sym) {
  const FcpSegMap m = *mp;
  int64_t lin = 0;
  for (int k = 0; k < m.n; ++k) {
    const int64_t v = load_seg(seg, segkind, 1, i * stride + k);
    if (v < 0) return -1; // not an index: sorts before every row
    lin += v * (m.sym_slot == k ? m.mul[k] * sym : m.mul[k]);
  }
  const int64_t div = m.sym_slot == 4 ? m.div * sym : m.div;
  return lin / div;
}

// One block = FCP_SEG_IDS_PER_BLOCK consecutive positions of one column's id stream, in rounds of 256 (neighbouring
// lanes hold neighbouring ids); every load of the block is issued before the first boundary test, so the block is one
// memory round trip long whatever the number of rounds.  (Round 1: one id per thread, 4x the blocks — the same time
// alone, but with several requests in flight the small blocks held wave slots the other requests' kernels wanted:
// RAGGED with SparseTensor indices 28-30 us per request overlapped, 26-27 us with this form.)
#ifndef FCP_SEG_ROUNDS
#define FCP_SEG_ROUNDS 4
#endif
#define FCP_SEG_IDS_PER_BLOCK (FCP_SEG_ROUNDS * FCP_BLOCK_THREADS)
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_segment_offsets_kernel(const FcpSegLaunch L) {
  const int c = L.seg_cols[blockIdx.y];
  const FcpColDyn cd = L.dyn[c];
  const int nnz = cd.nnz;
  const int64_t base = (int64_t)blockIdx.x * FCP_SEG_IDS_PER_BLOCK;
  if (base > nnz) return;
  const FcpColStatic cs = L.cols[c];
  const unsigned segkind = FCP_F_SEGKIND(cs.flags);
  const int stride = cs.seg_stride;
  const int64_t rows = cd.rows;
  const char *seg = L.blob + cd.seg_off;
  const int lane = threadIdx.x & (FCP_WAVE - 1);
  int32_t *csr = reinterpret_cast<int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base;
  if (FCP_F_FORM(cs.flags) == FCP_FORM_GATHER_SCATTER) {
    // ScatterNd row ids arrive in ANY order (GatherScatterRows, cuda_emitter.cc:296-345, scatters whatever it is
    // given): instead of row offsets the column's scratch (zeroed before this launch) receives the inverse map
    // inv[row] = 1 + the LAST position whose row id is `row` (atomic max: "the last write wins", the order of a
    // sequential scatter and of the oracle); ids the column's filter drops never reach the scatter.
    if (L.skip_inverse) return;
    const bool filtered = (cs.xform & 3u) == FCP_XFORM_FILTER;
    LdsCol lc;
    if (filtered) {
      static_cast<FcpColStatic &>(lc) = cs;
      lc.ids = L.blob + cd.ids_off;
      lc.csr = nullptr;
      lc.out_base = 0;
      lc.out_stride = 0;
      lc.nnz = nnz;
      lc.bnd_off = -1; // boundaries, if any, are read from global memory
    }
#pragma unroll 1
    for (int r = 0; r < FCP_SEG_ROUNDS; ++r) {
      const int64_t i = base + r * FCP_BLOCK_THREADS + threadIdx.x;
      if (i >= nnz) break;
      if (filtered) { // the filter op sits in front of the scatter: what it drops never gets there
        bool bad;
        if (fetch_slot_offset<1, false>(lc, L.xforms + c, i, nullptr, 0, 1, bad) == kFiltered) continue;
      }
      const int64_t row = load_seg(seg, segkind, stride, i);
      if (row < 0 || row >= rows) { // TF's ScatterNd on a GPU drops such rows
        if (L.bad_ids) atomicAdd(L.bad_ids, 1ull);
        continue;
      }
      atomicMax(csr + row, (int32_t)(i + 1));
    }
    return;
  }
  int64_t cur[FCP_SEG_ROUNDS], first_prev[FCP_SEG_ROUNDS];
  if (L.segmaps && L.segmaps[c].n > 0) { // (uniform per block)
    const FcpSegMap *mp = L.segmaps + c;
    const int64_t sym = cd.seg_sym;
#pragma unroll 1
    for (int r = 0; r < FCP_SEG_ROUNDS; ++r) {
      const int64_t i = base + r * FCP_BLOCK_THREADS + threadIdx.x;
      cur[r] = rows;
      if (i < nnz) cur[r] = load_seg_mapped(seg, segkind, stride, i, mp, sym);
      first_prev[r] = -1;
      if (lane == 0 && i > 0 && i <= nnz) first_prev[r] = load_seg_mapped(seg, segkind, stride, i - 1, mp, sym);
    }
  } else {
#pragma unroll
    for (int r = 0; r < FCP_SEG_ROUNDS; ++r) {
      const int64_t i = base + r * FCP_BLOCK_THREADS + threadIdx.x;
      cur[r] = rows;
      if (i < nnz) cur[r] = load_seg(seg, segkind, stride, i);
      first_prev[r] = -1;
      if (lane == 0 && i > 0 && i <= nnz) first_prev[r] = load_seg(seg, segkind, stride, i - 1); // the neighbour wave's last id
    }
  }
#pragma unroll
  for (int r = 0; r < FCP_SEG_ROUNDS; ++r) {
    const int64_t i = base + r * FCP_BLOCK_THREADS + threadIdx.x;
    const bool active = i <= nnz;
    const int64_t c0 = cur[r] > rows ? rows : cur[r];
    int64_t prev = __shfl_up(c0, 1);
    if (lane == 0) prev = first_prev[r] > rows ? rows : first_prev[r];
    const bool boundary = active && c0 > prev;
    // sorted-ascending is the caller's contract (TF SparseSegment*, SparseTensor indices); a descending step is
    // reported through the bad-id counter when the plan asks for it
    if (L.bad_ids && active && i < nnz && c0 < prev) atomicAdd(L.bad_ids, 1ull);
    if (boundary)
      for (int64_t id = prev + 1 < 0 ? 0 : prev + 1; id <= c0; ++id) csr[id] = (int32_t)i;
  }
}

// ---------------------------------------------------------------------------
// ConcatOutputs (reference layout pass, concat_outputs_op_gpu.cu.cc:85-131):
// out[p, off_k + e] = in_k[p*dim_k + e].  Used with FCP_LAYOUT_PER_COLUMN (hundreds of narrow inputs), by
// the host half of Addons>ConcatOutputs (a few payloads scattered into the EXTERNAL slots) and by the
// column-sharded step (8 wide blocks side by side: there it is half of a rank's work per request).
// One thread per VEC floats; a block is a tile of TY rows x TX vectors of one input, TX = the widest
// input of the launch rounded up to a power of two (at most 256), so that narrow inputs still fill
// their waves with rows.  No integer division per element, no grid cap (round 1: scalar copies behind a
// 64-block grid-stride loop: 2.35 TB/s on the 8 x [64, 15000] blocks of BASELINE configs[4]).
// ---------------------------------------------------------------------------
#define FCP_CONCAT_CHUNK 192
struct FcpConcatArgs {
  const float *in[FCP_CONCAT_CHUNK];
  int32_t off[FCP_CONCAT_CHUNK];
  int32_t dim[FCP_CONCAT_CHUNK];
  int32_t stride[FCP_CONCAT_CHUNK]; // row stride of the input in floats (= dim for a contiguous [prefix, dim] input)
  float *out;
  int64_t prefix;
  int32_t width;
  int32_t n;
  int32_t tx_log2; // threads along a row
  int32_t cpr;     // column chunks per row: ceil(max_dim / VEC / TX)
};

template <int VEC> __global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_concat_outputs_kernel(const FcpConcatArgs A) {
  const int k = blockIdx.y;
  const int dimv = A.dim[k] / VEC;
  const int tx = 1 << A.tx_log2;
  const int lx = threadIdx.x & (tx - 1), ly = threadIdx.x >> A.tx_log2;
  const int64_t tile = blockIdx.x / A.cpr;
  const int chunk = (int)(blockIdx.x - tile * A.cpr);
  const int64_t p = tile * (FCP_BLOCK_THREADS >> A.tx_log2) + ly;
  const int e = chunk * tx + lx;
  if (p >= A.prefix || e >= dimv) return;
  typedef typename VecType<VEC>::T T;
  const T v = *as_global(reinterpret_cast<const T *>(A.in[k]) + (p * (A.stride[k] / VEC) + e));
  __builtin_nontemporal_store(v, as_global(reinterpret_cast<T *>(A.out + (p * A.width + A.off[k])) + e));
}

// ---------------------------------------------------------------------------
// Row-shard finalize (no reference counterpart): after the
// all-to-all, rank h holds `world` partial-sum slices of its batch slice;
// add them in rank order (deterministic) and apply the mean division.
// ---------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS)
    fcp_shard_finalize_kernel(const FcpLaunch L, int g, const float *__restrict__ partials, int world,
                              int64_t row_begin, int64_t row_count, float *__restrict__ out) {
  const int nslots = L.groups[g].nslots;
  const int nxb = (nslots + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS; // blocks per output row (1-D grid: any row count)
  const int q = (int)(blockIdx.x % nxb) * FCP_BLOCK_THREADS + threadIdx.x;
  const int64_t bl = blockIdx.x / nxb;
  if (q >= nslots || bl >= row_count) return;
  const int64_t W = (int64_t)nslots * V;
  const uint32_t c = L.slot_map[L.groups[g].slot_map_off + q];
  const FcpColStatic cs = L.cols[c];
  const FcpColDyn cd = L.dyn[c];
  if (FCP_F_FORM(cs.flags) == FCP_FORM_EXTERNAL) return; // the hole stays for fcp_concat_outputs_host
  VF<V> acc = vzero<V>();
  for (int w = 0; w < world; ++w) {
    const VF<V> x = *reinterpret_cast<const VF<V> *>(partials + ((int64_t)w * row_count + bl) * W + (int64_t)q * V);
#pragma unroll
    for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] + x.v[t];
  }
  // (a plan with shard_world == 1 is not sharded: its kernels have already divided)
  if (L.shard_world > 1 && FCP_F_FORM(cs.flags) == FCP_FORM_SEGMENT_REDUCE && FCP_F_COMBINER(cs.flags) == FCP_COMBINER_MEAN) {
    const unsigned segkind = FCP_F_SEGKIND(cs.flags);
    const int32_t *csr = segkind == FCP_SEG_CSR_I32
                             ? reinterpret_cast<const int32_t *>(L.blob + cd.seg_off)
                             : reinterpret_cast<const int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base;
    const int64_t b = row_begin + bl;
    int lo = csr[b], hi = csr[b + 1];
    lo = min(max(lo, 0), cd.nnz);
    hi = min(max(hi, lo), cd.nnz);
    int kept = hi - lo;
    if ((cs.xform & 3u) == FCP_XFORM_FILTER) {
      // ids the column's filter drops do not count in the mean; which ones they are does not depend on the
      // rank (hash and intervals are applied to the raw id, before the ownership test), so the finalizing
      // rank re-reads the row's ids and counts the kept ones
      LdsCol lc;
      static_cast<FcpColStatic &>(lc) = cs;
      lc.ids = L.blob + cd.ids_off;
      lc.csr = nullptr;
      lc.out_base = 0;
      lc.out_stride = 0;
      lc.nnz = cd.nnz;
      lc.bnd_off = -1; // boundaries, if any, are read from global memory
      kept = 0;
      for (int i = lo; i < hi; ++i) {
        bool bad;
        kept += fetch_slot_offset<V, false>(lc, L.xforms + c, i, nullptr, 0, 1, bad) != kFiltered ? 1 : 0;
      }
    }
    if (kept > 0) {
      const float fc = (float)kept;
#pragma unroll
      for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] / fc;
    }
  }
  *reinterpret_cast<VF<V> *>(out + bl * W + (int64_t)q * V) = acc;
}


// ---------------------------------------------------------------------------
// Descriptor upload: copies the request's FcpColDyn[] from pinned host memory
// (read over PCIe through its device mapping) into device memory.  Stands in
// for the reference's per-call cudaMemcpyAsync of KnlArgs (cuda_emitter.cc
// :2216); hipMemcpyAsync of these ~10-50 KB cost ~28 us per request end to end
// on this path, a 2-block kernel costs a few.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_upload_kernel(const uint4 *__restrict__ src,
                                                                       uint4 *__restrict__ dst, int n16) {
  for (int i = blockIdx.x * FCP_BLOCK_THREADS + threadIdx.x; i < n16; i += gridDim.x * FCP_BLOCK_THREADS)
    dst[i] = src[i];
}

} // namespace

// ---------------------------------------------------------------------------
// The request stager's copy as a KERNEL (FCP_STAGER_COPY_KERNEL): bytes [0, n) from the pinned staging ring (read over
// PCIe through its device mapping) to the slot's device twin.  src and dst are 4-byte aligned and equally misaligned
// against 16 bytes (the same offset into two page-aligned buffers), n is a multiple of 4: head and tail by dwords, the
// body by 16-byte words, four independent loads in flight per thread (a PCIe read is ~2 us away: 64 blocks x 256
// threads x 4 x 16 B = 1 MB requested per round).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_h2d_copy_kernel(const char *__restrict__ src, char *__restrict__ dst, size_t n) {
  const size_t head = min((size_t)((16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15), n);
  const size_t body = (n - head) & ~(size_t)15, tail = n - head - body;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  if (t < head / 4) reinterpret_cast<uint32_t *>(dst)[t] = reinterpret_cast<const uint32_t *>(src)[t];
  if (t < tail / 4) reinterpret_cast<uint32_t *>(dst + head + body)[t] = reinterpret_cast<const uint32_t *>(src + head + body)[t];
  const uint4 *s4 = reinterpret_cast<const uint4 *>(src + head);
  uint4 *d4 = reinterpret_cast<uint4 *>(dst + head);
  const size_t n4 = body / 16;
  for (size_t i = t; i < n4; i += 4 * nt) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * nt < n4) v[k] = s4[i + k * nt];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * nt < n4) d4[i + k * nt] = v[k];
  }
}

// ------------------------------- launchers ---------------------------------

// Private-stream requests (fcp_lanes.hip): the completion event of a request is attached to the dispatch packet of its
// LAST kernel (hipExtLaunchKernelGGL's stop event) instead of being recorded as a marker packet of its own behind it.
// Thread-local: set by the request path just before it enqueues, taken (and cleared) by the fused / hybrid launcher.
static thread_local hipEvent_t tl_stop_event = nullptr;
void fcp_set_stop_event(void *ev) { tl_stop_event = static_cast<hipEvent_t>(ev); }
bool fcp_stop_event_pending() { return tl_stop_event != nullptr; }
// FCP_ORDER_INPUTS_READY plans (fcp_plan_set_request_order): the next fused / hybrid launch of this thread goes out WITHOUT
// the barrier bit (hipExtAnyOrderLaunch): the command processor need not wait for the queue to drain before it takes the
// packet, which makes the hand-over between two requests cheaper (S2: 28.3 -> 26.0-26.5 us per request back to back on one
// stream, no events, no extra streams).  It does NOT start while blocks of the kernel in front of it still run, not even
// into its tail (round 5, scripts/probes/any_order_probe.hip: 21.4 us after the start of a predecessor whose blocks retire
// between 10 and 20 us, against 22.7 us with the barrier bit).
static thread_local int tl_launch_flags = 0;
void fcp_set_any_order(bool on) { tl_launch_flags = on ? (int)hipExtAnyOrderLaunch : 0; }
#define FCP_KLAUNCH(KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                      \
  do {                                                                                          \
    hipEvent_t stop_ = tl_stop_event;                                                           \
    const int flags_ = tl_launch_flags;                                                         \
    tl_stop_event = nullptr;                                                                    \
    tl_launch_flags = 0;                                                                        \
    if (stop_ || flags_)                                                                        \
      hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, nullptr, stop_, flags_, __VA_ARGS__); \
    else                                                                                        \
      hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                        \
  } while (0)

#define FCP_LAUNCH_DENSE(VV, RR)                                                                            \
  do {                                                                                                      \
    if (L.shard_world > 1)                                                                                  \
      FCP_KLAUNCH((fcp_dense_kernel<VV, RR, true>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), lds_pad, s, L); \
    else                                                                                                    \
      FCP_KLAUNCH((fcp_dense_kernel<VV, RR, false>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), lds_pad, s, L); \
  } while (0)
#define FCP_LAUNCH_RAGGED(VV)                                                                               \
  do {                                                                                                      \
    if (L.shard_world > 1)                                                                                  \
      FCP_KLAUNCH((fcp_ragged_kernel<VV, true>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), lds_pad, s, L); \
    else                                                                                                    \
      FCP_KLAUNCH((fcp_ragged_kernel<VV, false>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), lds_pad, s, L); \
  } while (0)

// rows_per_wave: dense 1 | 2 | 4 (rows per block = 4 x that); ragged always 1.
int fcp_launch_fused(const FcpLaunch &L, int vec, bool dense_kernel, int grid_blocks, ihipStream_t *s) {
  if (grid_blocks <= 0) return 0;
  // tuning aid: FCP_DIAG=lds_pad=<bytes> of unused dynamic LDS caps the blocks per CU
  static const int lds_pad = (int)fcp::diag_ll("lds_pad", 0);
  if (dense_kernel) {
    const int R = L.rows_per_wave;
#define FCP_DENSE_R(VV)                         \
  switch (R) {                                  \
  case 4: FCP_LAUNCH_DENSE(VV, 4); break;       \
  case 2: FCP_LAUNCH_DENSE(VV, 2); break;       \
  default: FCP_LAUNCH_DENSE(VV, 1); break;      \
  }
    if (vec == 4) {
      FCP_DENSE_R(4)
    } else if (vec == 2) {
      FCP_DENSE_R(2)
    } else {
      FCP_DENSE_R(1)
    }
#undef FCP_DENSE_R
  } else {
    if (vec == 4) {
      FCP_LAUNCH_RAGGED(4);
    } else if (vec == 2) {
      FCP_LAUNCH_RAGGED(2);
    } else {
      FCP_LAUNCH_RAGGED(1);
    }
  }
  return (int)hipGetLastError();
}

int fcp_launch_hybrid(const FcpLaunch &Ldense, int dense_blocks, const FcpLaunch &Lragged, int ragged_blocks, int vec,
                      ihipStream_t *s) {
  FcpHybridLaunch H;
  H.ragged = Lragged;
  H.dense = Ldense;
  H.ragged_blocks = ragged_blocks;
  const dim3 grid(dense_blocks + ragged_blocks), block(FCP_BLOCK_THREADS);
  const bool sharded = Ldense.shard_world > 1;
#define FCP_HYB(VV, RR)                                                                   \
  do {                                                                                    \
    if (sharded)                                                                          \
      FCP_KLAUNCH((fcp_hybrid_kernel<VV, RR, true>), grid, block, 0, s, H);               \
    else                                                                                  \
      FCP_KLAUNCH((fcp_hybrid_kernel<VV, RR, false>), grid, block, 0, s, H);              \
  } while (0)
#define FCP_HYB_R(VV)                        \
  switch (Ldense.rows_per_wave) {            \
  case 4: FCP_HYB(VV, 4); break;             \
  case 2: FCP_HYB(VV, 2); break;             \
  default: FCP_HYB(VV, 1); break;            \
  }
  if (vec == 4) {
    FCP_HYB_R(4)
  } else if (vec == 2) {
    FCP_HYB_R(2)
  } else {
    FCP_HYB_R(1)
  }
#undef FCP_HYB_R
#undef FCP_HYB
  return (int)hipGetLastError();
}

int fcp_launch_h2d_copy(const void *host_mapped_src, void *dst, size_t bytes, ihipStream_t *s) {
  if (bytes == 0) return 0;
  if ((bytes & 3) || ((reinterpret_cast<uintptr_t>(host_mapped_src) ^ reinterpret_cast<uintptr_t>(dst)) & 15) ||
      (reinterpret_cast<uintptr_t>(dst) & 3))
    return (int)hipErrorInvalidValue;
  const int blocks = (int)std::min<size_t>(64, (bytes / 16 + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS + 1);
  hipLaunchKernelGGL(fcp_h2d_copy_kernel, dim3(blocks), dim3(FCP_BLOCK_THREADS), 0, s, static_cast<const char *>(host_mapped_src),
                     static_cast<char *>(dst), bytes);
  return (int)hipGetLastError();
}

int fcp_launch_upload(const void *host_mapped_src, void *dst, size_t bytes, ihipStream_t *s) {
  const int n16 = (int)((bytes + 15) / 16);
  if (n16 <= 0) return 0;
  const int blocks = n16 >= 4096 ? 8 : (n16 >= 1024 ? 4 : 1);
  hipLaunchKernelGGL(fcp_upload_kernel, dim3(blocks), dim3(FCP_BLOCK_THREADS), 0, s,
                     static_cast<const uint4 *>(host_mapped_src), static_cast<uint4 *>(dst), n16);
  return (int)hipGetLastError();
}

// any_order: FCP_ORDER_INPUTS_READY plans — the pre-pass reads the blob and writes the new arena's scratch only, so it needs
// no barrier against the previous request's kernel (a cheaper hand-over); the fused kernel behind it keeps the barrier bit
// and waits for it
int fcp_launch_segment_offsets(const FcpSegLaunch &L, int n_seg_cols, int max_nnz, ihipStream_t *s, bool any_order) {
  if (n_seg_cols <= 0) return 0;
  const int gx = (max_nnz + 1 + FCP_SEG_IDS_PER_BLOCK - 1) / FCP_SEG_IDS_PER_BLOCK;
  if (any_order)
    hipExtLaunchKernelGGL(fcp_segment_offsets_kernel, dim3(gx, n_seg_cols), dim3(FCP_BLOCK_THREADS), 0, s, nullptr, nullptr,
                          (int)hipExtAnyOrderLaunch, L);
  else
    hipLaunchKernelGGL(fcp_segment_offsets_kernel, dim3(gx, n_seg_cols), dim3(FCP_BLOCK_THREADS), 0, s, L);
  return (int)hipGetLastError();
}

int fcp_launch_concat_outputs(const void *const *inputs, const int32_t *dims, const int32_t *col_offsets, const int32_t *in_strides,
                              int32_t n, int64_t prefix, int32_t width, int32_t first_off, void *out,
                              ihipStream_t *s) {
  int32_t off = first_off;
  for (int32_t begin = 0; begin < n; begin += FCP_CONCAT_CHUNK) {
    FcpConcatArgs A;
    const int32_t m = (n - begin) < FCP_CONCAT_CHUNK ? (n - begin) : FCP_CONCAT_CHUNK;
    int32_t max_dim = 1;
    for (int32_t k = 0; k < m; ++k) {
      A.in[k] = static_cast<const float *>(inputs[begin + k]);
      A.dim[k] = dims[begin + k];
      A.stride[k] = in_strides ? in_strides[begin + k] : dims[begin + k];
      A.off[k] = col_offsets ? col_offsets[begin + k] : off;
      off += dims[begin + k];
      if (dims[begin + k] > max_dim) max_dim = dims[begin + k];
    }
    A.out = static_cast<float *>(out);
    A.prefix = prefix;
    A.width = width;
    A.n = m;
    // widest vector every address of the launch is aligned for
    int vec = 4;
    uintptr_t bits = reinterpret_cast<uintptr_t>(out) | (uintptr_t)(4u * (uint32_t)width);
    for (int32_t k = 0; k < m; ++k)
      bits |= reinterpret_cast<uintptr_t>(A.in[k]) | (uintptr_t)(4u * (uint32_t)A.dim[k]) | (uintptr_t)(4u * (uint32_t)A.off[k]) |
              (uintptr_t)(4u * (uint32_t)A.stride[k]);
    while (vec > 1 && (bits & (uintptr_t)(4 * vec - 1))) vec >>= 1;
    const int max_dimv = (max_dim + vec - 1) / vec;
    int tx_log2 = 0;
    while ((1 << tx_log2) < max_dimv && (1 << tx_log2) < FCP_BLOCK_THREADS) ++tx_log2;
    A.tx_log2 = tx_log2;
    A.cpr = (max_dimv + (1 << tx_log2) - 1) >> tx_log2;
    const int ty = FCP_BLOCK_THREADS >> tx_log2;
    const int64_t gx = (prefix + ty - 1) / ty * A.cpr;
    if (gx <= 0) continue;
    if (gx > 0x7fffffff) return (int)hipErrorInvalidValue;
    const dim3 grid((unsigned)gx, (unsigned)m), block(FCP_BLOCK_THREADS);
    if (vec == 4)
      hipLaunchKernelGGL(fcp_concat_outputs_kernel<4>, grid, block, 0, s, A);
    else if (vec == 2)
      hipLaunchKernelGGL(fcp_concat_outputs_kernel<2>, grid, block, 0, s, A);
    else
      hipLaunchKernelGGL(fcp_concat_outputs_kernel<1>, grid, block, 0, s, A);
    const int err = (int)hipGetLastError();
    if (err) return err;
  }
  return 0;
}

int fcp_launch_shard_finalize(const FcpLaunch &L, int group, const float *partials, int world,
                              int64_t row_begin, int64_t row_count, float *out, int vec,
                              ihipStream_t *s) {
  if (row_count <= 0) return 0;
  const int nslots = L.groups[group].nslots;
  const int64_t nblocks = (int64_t)((nslots + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS) * row_count;
  if (nblocks > 0x7fffffff) return (int)hipErrorInvalidValue;
  dim3 grid((unsigned)nblocks);
  if (vec == 4) {
    hipLaunchKernelGGL((fcp_shard_finalize_kernel<4>), grid, dim3(FCP_BLOCK_THREADS), 0, s, L, group,
                       partials, world, row_begin, row_count, out);
  } else if (vec == 2) {
    hipLaunchKernelGGL((fcp_shard_finalize_kernel<2>), grid, dim3(FCP_BLOCK_THREADS), 0, s, L, group,
                       partials, world, row_begin, row_count, out);
  } else {
    hipLaunchKernelGGL((fcp_shard_finalize_kernel<1>), grid, dim3(FCP_BLOCK_THREADS), 0, s, L, group,
                       partials, world, row_begin, row_count, out);
  }
  return (int)hipGetLastError();
}
