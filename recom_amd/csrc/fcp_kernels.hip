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
//   * the 8 ids (64 B) a wave needs from one column for its R=8 rows share a
//     cache line, and all lanes of that column broadcast one address;
//   * 1000 columns x batch 512 give ~1900 blocks / ~7500 waves instead of 1000
//     waves, enough to keep >16 MB of loads in flight (HBM latency x bandwidth);
//   * blocks that share a span (the same columns / tables) are given the same
//     `blockIdx % 8`, i.e. the same XCD and L2 under round-robin dispatch, so
//     skewed (Zipf) ids and small bucketize tables are served from one L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fcp_hip.h"
#include "fcp_internal.h"

namespace {

template <int V> struct alignas(4 * V) VF { float v[V]; };

template <int V> __device__ __forceinline__ VF<V> vzero() {
  VF<V> r;
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = 0.0f;
  return r;
}

// Blob tensors are only guaranteed 4-byte aligned (ConcatInputs packs bytes
// back to back, concat_inputs_ops.cc:52-60), so 8-byte ids are read as two
// dwords.
__device__ __forceinline__ int64_t ld_i64_a4(const char *p) {
  const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
  const uint32_t lo = q[0], hi = q[1];
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int V> __device__ __forceinline__ VF<V> ld_blob_f32(const char *p) {
  VF<V> r;
  const float *q = reinterpret_cast<const float *>(p);
#pragma unroll
  for (int i = 0; i < V; ++i) r.v[i] = q[i];
  return r;
}

// cuda_emitter.cc:233-247 — r+1 = number of boundaries <= value.
__device__ __forceinline__ int bucketize(const float *__restrict__ b, int n, float value) {
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

// The index expression the reference inlines per column (EmitInputInline,
// :1769-1949): raw int32 / int64 ids, or Bucketize(float value).  The fetch is
// branch-free on purpose — one code path for every id source, so that the
// compiler can issue the fetches of all rows of a wave back to back and wait
// once (a switch per id source serialises them behind one s_waitcnt each).
__device__ __forceinline__ void ld_raw_id(const char *ids, bool is64, int64_t p, uint32_t &lo, uint32_t &hi) {
  const char *a = ids + (is64 ? 8 : 4) * p;
  lo = *reinterpret_cast<const uint32_t *>(a);
  hi = *reinterpret_cast<const uint32_t *>(a + (is64 ? 4 : 0));
}

__device__ __forceinline__ int64_t raw_to_id(bool is64, uint32_t lo, uint32_t hi) {
  return is64 ? (int64_t)(((uint64_t)hi << 32) | lo) : (int64_t)(int32_t)lo;
}

// Validity + row sharding.  Returns true when this GPU must read a row; `id`
// becomes the local row.  Ids outside [0, vocab) read as zeros (the reference
// reads out of bounds, TF-GPU GatherV2 returns zeros).
__device__ __forceinline__ bool resolve_id(int64_t &id, int64_t vocab, int rank, int world,
                                           bool &bad) {
  bad = (uint64_t)id >= (uint64_t)vocab;
  if (bad) return false;
  if (world > 1) {
    int64_t q;
    if (id < 0x7fffffffLL) {
      q = (int64_t)((uint32_t)id / (uint32_t)world);
    } else {
      q = id / world;
    }
    if (id - q * world != rank) return false;
    id = q;
  }
  return true;
}

template <int V>
__device__ __forceinline__ VF<V> ld_row(const float *__restrict__ table, int64_t id, int dim, int e) {
  return *reinterpret_cast<const VF<V> *>(table + id * (int64_t)dim + e);
}

template <int V> struct VecType;
template <> struct VecType<4> { typedef float __attribute__((ext_vector_type(4))) T; };
template <> struct VecType<2> { typedef float __attribute__((ext_vector_type(2))) T; };
template <> struct VecType<1> { typedef float T; };

// Output rows are written once and consumed by a later kernel; table rows are
// read once per request: both use the non-temporal forms (measured on S2:
// 34.3 -> 30.5 us per request; stores give most of it).  -DFCP_NO_NT restores
// the default cache policy (tuning builds).
#if !defined(FCP_NO_NT)
#define FCP_NT_STORE 1
#define FCP_NT_LOAD 1
#endif
template <int V> __device__ __forceinline__ void st_out(float *p, const VF<V> &v) {
#if defined(FCP_NT_STORE)
  typedef typename VecType<V>::T T;
  T t;
  __builtin_memcpy(&t, &v, sizeof(T));
  __builtin_nontemporal_store(t, reinterpret_cast<T *>(p));
#else
  *reinterpret_cast<VF<V> *>(p) = v;
#endif
}

template <int V> __device__ __forceinline__ VF<V> ld_table(const float *p) {
#if defined(FCP_NT_LOAD)
  typedef typename VecType<V>::T T;
  T t = __builtin_nontemporal_load(reinterpret_cast<const T *>(p));
  VF<V> r;
  __builtin_memcpy(&r, &t, sizeof(T));
  return r;
#else
  return *reinterpret_cast<const VF<V> *>(p);
#endif
}

// ---------------------------------------------------------------------------
// Dense kernel: every column of the plan is GATHER or PASSTHROUGH (exactly one
// source row per output row) — BASELINE.json's S2 and DLRM shapes.
//
// A block owns one span (64 slots = 1 KiB of the output row) for RB = 4*R rows.
//   phase 0  the span's column records (static + dynamic, contiguous because
//            the device arrays are kept in concat order) are copied to LDS,
//            one thread per column;
//   phase 1  the block's (column, row) id pairs are fetched with one thread
//            per pair — consecutive threads take consecutive rows of one
//            column, so every id cache line is requested exactly once — turned
//            into final local row numbers (Bucketize, range check, row shard)
//            and parked in LDS;
//   phase 2  every lane reads its column record and its R row numbers from
//            LDS (broadcast reads), issues its R 16-byte table reads back to
//            back, then its R stores: 1 KiB contiguous per wave instruction,
//            straight into the concat layout.
// Without the LDS staging every lane fetched its own copy of the id and of the
// 96-byte column record: ~80 vector-memory instructions per wave and — measured
// with rocprofv3 — about half of the kernel time in TCP_PENDING_STALL_CYCLES
// (lanes and waves queueing on the same in-flight cache lines).
// ---------------------------------------------------------------------------
struct alignas(16) LdsCol {   // 80 bytes
  const float *table;         // table base, or the passthrough payload
  const char *ids;            // id / value stream of this request
  const float *boundaries;
  const int32_t *csr;         // CSR offsets of this request (blob or arena scratch), ragged kernel only
  int64_t vocab;
  int64_t out_base;           // byte offset in the arena of element (0,0)
  int32_t dim;
  int32_t out_off;
  int32_t out_stride;
  uint32_t flags;
  int32_t n_boundaries;
  int32_t bnd_off;            // offset of the staged boundaries in LDS, or -1
  int32_t nnz;
  int32_t inner;
};

__device__ __forceinline__ LdsCol make_lds_col(const FcpLaunch &L, const FcpColStatic &cs, const FcpColDyn &cd) {
  LdsCol c;
  c.ids = L.blob + cd.ids_off;
  c.table = FCP_F_FORM(cs.flags) == FCP_FORM_PASSTHROUGH ? reinterpret_cast<const float *>(c.ids) : cs.table;
  c.boundaries = cs.boundaries;
  const unsigned segkind = FCP_F_SEGKIND(cs.flags);
  c.csr = segkind == FCP_SEG_CSR_I32   ? reinterpret_cast<const int32_t *>(L.blob + cd.seg_off)
          : segkind != FCP_SEG_NONE    ? reinterpret_cast<const int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base
                                       : nullptr;
  c.vocab = cs.vocab;
  c.out_base = cd.out_base;
  c.dim = cs.dim;
  c.out_off = cs.out_off;
  c.out_stride = cd.out_stride;
  c.flags = cs.flags;
  c.n_boundaries = cs.n_boundaries;
  c.bnd_off = -1;
  c.nnz = cd.nnz;
  c.inner = cd.inner;
  return c;
}

// One id of a column -> final local row number, or -1 (out of range / owned by
// another shard).  `bnd` = boundaries in LDS when staged, else nullptr.
__device__ __forceinline__ int64_t fetch_row_number(const LdsCol &c, int64_t pos, const float *bnd, int rank,
                                                    int world, bool &bad) {
  const unsigned idsrc = FCP_F_IDSRC(c.flags);
  const bool is64 = idsrc == FCP_IDS_I64;
  uint32_t lo, hi;
  ld_raw_id(c.ids, is64, pos, lo, hi);
  int64_t v;
  if (idsrc == FCP_IDS_F32_BUCKETIZE) {
    v = bucketize(bnd ? bnd : c.boundaries, c.n_boundaries, __uint_as_float(lo));
  } else {
    v = raw_to_id(is64, lo, hi);
  }
  return resolve_id(v, c.vocab, rank, world, bad) ? v : -1;
}

template <int V, int R>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_dense_kernel(const FcpLaunch L) {
  constexpr int RB = FCP_WAVES_PER_BLOCK * R; // rows per block
  constexpr int IDS = RB + 1;                 // padded row of the id tile (LDS banks)
  constexpr int BND = 1024;                   // floats of bucketize boundaries staged per block
  __shared__ LdsCol s_col[FCP_WAVE];
  __shared__ int32_t s_id[FCP_WAVE * IDS];    // local row numbers (< 2^31, checked at plan creation)
  __shared__ float s_bnd[BND];

  int bid = blockIdx.x;
  int g = 0;
  for (int k = 1; k < L.n_groups; ++k)
    if (bid >= L.groups[k].block_begin) g = k;
  const int rows = L.groups[g].rows;
  const int nslots = L.groups[g].nslots;
  const int nsp8 = L.groups[g].nsp8;
  const uint32_t *__restrict__ map = L.slot_map + L.groups[g].slot_map_off;
  bid -= L.groups[g].block_begin;
  const int xcd = bid & 7, j8 = bid >> 3;
  const int span = (j8 % nsp8) * 8 + xcd;
  const int tile = j8 / nsp8;
  const int q0 = span * FCP_WAVE;
  const int row_blk = tile * RB;
  if (q0 >= nslots || row_blk >= rows) return; // uniform: whole block leaves

  const int tid = threadIdx.x;
  const int lane = tid & (FCP_WAVE - 1);
  const int wave = tid >> 6;
  const int q = q0 + lane;
  const int qc = min(q, nslots - 1);
  const uint32_t my_col = map[qc];
  const uint32_t first_col = map[q0];
  const int ncols = (int)(map[min(q0 + FCP_WAVE - 1, nslots - 1)] - first_col) + 1;
  const int world = L.shard_world, rank = L.shard_rank;

  // ---- phase 0: column records -> LDS -------------------------------------------
  if (tid < ncols) s_col[tid] = make_lds_col(L, L.cols[first_col + tid], L.dyn[first_col + tid]);
  __syncthreads();

  // ---- phase 0b: bucketize boundaries -> LDS (cuda_emitter.cc:1818-1825 stages
  // them per block too).  Wave 0 assigns LDS offsets with a shuffle prefix sum;
  // then all threads copy.  Columns that do not fit keep searching in L2.
  if (wave == 0) {
    int nb = 0;
    if (lane < ncols && FCP_F_IDSRC(s_col[lane].flags) == FCP_IDS_F32_BUCKETIZE &&
        FCP_F_FORM(s_col[lane].flags) != FCP_FORM_PASSTHROUGH)
      nb = s_col[lane].n_boundaries;
    int incl = nb;
#pragma unroll
    for (int d = 1; d < FCP_WAVE; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (nb > 0 && incl <= BND) s_col[lane].bnd_off = incl - nb;
  }
  __syncthreads();
  for (int j = 0; j < ncols; ++j) {
    const int off = s_col[j].bnd_off;
    if (off < 0) continue;
    const float *__restrict__ src = s_col[j].boundaries;
    for (int i = tid; i < s_col[j].n_boundaries; i += FCP_BLOCK_THREADS) s_bnd[off + i] = src[i];
  }
  __syncthreads();

  // ---- phase 1: (column, row) pairs -> final row numbers in LDS -------------------
  for (int p = tid; p < ncols * RB; p += FCP_BLOCK_THREADS) {
    const int j = p / RB, r = p % RB;
    const int b = row_blk + r;
    int64_t id = -1;
    if (b < rows) {
      const unsigned flags = s_col[j].flags;
      if (FCP_F_FORM(flags) == FCP_FORM_PASSTHROUGH) {
        id = rank == 0 ? b : -1; // table-free columns belong to shard rank 0
      } else {
        bool bad;
        const int boff = s_col[j].bnd_off;
        id = fetch_row_number(s_col[j], b, boff >= 0 ? s_bnd + boff : nullptr, rank, world, bad);
        if (bad && L.bad_ids) atomicAdd(L.bad_ids, 1ull);
      }
    }
    s_id[j * IDS + r] = (int32_t)id;
  }
  __syncthreads();
  if (q >= nslots) return;

  // ---- phase 2: R table reads in flight per lane, then R coalesced stores -----------
  const int j = (int)(my_col - first_col);
  const float *table = s_col[j].table;
  const int dim = s_col[j].dim;
  const int e = q * V - s_col[j].out_off;
  const int64_t ostride = s_col[j].out_stride;
  float *outp = reinterpret_cast<float *>(L.arena + s_col[j].out_base) + e;
  const int r0 = wave * R;
  int64_t id[R];
#pragma unroll
  for (int r = 0; r < R; ++r) id[r] = s_id[j * IDS + r0 + r];
#if defined(FCP_ABLATE) && FCP_ABLATE == 3 // timing-only build: sequential instead of random rows
#pragma unroll
  for (int r = 0; r < R; ++r) id[r] = ((int64_t)(row_blk + r0 + r) * 131 + my_col * 977) % s_col[j].vocab;
#endif
  VF<V> v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    v[r] = vzero<V>();
#if !(defined(FCP_ABLATE) && FCP_ABLATE == 1) // timing-only build 1: no table reads
    if (id[r] >= 0) v[r] = ld_table<V>(table + id[r] * (int64_t)dim + e);
#else
    v[r].v[0] = (float)id[r];
#endif
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int b = row_blk + r0 + r;
#if defined(FCP_ABLATE) && FCP_ABLATE == 2 // timing-only build 2: no output stores
    asm volatile("" ::"v"(v[r].v[0]), "v"(v[r].v[V - 1]));
    if (b < rows && v[r].v[0] == 1234.5f) st_out<V>(outp + (int64_t)b * ostride, v[r]);
#else
    if (b < rows) st_out<V>(outp + (int64_t)b * ostride, v[r]);
#endif
  }
}


// ---------------------------------------------------------------------------
// Ragged kernel: any mix of column forms (dynamic shapes: multi-hot bags of
// variable length, scatter columns, passthrough, Sum(axis=1)).
//
// Same block shape as the dense kernel with R = 1 or 2 output rows per wave.
//   phase 0  column records -> LDS;
//   phase 1  one thread per (column, row) pair reads the pair's CSR range
//            [lo, lo+cnt) — the LDS-staged row-offset buffer — and a block-wide
//            scan of the counts (wave shuffle scan + per-wave totals) assigns
//            every bag a slice of the LDS id tile; then one thread per *id*
//            (its bag found by binary search over the scanned offsets) fetches
//            it and stores the final local row number (Bucketize, range check,
//            row shard: once per id instead of once per lane, all ids of the
//            block in one memory round trip);
//   phase 2  every lane walks its bag in LDS: 8 row numbers -> 8 independent
//            16-byte table reads in flight -> 8 adds in id order (sequential
//            fp32 order: deterministic, equal to TF-CPU's and the oracle's),
//            divides for mean (sum / count, cuda_emitter.cc:625, :903), and the
//            wave stores 1 KiB contiguous of the concat row.
// Bags longer than 64 ids, or bags that do not fit the 1536-entry tile, are
// walked from global memory by the lanes themselves (same arithmetic order).
// ---------------------------------------------------------------------------
template <int V, int R>
__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_ragged_kernel(const FcpLaunch L) {
  constexpr int RB = FCP_WAVES_PER_BLOCK * R; // rows per block, R per wave
  constexpr int NP = FCP_WAVE * RB;           // (column, row) pairs per block, at most
  constexpr int PT = NP / FCP_BLOCK_THREADS;  // pairs per thread (= R)
  constexpr int CAP = 1536;                   // staged row numbers per block
  constexpr int LONG_BAG = 64;
  __shared__ LdsCol s_col[FCP_WAVE];
  __shared__ int32_t s_lo[NP], s_cnt[NP];
  __shared__ int32_t s_offx[NP + 1];          // exclusive scan of the staged counts
  __shared__ int32_t s_ids[CAP];
  __shared__ int32_t s_wsum[PT * FCP_WAVES_PER_BLOCK];

  int bid = blockIdx.x;
  int g = 0;
  for (int k = 1; k < L.n_groups; ++k)
    if (bid >= L.groups[k].block_begin) g = k;
  const int rows = L.groups[g].rows;
  const int nslots = L.groups[g].nslots;
  const int nsp8 = L.groups[g].nsp8;
  const uint32_t *__restrict__ map = L.slot_map + L.groups[g].slot_map_off;
  bid -= L.groups[g].block_begin;
  const int xcd = bid & 7, j8 = bid >> 3;
  const int span = (j8 % nsp8) * 8 + xcd;
  const int tile = j8 / nsp8;
  const int q0 = span * FCP_WAVE;
  const int row_blk = tile * RB;
  if (q0 >= nslots || row_blk >= rows) return;

  const int tid = threadIdx.x;
  const int lane = tid & (FCP_WAVE - 1);
  const int wave = tid >> 6;
  const int q = q0 + lane;
  const uint32_t my_col = map[min(q, nslots - 1)];
  const uint32_t first_col = map[q0];
  const int ncols = (int)(map[min(q0 + FCP_WAVE - 1, nslots - 1)] - first_col) + 1;
  const int npairs = ncols * RB;
  const int world = L.shard_world, rank = L.shard_rank;

  // ---- phase 0: column records -> LDS --------------------------------------------
  if (tid < ncols) s_col[tid] = make_lds_col(L, L.cols[first_col + tid], L.dyn[first_col + tid]);
  __syncthreads();

  // ---- phase 1a: row ranges of the block's (column, row) pairs + scan -------------
  int want[PT];
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    const int p = h * FCP_BLOCK_THREADS + tid;
    int lo = 0, cnt = 0;
    if (p < npairs) {
      const int pj = p / RB, pr = p % RB;
      const int b = row_blk + pr;
      if (b < rows) {
        const unsigned form = FCP_F_FORM(s_col[pj].flags);
        if (form == FCP_FORM_GATHER) {
          lo = b;
          cnt = 1;
        } else if (form == FCP_FORM_SEGMENT_REDUCE || form == FCP_FORM_GATHER_SCATTER) {
          const int nnz = s_col[pj].nnz;
          const int32_t *__restrict__ csr = s_col[pj].csr;
          lo = min(max(csr[b], 0), nnz);
          const int hi = min(max(csr[b + 1], lo), nnz);
          cnt = hi - lo;
          if (form == FCP_FORM_GATHER_SCATTER && cnt > 0) { // the last id of the row wins
            lo = hi - 1;
            cnt = 1;
          }
        }
      }
      s_lo[p] = lo;
      s_cnt[p] = cnt;
    }
    want[h] = cnt <= LONG_BAG ? cnt : 0;
  }
  int incl[PT];
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    incl[h] = want[h];
#pragma unroll
    for (int d = 1; d < FCP_WAVE; d <<= 1) {
      const int up = __shfl_up(incl[h], d);
      if (lane >= d) incl[h] += up;
    }
    if (lane == FCP_WAVE - 1) s_wsum[h * FCP_WAVES_PER_BLOCK + wave] = incl[h];
  }
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int h = 0; h < PT; ++h) {
    int base = 0;
    for (int w = 0; w < h * FCP_WAVES_PER_BLOCK + wave; ++w) base += s_wsum[w];
    s_offx[h * FCP_BLOCK_THREADS + tid] = base + incl[h] - want[h];
  }
  for (int w = 0; w < PT * FCP_WAVES_PER_BLOCK; ++w) total += s_wsum[w];
  if (tid == 0) s_offx[NP] = total;
  __syncthreads();

  // ---- phase 1b: one thread per staged id -> final row number in LDS -----------------
  // (ids that do not fit the tile are walked from global memory in phase 2)
  for (int k = tid; k < min(total, CAP); k += FCP_BLOCK_THREADS) {
    int lo_p = 0, hi_p = NP; // last pair with s_offx[p] <= k
    while (hi_p - lo_p > 1) {
      const int mid = (lo_p + hi_p) >> 1;
      if (s_offx[mid] <= k) lo_p = mid; else hi_p = mid;
    }
    const int p = lo_p;
    if (s_offx[p] + s_cnt[p] <= CAP) {
      bool bad;
      const int64_t id = fetch_row_number(s_col[p / RB], s_lo[p] + (k - s_offx[p]), nullptr, rank, world, bad);
      if (bad && L.bad_ids) atomicAdd(L.bad_ids, 1ull);
      s_ids[k] = (int32_t)id;
    }
  }
  __syncthreads();
  if (q >= nslots) return;

  // ---- phase 2 --------------------------------------------------------------------
  const int j = (int)(my_col - first_col);
  const LdsCol &C = s_col[j];
  const unsigned form = FCP_F_FORM(C.flags);
  const float *table = C.table;
  const int dim = C.dim;
  const int e = q * V - C.out_off;
  const bool mean = FCP_F_COMBINER(C.flags) == FCP_COMBINER_MEAN && world == 1;

  for (int r = 0; r < R; ++r) {
    const int b = row_blk + wave * R + r;
    if (b >= rows) break;
    float *outp = reinterpret_cast<float *>(L.arena + C.out_base) + e + (int64_t)b * C.out_stride;
    const int p = j * RB + wave * R + r;
    const int plo = s_lo[p], pcnt = s_cnt[p];
    const int poff = (pcnt <= LONG_BAG && s_offx[p] + pcnt <= CAP) ? s_offx[p] : -1;
    VF<V> acc = vzero<V>();

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
    } else if (form != FCP_FORM_SEGMENT_REDUCE) {
      // GATHER / GATHER_SCATTER: a pure copy of one row (rows without ids stay zero)
      if (pcnt > 0) {
        int64_t id;
        if (poff >= 0) {
          id = s_ids[poff];
        } else {
          bool bad;
          id = fetch_row_number(C, plo, nullptr, rank, world, bad);
          if (bad && e == 0 && L.bad_ids) atomicAdd(L.bad_ids, 1ull);
        }
        if (id >= 0) acc = ld_table<V>(table + id * (int64_t)dim + e);
      }
    } else {
      for (int i = 0; i < pcnt; i += 8) {
        int64_t id[8];
        if (poff >= 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) id[k] = (i + k < pcnt) ? (int64_t)s_ids[poff + i + k] : -1;
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            id[k] = -1;
            if (i + k < pcnt) {
              bool bad;
              id[k] = fetch_row_number(C, plo + i + k, nullptr, rank, world, bad);
              if (bad && e == 0 && L.bad_ids) atomicAdd(L.bad_ids, 1ull);
            }
          }
        }
        VF<V> w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          w[k] = vzero<V>();
          if (id[k] >= 0) w[k] = ld_table<V>(table + id[k] * (int64_t)dim + e);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (id[k] >= 0) {
#pragma unroll
            for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] + w[k].v[t];
          }
      }
      if (mean && pcnt > 0) {
        const float fc = (float)pcnt; // sum / count
#pragma unroll
        for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] / fc;
      }
    }
    st_out<V>(outp, acc);
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

__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_segment_offsets_kernel(const FcpSegLaunch L) {
  const int c = L.seg_cols[blockIdx.y];
  const FcpColDyn cd = L.dyn[c];
  const int nnz = cd.nnz;
  const int64_t base = (int64_t)blockIdx.x * FCP_BLOCK_THREADS;
  if (base > nnz) return;
  const FcpColStatic cs = L.cols[c];
  const unsigned segkind = FCP_F_SEGKIND(cs.flags);
  const int stride = cs.seg_stride;
  const int64_t rows = cd.rows;
  const char *seg = L.blob + cd.seg_off;
  const int64_t i = base + threadIdx.x;
  const int lane = threadIdx.x & (FCP_WAVE - 1);
  const bool active = i <= nnz;

  int64_t cur = rows;
  if (active && i < nnz) cur = load_seg(seg, segkind, stride, i);
  if (cur > rows) cur = rows;
  int64_t prev = __shfl_up(cur, 1);
  if (lane == 0) {
    prev = -1;
    if (active && i > 0) {
      prev = load_seg(seg, segkind, stride, i - 1);
      if (prev > rows) prev = rows;
    }
  }
  const bool boundary = active && cur > prev;
  if (__ballot(boundary) == 0ull) return;
  if (!boundary) return;
  int32_t *csr = reinterpret_cast<int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base;
  for (int64_t id = prev + 1 < 0 ? 0 : prev + 1; id <= cur; ++id) csr[id] = (int32_t)i;
}

// ---------------------------------------------------------------------------
// ConcatOutputs (reference layout pass, concat_outputs_op_gpu.cu.cc:85-131):
// out[p, off_k + e] = in_k[p*dim_k + e].  Only used with FCP_LAYOUT_PER_COLUMN;
// the fused kernel above writes the concat layout directly.
// ---------------------------------------------------------------------------
#define FCP_CONCAT_CHUNK 192
struct FcpConcatArgs {
  const float *in[FCP_CONCAT_CHUNK];
  int32_t off[FCP_CONCAT_CHUNK];
  int32_t dim[FCP_CONCAT_CHUNK];
  float *out;
  int64_t prefix;
  int32_t width;
  int32_t n;
};

__global__ void __launch_bounds__(FCP_BLOCK_THREADS) fcp_concat_outputs_kernel(const FcpConcatArgs A) {
  const int k = blockIdx.y;
  const int dim = A.dim[k];
  const int off = A.off[k];
  const float *__restrict__ in = A.in[k];
  const int64_t total = A.prefix * dim;
  for (int64_t i = (int64_t)blockIdx.x * FCP_BLOCK_THREADS + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * FCP_BLOCK_THREADS) {
    const int64_t p = i / dim;
    const int e = (int)(i - p * dim);
    A.out[p * A.width + off + e] = in[i];
  }
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
  const int q = blockIdx.x * FCP_BLOCK_THREADS + threadIdx.x;
  const int64_t bl = blockIdx.y;
  if (q >= nslots || bl >= row_count) return;
  const int64_t W = (int64_t)nslots * V;
  const uint32_t c = L.slot_map[L.groups[g].slot_map_off + q];
  const FcpColStatic cs = L.cols[c];
  const FcpColDyn cd = L.dyn[c];
  VF<V> acc = vzero<V>();
  for (int w = 0; w < world; ++w) {
    const VF<V> x = *reinterpret_cast<const VF<V> *>(partials + ((int64_t)w * row_count + bl) * W + (int64_t)q * V);
#pragma unroll
    for (int t = 0; t < V; ++t) acc.v[t] = acc.v[t] + x.v[t];
  }
  if (FCP_F_FORM(cs.flags) == FCP_FORM_SEGMENT_REDUCE && FCP_F_COMBINER(cs.flags) == FCP_COMBINER_MEAN) {
    const unsigned segkind = FCP_F_SEGKIND(cs.flags);
    const int32_t *csr = segkind == FCP_SEG_CSR_I32
                             ? reinterpret_cast<const int32_t *>(L.blob + cd.seg_off)
                             : reinterpret_cast<const int32_t *>(L.arena + L.csr_arena_off) + cd.csr_base;
    const int64_t b = row_begin + bl;
    int lo = csr[b], hi = csr[b + 1];
    lo = min(max(lo, 0), cd.nnz);
    hi = min(max(hi, lo), cd.nnz);
    if (hi > lo) {
      const float fc = (float)(hi - lo);
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

// ------------------------------- launchers ---------------------------------

#define FCP_LAUNCH_DENSE(VV, RR) \
  hipLaunchKernelGGL((fcp_dense_kernel<VV, RR>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), 0, s, L)
#define FCP_LAUNCH_GENERIC(VV)                                                                          \
  do {                                                                                                  \
    if (L.rows_per_wave >= 2)                                                                           \
      hipLaunchKernelGGL((fcp_ragged_kernel<VV, 2>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), 0, s, L); \
    else                                                                                                \
      hipLaunchKernelGGL((fcp_ragged_kernel<VV, 1>), dim3(grid_blocks), dim3(FCP_BLOCK_THREADS), 0, s, L); \
  } while (0)

int fcp_launch_fused(const FcpLaunch &L, int vec, bool dense_only, int grid_blocks, ihipStream_t *s) {
  if (grid_blocks <= 0) return 0;
  if (dense_only) {
    const int R = L.rows_per_wave;
#define FCP_DENSE_R(VV)                         \
  switch (R) {                                  \
  case 8: FCP_LAUNCH_DENSE(VV, 8); break;       \
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
      FCP_LAUNCH_GENERIC(4);
    } else if (vec == 2) {
      FCP_LAUNCH_GENERIC(2);
    } else {
      FCP_LAUNCH_GENERIC(1);
    }
  }
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

int fcp_launch_segment_offsets(const FcpSegLaunch &L, int n_seg_cols, int max_nnz, ihipStream_t *s) {
  if (n_seg_cols <= 0) return 0;
  const int gx = (max_nnz + 1 + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS;
  hipLaunchKernelGGL(fcp_segment_offsets_kernel, dim3(gx, n_seg_cols), dim3(FCP_BLOCK_THREADS), 0, s, L);
  return (int)hipGetLastError();
}

int fcp_launch_concat_outputs(const void *const *inputs, const int32_t *dims, int32_t n,
                              int64_t prefix, int32_t width, int32_t first_off, void *out,
                              ihipStream_t *s) {
  int32_t off = first_off;
  for (int32_t begin = 0; begin < n; begin += FCP_CONCAT_CHUNK) {
    FcpConcatArgs A;
    const int32_t m = (n - begin) < FCP_CONCAT_CHUNK ? (n - begin) : FCP_CONCAT_CHUNK;
    int32_t max_dim = 1;
    for (int32_t k = 0; k < m; ++k) {
      A.in[k] = static_cast<const float *>(inputs[begin + k]);
      A.dim[k] = dims[begin + k];
      A.off[k] = off;
      off += dims[begin + k];
      if (dims[begin + k] > max_dim) max_dim = dims[begin + k];
    }
    A.out = static_cast<float *>(out);
    A.prefix = prefix;
    A.width = width;
    A.n = m;
    int64_t gx = (prefix * max_dim + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS;
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(fcp_concat_outputs_kernel, dim3((unsigned)gx, m), dim3(FCP_BLOCK_THREADS), 0, s, A);
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
  dim3 grid((nslots + FCP_BLOCK_THREADS - 1) / FCP_BLOCK_THREADS, (unsigned)row_count);
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
