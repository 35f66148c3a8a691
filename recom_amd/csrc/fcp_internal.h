// fcp_internal.h — device-visible plan records shared by the host side
// (fcp_plan.hip, fcp_process.hip, ...: fcp_host.h) and the kernels (fcp_kernels.hip).  gfx950 only.
//
// What the reference passes to its generated kernel as one flat `KnlArgs`
// struct of per-model pointers and ints (graph_optimizers/cuda_emitter.cc
// :2059-2087, uploaded H2D on every call :2182-2217) is split here into
//   * FcpColStatic[]  — per-column facts fixed at plan creation (+ table bind),
//   * FcpColDyn[]     — per-column facts that follow the request's shapes,
//   * FcpLaunch       — a few scalars passed by value as the kernel argument.
// Only FcpColDyn is re-uploaded, and only when the request's shapes differ from
// the previous request's (steady-state serving uploads nothing).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef FCP_MAX_GROUPS
#define FCP_MAX_GROUPS 16
#endif
#define FCP_BLOCK_THREADS 256
#define FCP_WAVES_PER_BLOCK 4
#define FCP_WAVE 64

// flags layout of FcpColStatic::flags
#define FCP_F_FORM(f) ((f) & 0xFu)
#define FCP_F_COMBINER(f) (((f) >> 4) & 0xFu)
#define FCP_F_IDSRC(f) (((f) >> 8) & 0xFu)
#define FCP_F_SEGKIND(f) (((f) >> 12) & 0xFu)
#define FCP_F_PACK(form, comb, idsrc, segkind) \
  ((uint32_t)(form) | ((uint32_t)(comb) << 4) | ((uint32_t)(idsrc) << 8) | ((uint32_t)(segkind) << 12))

struct FcpColStatic {      // 64 bytes: one record per cache line
  const float *table;      // device address of the table (bound from input_ptrs)
  const float *boundaries; // device const buffer (CreateConstBuffers, :2260-2301)
  int64_t vocab;           // global vocabulary (rows of the unsharded table)
  int32_t dim;
  int32_t out_off;         // element offset of the column in its concat row
  uint32_t flags;          // FCP_F_PACK(...)
  int32_t n_boundaries;
  int32_t seg_stride;
  // Evenly spaced boundaries (the reference's bucketized columns: 0, 5, ..., 495, microbenchmark.py:46):
  // bucket guess = floor((x - bnd_b0) * bnd_inv) + 1, verified against the two neighbouring boundaries
  // (exact for any array; see bucketize_fast).  bnd_inv == 0: no guess, plain binary search.
  // bnd_step != 0: every boundary is REPRODUCIBLE as fma(i, bnd_step, bnd_b0) (checked bit for bit at plan
  // creation), so the kernels never read the array (bucketize_arith).
  float bnd_b0, bnd_inv, bnd_step;
  // id transform, 0 = none: FCP_XFORM_* in the low 2 bits, number of closed intervals in bits 2..30, bit 31 =
  // hash the id into buckets first; the parameters live in FcpLaunch::xforms[column] and are only read by
  // columns that have a transform
  uint32_t xform;
  int32_t bnd_off;         // always -1 in memory; in the LDS copy: where a block staged the boundaries (or -1)
};

struct FcpXform {          // 48 bytes per column (allocated only for plans with id transforms)
  int64_t lo0, hi0;        // first closed interval
  int64_t sub;             // FCP_XFORM_SELECT: what an id outside the intervals becomes
  const int64_t *extra;    // intervals 1.. as (lo, hi) pairs in the const buffer
  int64_t hash_buckets;    // bit 31 of xform: id = Fingerprint64(decimal(id)) % hash_buckets
  int64_t pad_;
};
#define FCP_XFORM_HASH_BIT 0x80000000u

struct FcpColDyn {         // 48 bytes
  int64_t ids_off;         // byte offset of the id / value / payload stream in the blob
  int64_t seg_off;         // byte offset of seg ids / CSR offsets in the blob
  int64_t out_base;        // byte offset in the arena of element (0,0) of this column
  int32_t out_stride;      // row stride of the output in elements
  int32_t nnz;             // number of ids
  int32_t csr_base;        // int32 index into the arena CSR scratch (seg-id columns), or -1
  int32_t inner;           // BatchColReduction: rows reduced per output row
  int32_t rows;            // output rows (prefix size) of this column
  int32_t seg_sym;         // segment-id map: the request's value of the symbol one factor is multiplied by (else 1)
};

// Segment ids computed from several index coordinates (fcp_column_ext_t::seg_map_*: a SparseReshape folded into the
// index expression, cuda_emitter.cc:1874-1916); per column, allocated only for plans that have one.  Read by the
// segment-offset pre-pass only.
struct FcpSegMap {         // 48 bytes
  int64_t mul[4];
  int64_t div;
  int32_t n;               // 0: plain segment ids
  int32_t sym_slot;        // which factor takes FcpColDyn::seg_sym: 0..3 = mul[slot], 4 = div, -1 = none
};

struct FcpGroupLaunch {
  int32_t rows;            // prefix size of the group
  int32_t nslots;          // concat row width in V-element slots
  int32_t nsp8;            // ceil(nlist / 8): listed spans per XCD; or -nlist when nlist < 8 (no XCD padding)
  int32_t block_begin;     // first block of this group in the grid
  int32_t slot_map_off;    // offset of the group's slot map
  int32_t span_list_off;   // offset into FcpLaunch::span_list of the spans this launch covers, or -1: all spans
  int32_t nlist;           // number of spans this launch covers in this group
  int32_t csr_reg_stride;  // groups[0] only (regular CSR needs a one-group plan): see FcpLaunch::csr_reg; 0 = not regular
};

// Field order matters: everything a block of a one-group plan reads lies in the first two 64-byte
// lines of the argument block (the kernels fetch both with one batch of scalar loads and wait once).
struct FcpLaunch {
  const uint32_t *slot_map;  // slot -> column index
  const uint32_t *span_list; // span indices per (group, kernel kind): hybrid dense / ragged dispatch
  const FcpColStatic *cols;
  const FcpColDyn *dyn;
  const char *blob;
  char *arena;
  unsigned long long *bad_ids; // nullable
  int64_t csr_arena_off;       // byte offset of the CSR scratch inside the arena
  int32_t n_groups;
  int32_t rows_per_wave;
  int32_t shard_rank, shard_world;
  int32_t seg_search;          // 1: blocks find their rows' ranges in the sorted segment ids themselves (no pre-pass)
  int32_t store_through;       // bit 0: output stores are write-through (`sc1 nt`): outputs larger than the L2s (host decides);
                               // bit 1: some table (shard) of the plan has >= 2^32 - 3 slots: the dense body parks rows and
                               // multiplies in 64 bits instead of parking pre-scaled 32-bit slot offsets
  // (r6) Regular CSR: when the row-offset arrays of ALL column positions lie groups[0].csr_reg_stride int32 apart — the arena
  // scratch the pre-pass fills, laid out by column position (fcp_plan::csr_by_pos), or CSR inputs that arrive that way in
  // the blob — row b of the column at concat position c is csr_reg[c * stride + b]: the ragged body asks for its ranges
  // TOGETHER with the column records instead of behind them (one dependent round trip less in every block's front).
  // Stride 0: not regular, the ranges come through the records (LdsCol::csr).  (Pointer and stride sit in the first two
  // 64-byte lines of the argument block, like everything a block of a one-group plan reads.)
  const int32_t *csr_reg;
  FcpGroupLaunch groups[FCP_MAX_GROUPS];
  const FcpXform *xforms;      // per column (concat order), or null: no column has an id transform
  const float *zeros;          // 256 zero bytes (plan-owned): what a skipped id of a bag reads
  unsigned long long *stamps;  // diagnostic builds (-DFCP_STAMPS) only: 8 timestamps per block
};

// Segment-offset pre-pass (ComputeSegmentOffsets, cuda_emitter.cc:768-818)
struct FcpSegLaunch {
  const int32_t *seg_cols;   // indices of the columns whose seg_kind is IDS_*
  const FcpColStatic *cols;
  const FcpColDyn *dyn;
  const char *blob;
  char *arena;
  int64_t csr_arena_off;     // byte offset of the CSR scratch in the arena
  unsigned long long *bad_ids; // nullable: FCP_FLAG_COUNT_BAD_IDS also counts unsorted segment ids here
  const FcpXform *xforms;    // per column, or null (id filter of any-order scatter columns)
  const FcpSegMap *segmaps;  // per column, or null (segment ids from several index coordinates)
  int32_t skip_inverse;      // 1: leave the any-order scatter columns alone (fcp_shard_finalize only wants row lengths)
};

// ---- launchers implemented in fcp_kernels.hip --------------------------------
struct ihipStream_t;
// the next fused / hybrid launch of THIS thread carries `event` (hipEvent_t) as its stop event; pending = not taken yet
void fcp_set_stop_event(void *event);
bool fcp_stop_event_pending();
// the next fused / hybrid launch of THIS thread is an any-order launch (no barrier bit); cleared by the launch
void fcp_set_any_order(bool on);
int fcp_launch_fused(const FcpLaunch &L, int vec, bool dense_kernel, int grid_blocks, ihipStream_t *s);
int fcp_launch_hybrid(const FcpLaunch &Ldense, int dense_blocks, const FcpLaunch &Lragged, int ragged_blocks, int vec,
                      ihipStream_t *s);
int fcp_launch_upload(const void *host_mapped_src, void *dst, size_t bytes, ihipStream_t *s);
// bytes (a multiple of 4) from host-mapped pinned memory to device memory by a kernel; both 4-byte aligned and equally
// misaligned against 16 bytes
int fcp_launch_h2d_copy(const void *host_mapped_src, void *dst, size_t bytes, ihipStream_t *s);
int fcp_launch_segment_offsets(const FcpSegLaunch &L, int n_seg_cols, int max_nnz, ihipStream_t *s, bool any_order = false);
// col_offsets: destination column of every input, or NULL = side by side starting at first_off;
// in_strides: row stride of every input in floats, or NULL = contiguous [prefix, dims[k]] inputs
int fcp_launch_concat_outputs(const void *const *inputs, const int32_t *dims, const int32_t *col_offsets, const int32_t *in_strides,
                              int32_t n, int64_t prefix, int32_t width, int32_t first_off, void *out,
                              ihipStream_t *s);
int fcp_launch_shard_finalize(const FcpLaunch &L, int group, const float *partials, int world,
                              int64_t row_begin, int64_t row_count, float *out, int vec,
                              ihipStream_t *s);
