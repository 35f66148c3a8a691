/*
 * fcp_hip.h — C ABI of libfcp_hip.so: the MI355X-native fused feature-column
 * (embedding-column) inference path.
 *
 * This is the drop-in boundary for the three custom ops of the reference
 * (AlibabaResearch/recom, paths relative to the reference root):
 *
 *   Addons>ConcatInputs                      custom_ops/concat_inputs/concat_inputs_ops.cc:42-88
 *   Addons>FeatureColumnProcess[WithSymbols] custom_ops/feature_column_process/feature_column_process_op_gpu.cu.cc:65-175
 *   Addons>ConcatOutputs[NoHost]             custom_ops/concat_outputs/concat_outputs_op_gpu.cu.cc:180-288
 *
 * and for the `extern "C"` entry points those ops dlsym from the JIT-compiled
 * artifact today:
 *
 *   CreateConstBuffers      graph_optimizers/cuda_emitter.cc:2260-2301
 *   ProcessFeatureColumns   graph_optimizers/cuda_emitter.cc:2303-2494 (decl feature_column_process_ops.h:37-43)
 *   ConcatOutputs           custom_ops/concat_outputs/concat_outputs_op_gpu.cu.cc:133-140 (decl concat_outputs_ops.h:34)
 *
 * The reference entry points are C-named but C++-typed (std::vector,
 * std::function, references).  Here everything is plain C: pointers, sizes,
 * int status codes, function-pointer allocators.  No torch / TF / HIP types
 * appear in any signature (a HIP stream is passed as an opaque pointer).
 *
 * What the reference generates as CUDA text per model (one `struct FCi` per
 * feature column, cuda_emitter.cc:1976-2055) is described here by a static
 * *column plan* (fcp_column_desc_t[]) that pre-compiled gfx950 kernels
 * interpret.  No code generation and no runtime compiler are involved.
 *
 * Threading: a plan is immutable after creation; fcp_process_feature_columns
 * and fcp_concat_outputs are re-entrant on a shared plan (the reference ops
 * keep no per-call state in members either, SURVEY.md §8b "Threading").
 * Concurrent calls (serve workers, one stream each) serialise only on the
 * plan's descriptor-slot bookkeeping; shape evaluation, the allocator callback
 * and the kernel launches of different callers overlap.  No entry point
 * synchronises the stream in the steady state: work is enqueued and the call
 * returns (the reference blocks three times per request).  A
 * call may wait for an EARLIER request's kernel when more than 8 requests with
 * new shapes are in flight (back-pressure on the descriptor slots).
 */
#ifndef FCP_HIP_H_
#define FCP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCP_ABI_VERSION 2 /* 2: id transforms (fcp_column_desc_t::xform_*), external slots, placement gate; additions since keep every v2 struct as it was */

/* ---- status codes (reference: void returns + CubDebugExit/exit(1)) ------ */
enum {
  FCP_OK = 0,
  FCP_ERR_INVALID_ARGUMENT = 1, /* bad descriptor / null pointer / bad enum  */
  FCP_ERR_SHAPE_MISMATCH = 2,   /* run-time shapes inconsistent with the plan */
  FCP_ERR_ALLOC = 3,            /* an allocator callback returned NULL        */
  FCP_ERR_HIP = 4,              /* a HIP runtime call failed                  */
  FCP_ERR_UNSUPPORTED = 5,      /* valid but not implemented for this plan    */
  FCP_ERR_NO_DEVICE = 6         /* no gfx950 device / code object not loadable*/
};

/* ---- column forms: the canonical per-column rewrites of the reference ---- */
/* (lookup_optimizer.cc:157-440; emitter dispatch cuda_emitter.cc:1096-1152)  */
enum {
  /* GatherV2(table, ids): out[i,:] = W[ids[i],:]
   * cuda_emitter.cc:250-293 (GatherRowsToGlbMem), driver :1246-1330 */
  FCP_FORM_GATHER = 1,
  /* SparseSegment{Sum,Mean}[WithNumSegments](table, ids, seg_ids, B)
   * cuda_emitter.cc:402-501, :564-661 (dim<=20) and :768-962 (dim>20) */
  FCP_FORM_SEGMENT_REDUCE = 2,
  /* ScatterNd(rows, GatherV2(table, ids), [B,dim]): zero-fill, then
   * out[rows[i],:] = W[ids[i],:]   cuda_emitter.cc:296-345, :1332-1442
   * The row ids (seg_kind FCP_SEG_IDS_*) arrive in ANY order, as the reference's
   * kernel takes them; of several writes to one row the LAST one stays (the order
   * of a sequential scatter; the reference races inside a 64-id tile), rows
   * outside [0, B) are dropped (TF's ScatterNd on a GPU) and, with
   * FCP_FLAG_COUNT_BAD_IDS, counted.  The segment-offset pre-pass builds the
   * row -> position map for such columns.  With seg_kind FCP_SEG_CSR_I32 the ids
   * are grouped by row and the last id of a row's range wins.  An id filter
   * (xform) drops ids before the scatter: the last id it KEEPS wins. */
  FCP_FORM_GATHER_SCATTER = 3,
  /* A tensor of the ConcatInputs blob copied straight into its concat slot
   * (ConcatOutputs `host_inputs`, concat_outputs_op_gpu.cu.cc:186-216) */
  FCP_FORM_PASSTHROUGH = 4,
  /* Sum(x, axis=1) on a rank-3 [B,R,C] tensor of the blob
   * cuda_emitter.cc:1180-1244 (BatchColReduction) */
  FCP_FORM_BATCH_COL_REDUCTION = 5,
  /* A concat slot somebody else fills: an Addons>ConcatOutputs `host_inputs`
   * tensor (concat_outputs_op_gpu.cu.cc:186-216; Rewrite wires every non-FC
   * input of the ConcatV2 that way, cuda_emitter.cc:2594-2611).  The plan only
   * reserves [rows, dim] at the slot's concat offset — the kernels never touch
   * it — and fcp_concat_outputs_host copies the host tensor there.  It is NOT an
   * output of FeatureColumnProcess (fcp_plan_output_columns skips it), has no
   * inputs (ids_input = table_input = -1) and takes its row count from its
   * group (FCP_ROWS_FROM_GROUP).  FCP_LAYOUT_CONCAT only. */
  FCP_FORM_EXTERNAL = 6
};

enum { FCP_COMBINER_NONE = 0, FCP_COMBINER_SUM = 1, FCP_COMBINER_MEAN = 2 };

/* How the id stream of a column is stored in the blob (what EmitInputInline
 * folds into the index expression, cuda_emitter.cc:1769-1949). */
enum {
  FCP_IDS_I32 = 0,           /* int32 ids                                    */
  FCP_IDS_I64 = 1,           /* int64 ids (reference truncates to int, :270) */
  FCP_IDS_F32_BUCKETIZE = 2  /* float32 values -> Bucketize(boundaries), :233-247, :1798-1835 */
};

/* Id transform, applied to every id after it has been decoded (raw integer or
 * Bucketize result) and before the lookup: what the CPU ops the
 * reference's PreLookupOptimizer puts in front of the lookup compute
 * (graph_optimizers/pre_lookup_optimizer.cc:596-654).  The interval set is a list of
 * CLOSED integer intervals [lo_i, hi_i] — the ops' left_boundaries / right_boundaries
 * attrs.  An id is "in" when lo_i <= id && id <= hi_i for some i.  (The reference's
 * kernels test `x >= l || x <= r`, select_value_ops.cc:35-41 — true for every x, so
 * they filter nothing; this is the intended test, SURVEY.md App. A.) */
enum {
  FCP_XFORM_NONE = 0,
  /* Addons>SelectValue (custom_ops/select_value/select_value_ops.cc:33-56):
   * id' = in ? id : substitute */
  FCP_XFORM_SELECT = 1,
  /* Addons>GatherIndiceValue / Addons>GatherValueGenIndice
   * (gather_indice_value_ops.cc:33-78, gather_value_gen_indice_ops.cc:33-67): ids that
   * are not "in" are DROPPED: they add nothing to their row and do not count in a
   * mean; a row left without ids is zeros.  (Row-sharded plans: fcp_shard_finalize
   * re-reads the row's ids and divides a mean by the kept count.) */
  FCP_XFORM_FILTER = 2
};

/* How segment membership of the id stream is given. */
enum {
  FCP_SEG_NONE = 0,     /* forms 1,4,5                                        */
  FCP_SEG_IDS_I32 = 1,  /* sorted segment / row ids, element stride seg_stride */
  FCP_SEG_IDS_I64 = 2,  /* e.g. SparseTensor indices[nnz,2] int64 with stride 2
                           (inlined StridedSlice [:,0], cuda_emitter.cc:1836-1873) */
  FCP_SEG_CSR_I32 = 3   /* CSR offsets int32[B+1] (what ComputeSegmentOffsets,
                           cuda_emitter.cc:768-818, produces on the fly)      */
};

/* Where the number of output rows ("prefix size") of a column comes from. */
enum {
  FCP_ROWS_FROM_IDS = 0,    /* rows = number of elements of the ids tensor (form 1) */
  FCP_ROWS_FROM_SYMBOL = 1, /* rows = symbols[rows_arg] (Addons>ShapeConstruct
                               result, cuda_emitter.cc:2438-2455)             */
  FCP_ROWS_FROM_INPUT_DIM0 = 2, /* rows = shape[0] of host input rows_arg      */
  FCP_ROWS_FROM_GROUP = 3   /* FCP_FORM_EXTERNAL: rows of the other columns of its concat group */
};

/* Output arena layouts. */
enum {
  /* One [rows, sum(dim)] row-major matrix per concat group; every column is
   * written directly at its concat offset: ConcatOutputsKnl
   * (concat_outputs_op_gpu.cu.cc:85-131) is fused away. */
  FCP_LAYOUT_CONCAT = 0,
  /* The reference arena: one contiguous [rows, dim] buffer per column, each
   * 128-byte aligned (alignmem, cuda_emitter.cc:967-969, :2151-2179);
   * fcp_concat_outputs then performs the reference's second pass. */
  FCP_LAYOUT_PER_COLUMN = 1
};

enum {
  FCP_FLAG_NONE = 0,
  /* Count ids outside [0, vocab) into the plan's device error counter.  Such
   * rows always read as zeros (TF-GPU GatherV2 semantics); the reference
   * reads out of bounds. */
  FCP_FLAG_COUNT_BAD_IDS = 1u << 0
};
/* A plan without device resources: layout / arena / table-byte queries and plan-file checks on a machine
 * without a GPU (offline graph tooling).  Anything that computes returns FCP_ERR_NO_DEVICE — there is no
 * CPU fallback.  (A macro: the value does not fit an int enumerator.) */
#define FCP_FLAG_HOST_ONLY 0x80000000u

typedef struct fcp_column_desc {
  int32_t form;         /* FCP_FORM_*                                         */
  int32_t combiner;     /* FCP_COMBINER_* (form 2)                            */
  int32_t dim;          /* embedding width (table.shape[1]); form 4/5: width  */
  int32_t id_source;    /* FCP_IDS_*                                          */
  int64_t vocab;        /* table.shape[0]                                     */
  int32_t table_input;  /* index into fcp_process_args_t.input_ptrs           */
  int32_t ids_input;    /* host-input (blob tensor) index of ids / values /
                           passthrough payload                                */
  int32_t seg_input;    /* host-input index of seg ids / CSR offsets, or -1   */
  int32_t seg_kind;     /* FCP_SEG_*                                          */
  int32_t seg_stride;   /* element stride between consecutive seg ids (>=1)   */
  int32_t rows_source;  /* FCP_ROWS_*                                         */
  int32_t rows_arg;     /* symbol index / host-input index                    */
  int32_t n_boundaries; /* FCP_IDS_F32_BUCKETIZE: number of boundaries        */
  const float *boundaries; /* host pointer, copied at plan creation           */
  int32_t concat_group; /* which ConcatV2 this column feeds, 0..n_groups-1    */
  int32_t concat_slot;  /* position inside the group (ConcatOutputs
                           device_concat_indices / host_concat_indices)       */
  int32_t xform_mode;   /* FCP_XFORM_* (lookup forms only)                    */
  int32_t xform_n;      /* number of closed intervals                         */
  const int64_t *xform_lo; /* host int64[xform_n], copied at plan creation    */
  const int64_t *xform_hi;
  int64_t xform_substitute; /* FCP_XFORM_SELECT                               */
  /* > 0: categorical_column_with_hash_bucket over INTEGER ids, on the device:
   * id' = Fingerprint64(decimal string of id) % hash_buckets — what TensorFlow's
   * AsString -> StringToHashBucketFast pair computes (TF 2.6.2
   * core/kernels/string_to_hash_bucket_fast_op.h: Fingerprint64 = FarmHash
   * farmhashna::Hash64, third_party farmhash commit 816a4ae6).  Applied FIRST,
   * before the interval transform.  String features are hashed on the CPU. */
  int64_t hash_buckets;
} fcp_column_desc_t;

typedef struct fcp_plan_desc {
  int32_t abi_version;            /* FCP_ABI_VERSION                          */
  int32_t n_columns;
  const fcp_column_desc_t *columns;
  int32_t n_host_inputs;          /* ConcatInputs attr T / ranks sizes        */
  const int32_t *host_input_ranks;      /* attr "ranks"                       */
  const int32_t *host_input_elem_sizes; /* DataTypeSize of attr "T" entries   */
  int32_t n_device_inputs;        /* FeatureColumnProcess `inputs` (tables)   */
  int32_t n_groups;               /* number of concat groups                  */
  int32_t n_symbols;              /* length of the `symbols` host tensor      */
  int32_t layout;                 /* FCP_LAYOUT_*                             */
  int32_t device;                 /* HIP device ordinal                       */
  int32_t shard_rank;             /* row sharding: this GPU owns ids with     */
  int32_t shard_world;            /*   id % shard_world == shard_rank, local
                                       row id / shard_world. 1 = unsharded.   */
  uint32_t flags;                 /* FCP_FLAG_*                               */
} fcp_plan_desc_t;

typedef struct fcp_plan fcp_plan_t; /* opaque; owns const buffers on device  */

/* Allocator callback: return device memory of `bytes` bytes (or NULL).
 * Reference: std::function<void(void**,int)> malloc_temp / malloc_buff
 * (feature_column_process_op_gpu.cu.cc:99-111).  The library never frees
 * caller memory.  malloc_buff is called at most once per process call
 * (it maps to allocate_output(2)). */
typedef void *(*fcp_alloc_fn)(void *ctx, size_t bytes);

typedef struct fcp_process_args {
  const void *concated_inputs;     /* device: ConcatInputs blob (input 0)     */
  int64_t concated_bytes;
  const int32_t *concated_offsets; /* host int32[n_host_inputs]   (input 1)   */
  const int32_t *concated_shapes;  /* host int32[sum ranks]       (input 2)   */
  const void *const *input_ptrs;   /* host array of device table pointers     */
  const int32_t *input_shapes;     /* host int32[2*n_device_inputs] or NULL   */
  const int32_t *symbols;          /* host int32[n_symbols] or NULL           */
  void *stream;                    /* hipStream_t                             */
  fcp_alloc_fn malloc_temp;        /* may be NULL: plan-owned scratch is used */
  void *malloc_temp_ctx;
  fcp_alloc_fn malloc_buff;        /* required: allocates the output arena    */
  void *malloc_buff_ctx;
} fcp_process_args_t;

typedef struct fcp_process_result {
  void **output_ptrs;          /* host void*[n_columns]: device address of
                                  element (0,0) of each column's output       */
  int32_t *output_shapes;      /* host int32[2*n_columns]: rows, dim          */
  int64_t *output_row_strides; /* host int64[n_columns]: row stride, elements */
  void **group_ptrs;           /* host void*[n_groups]: concat matrices
                                  (FCP_LAYOUT_CONCAT) or NULLs                */
  int32_t *group_shapes;       /* host int32[2*n_groups]: rows, sum(dim)      */
  void *buffer;                /* the arena returned by malloc_buff           */
  int64_t buffer_bytes;
} fcp_process_result_t;

/* Plain view of a host tensor for fcp_concat_inputs. */
typedef struct fcp_host_tensor {
  const void *data;
  int32_t elem_size;
  int32_t rank;
  const int64_t *dims;
} fcp_host_tensor_t;

/* ---- library ------------------------------------------------------------- */
int fcp_abi_version(void);
const char *fcp_status_string(int status);
/* Last HIP error string seen by this thread ("" if none). */
const char *fcp_last_error(void);

/* ---- Addons>ConcatInputs (concat_inputs_ops.cc:42-77), host only --------- */
/* Sizes of the three outputs for these inputs. */
int fcp_concat_inputs_sizes(const fcp_host_tensor_t *inputs, int32_t n_inputs,
                            int64_t *blob_bytes, int32_t *rank_sum);
/* Pack: blob = byte concatenation, offsets[i] = byte offset of input i
 * (int32, as the reference), shapes = all dims in order. */
int fcp_concat_inputs(const fcp_host_tensor_t *inputs, int32_t n_inputs,
                      void *blob, int64_t blob_capacity, int32_t *offsets,
                      int32_t *shapes);

/* ---- plan: replaces code generation + CreateConstBuffers ------------------ */
int fcp_plan_create(const fcp_plan_desc_t *desc, fcp_plan_t **plan);

/* Per-column extensions (the v2 structs above stay as they are; a column without extensions is all zeros).
 *
 * Segment ids that are a FUNCTION OF SEVERAL INDEX COORDINATES — a SparseReshape between the SparseTensor and
 * the lookup, which the reference folds into the index expression of the generated code
 * (EmitInputInline, cuda_emitter.cc:1874-1916: flat index from the input shape, then / and % by the output
 * shape).  For the row coordinate of the reshaped tensor that is
 *     seg(i) = (sum_{k < seg_map_n} idx[i * seg_stride + k] * seg_map_mul[k]) / seg_map_div
 * over the first seg_map_n coordinates of element i (seg_stride = rank of the ORIGINAL indices matrix,
 * seg_kind FCP_SEG_IDS_I32/I64, form FCP_FORM_SEGMENT_REDUCE).  Factors are >= 1; one of them may additionally
 * be multiplied by a request's symbol (a dense_shape entry that is only known per request: [B, T, L] ->
 * [B*T, L] gives seg = idx0 * T + idx1): seg_map_sym >= 0 names the symbol, seg_map_sym_slot the factor
 * (0..3 = seg_map_mul[slot], 4 = seg_map_div).  seg_map_n = 0: plain segment ids, seg(i) = idx[i * seg_stride].
 * Lexicographically sorted indices (TF's SparseTensor contract) give non-decreasing seg ids. */
#define FCP_SEG_MAP_MAX 4
typedef struct fcp_column_ext {
  int32_t seg_map_n;        /* 0 = none, else 1..FCP_SEG_MAP_MAX coordinates  */
  int32_t seg_map_sym;      /* symbol index, or -1                            */
  int32_t seg_map_sym_slot; /* 0..3: seg_map_mul[slot]; 4: seg_map_div        */
  int32_t reserved0;
  int64_t seg_map_mul[FCP_SEG_MAP_MAX];
  int64_t seg_map_div;
  int64_t reserved1[2];
} fcp_column_ext_t;
/* `ext`: NULL, or one record per column of `desc`. */
int fcp_plan_create_ex(const fcp_plan_desc_t *desc, const fcp_column_ext_t *ext, fcp_plan_t **plan);
/* The same from a column-plan FILE — what the `dlpath` attr of
 * Addons>FeatureColumnProcess names in this build (the reference dlopen()s a
 * JIT-compiled .so there, feature_column_process_op_gpu.cu.cc:49-62).  Text
 * format, written by recom_amd.plan_io.save_plan / `python -m recom_amd.graph`:
 *   fcp_plan 1 / layout L / groups G symbols S device_inputs D / host_inputs N,
 *   N lines "rank elem_size" / columns C, C lines "form combiner dim id_source
 *   vocab table_input ids_input seg_input seg_kind seg_stride rows_source
 *   rows_arg concat_group concat_slot n_boundaries b0 b1 ..." — version 2 files
 *   ("fcp_plan 2") append "xform_mode xform_n substitute hash_buckets lo0 hi0 lo1
 *   hi1 ..." to every column line; version 3 files may end with a stage section
 *   (fcp_plan_file_stage_info) that tells Addons>ConcatInputs how to pack;
 *   version 4 files may carry, before it, "segmaps M" + M lines "column n sym
 *   slot mul0 mul1 mul2 mul3 div" (fcp_column_ext_t::seg_map_*).
 * `flags`: fcp_plan_desc_t::flags.  FCP_ERR_INVALID_ARGUMENT for a missing or
 * malformed file. */
int fcp_plan_create_from_file(const char *path, int32_t device, uint32_t flags,
                              fcp_plan_t **plan);
int fcp_plan_destroy(fcp_plan_t *plan);
/* Static facts: number of columns / groups / host inputs / tables / symbols
 * (any pointer may be NULL), */
int fcp_plan_counts(const fcp_plan_t *plan, int32_t *n_columns, int32_t *n_groups,
                    int32_t *n_host_inputs, int32_t *n_device_inputs,
                    int32_t *n_symbols);
/* The plan columns that are outputs of Addons>FeatureColumnProcess, in plan order
 * (every column except FCP_FORM_EXTERNAL ones): `indices` receives up to `capacity`
 * column indices, *n the count (either may be NULL). */
int fcp_plan_output_columns(const fcp_plan_t *plan, int32_t *n, int32_t *indices,
                            int32_t capacity);
/* Bytes of embedding tables this plan reads on THIS device (its shard of every
 * table, shared tables counted once) and the largest single table's bytes
 * (unsharded) — the inputs of the placement gate, fcp_placement_decide. */
int fcp_plan_table_bytes(const fcp_plan_t *plan, int64_t *shard_bytes,
                         int64_t *max_table_bytes_unsharded);
/* ---- placement gate (replaces check_table_size, cuda_emitter.cc:1080-1094) -------- */
/* The reference keeps a column on the CPU when its table exceeds max_table_size =
 * 256 MiB (fc_optimize_pass.cc:71, RECOM_CPU_GPU_CO_RUN).  On MI355X the question is
 * whether the model's tables fit ONE GPU's HBM: if they do, every GPU serves requests
 * on its own replica and nothing is exchanged; only when the aggregate exceeds one
 * GPU are the tables sharded over the `world` GPUs of the node, by whole columns
 * (final blocks exchanged, bit-identical results; needs every table to fit one GPU)
 * or by rows (partial sums exchanged + fcp_shard_finalize; any table size). */
enum {
  FCP_PLACE_REPLICATE = 0,
  FCP_PLACE_COLUMN_SHARD = 1,
  FCP_PLACE_ROW_SHARD = 2,
  /* whole tables wherever a table fits one GPU, rows only for the tables that do not: the fewest bytes on the wire
   * (a whole column sends its FINAL block, 1/world of a row-sharded column's dense partial sums).  As a preference
   * (prefer_mode) it resolves to COLUMN_SHARD when every table fits, to MIXED when some do not, and falls back to
   * ROW_SHARD when whole tables cannot be packed or fewer tables than ranks would stay whole (the whole-column step
   * gives every rank a block). */
  FCP_PLACE_MIXED = 3
};
typedef struct fcp_placement {
  int32_t mode;          /* FCP_PLACE_*                                        */
  int32_t min_world;     /* fewest GPUs on which the tables fit at all          */
  int64_t bytes_per_gpu; /* table bytes the fullest GPU holds under `mode`      */
} fcp_placement_t;
/* table_bytes[n_tables]: bytes of every (unsharded) table; hbm_bytes: one GPU's
 * memory; reserve_bytes: what must stay free (arenas, request blobs, runtime);
 * prefer_mode: FCP_PLACE_COLUMN_SHARD or FCP_PLACE_ROW_SHARD, taken when both are
 * feasible, or FCP_PLACE_MIXED.  FCP_ERR_UNSUPPORTED (with min_world set) when the
 * tables do not fit `world` GPUs. */
int fcp_placement_decide(const int64_t *table_bytes, int32_t n_tables,
                         int64_t hbm_bytes, int64_t reserve_bytes, int32_t world,
                         int32_t prefer_mode, fcp_placement_t *out);
/* The same decision with the assignment: owner[t] = the rank that holds table t whole, or -1 = its rows are spread over
 * all ranks (every table under ROW_SHARD; under MIXED only the tables larger than one GPU's budget), or 0 under
 * REPLICATE (every rank holds everything).  Whole tables are dealt longest-first onto the least loaded rank, on top of
 * the row shares every rank holds. */
int fcp_placement_assign(const int64_t *table_bytes, int32_t n_tables,
                         int64_t hbm_bytes, int64_t reserve_bytes, int32_t world,
                         int32_t prefer_mode, int32_t *owner, fcp_placement_t *out);

/* total concat width of a group (sum of dims in slot order)
 * and the element offset of a column inside its group. */
int fcp_plan_group_width(const fcp_plan_t *plan, int32_t group, int32_t *width);
int fcp_plan_column_offset(const fcp_plan_t *plan, int32_t column,
                           int32_t *offset);
/* Arena bytes ProcessFeatureColumns will request for these run-time shapes
 * (buffer_size_sum, cuda_emitter.cc:2151-2158). */
int fcp_plan_arena_bytes(fcp_plan_t *plan, const int32_t *concated_shapes,
                         const int32_t *symbols, int64_t *bytes);
/* Device error counter (FCP_FLAG_COUNT_BAD_IDS); synchronises `stream`. */
int fcp_plan_read_bad_ids(fcp_plan_t *plan, void *stream, int64_t *count);

/* ---- ProcessFeatureColumns (cuda_emitter.cc:2303-2494) ------------------- */
/* HIP graphs: a request whose shapes are resident (it ran once on this stream) only
 * enqueues kernels, so the call may be made while `stream` is being captured.  The
 * recorded launch reads the request's descriptor slot whenever the graph is replayed:
 * the slot is then kept until fcp_plan_release_captures (call it once the graphs are
 * destroyed).  Capturing a request whose shapes are NOT resident returns
 * FCP_ERR_UNSUPPORTED (descriptors cannot be installed inside a capture), as does a
 * request with new shapes once all 8 slots belong to captured graphs — a plan serves
 * at most 8 captured shapes at a time — and a request whose tables are not bound to
 * the plan yet, or have moved (binding copies records and may synchronise the device).
 * fcp_shard_finalize follows the same rules. */
int fcp_plan_release_captures(fcp_plan_t *plan);
int fcp_process_feature_columns(fcp_plan_t *plan,
                                const fcp_process_args_t *args,
                                fcp_process_result_t *result);

/* ---- plan-owned private streams: overlap behind ONE caller stream (EXPERIMENTAL: opt-in, frozen since round 5; the default
 * request path never enters it; recom_amd/csrc/fcp_lanes.hip) ---------- */
/* TensorFlow hands a GPU op exactly one compute stream
 * (feature_column_process_op_gpu.cu.cc:65-131 takes it from the op context; the
 * serve workers of the reference harness share one Session and therefore that one
 * stream, examples/cc/recom_examples.patch:193-216).  On one stream every request
 * pays its own kernel boundary, dependent front and drain; requests on
 * neighbouring streams hide them (S2, batch 512: ~28.5 us alone, ~23 us
 * overlapped).  fcp_plan_set_private_streams(plan, n, flags) gives the plan n
 * streams of its own (0 = off, the default; at most 16).  From then on
 * fcp_process_feature_columns
 *   - calls malloc_buff, THEN records an event on args->stream (everything
 *     enqueued there up to the allocation: the producer of the blob, and whatever
 *     still uses the memory malloc_buff has just handed out — TF's allocator reuses
 *     memory in compute-stream order, and another Session::Run thread may queue a
 *     reader of that memory at any time before the allocation),
 *   - runs the request on the next private stream (round robin), which waits for
 *     that event before its first command that touches the arena,
 *   - files the request's completion event under the address range of its arena
 *     (fcp_process_result_t::buffer, buffer_bytes).
 * args->stream itself does NOT wait for the kernels.  Whoever reads the arena —
 * Addons>ConcatOutputs in the rewritten graph, whose `tensor_buffers` inputs keep
 * blob, tables and arena alive until then (cuda_emitter.cc:2632-2643) — calls
 * fcp_result_wait(pointer into the arena, its stream) before it enqueues the
 * reader: a device-side wait, the host never blocks.  fcp_result_synchronize is
 * the same for a host reader.  Both return FCP_OK at once when nothing is pending
 * for that address (private streams off, or the request long complete).
 * fcp_concat_outputs_host performs the wait itself for its `out` pointer, fcp_concat_outputs and the
 * scatter variants for EVERY input (columns of one arena or of several) and for `out`.
 *
 * FCP_PRIVATE_NO_CALLER_WAIT: the private stream does not wait for args->stream.
 * Only for callers that guarantee by other means that the blob is complete and
 * that nothing still enqueued on their stream touches the arena memory (a
 * harness with its own arena ring); never under TensorFlow's allocator.
 * Requests issued while args->stream is being captured into a HIP graph stay on
 * args->stream.  Call it while no request of the plan is in flight (normally
 * right after plan creation); changing the count synchronises the old streams.
 * The sharded step (fcp_shard_step_run) always runs on args->stream.
 *
 * When the mode pays (round 5, profiles/r05_caller_threads_grid.txt): only when
 * the READER of a request is queued on args->stream at least one other request
 * later — every request waits for everything recorded on args->stream before its
 * allocation, readers included, so a reader right behind its request (what one
 * Session::Run of the rewritten graph does: FeatureColumnProcess, then
 * ConcatOutputs) serialises the requests again and the events are pure cost (S2
 * 30 -> 36-39 us, with 1 to 4 host threads on the stream); two requests behind:
 * 30.1 -> 24.7-25.3 us.  The supervisor below demotes callers for whom it does
 * not pay.
 *
 * Which requests take a private stream: the cross-stream events cost the host
 * about 8 us per request, so a request only gains when its kernel is long enough
 * to have something to overlap (S2 28 -> 24.5 us, RAGGED 27.8 -> 22 us per
 * request; the reference's models E / F, 10 us kernels, LOSE: 10 -> 16 us).  The
 * plan therefore keeps requests whose work — table rows gathered + output bytes
 * of the shapes it installed last — is below 48 MiB on args->stream
 * (FCP_PRIVATE_MIN_WORK_BYTES overrides); FCP_PRIVATE_ALWAYS sends every request
 * to a private stream.  Three streams measured best (two: 25-27 us on S2); with
 * four or more event-linked streams in flight every request took 35-100 us, so
 * at most three are used whatever n_streams says (the call still succeeds) — and
 * the streams belong to the DEVICE, not to the plan: every plan of the process on
 * that GPU rotates over the same three (two models with streams of their own
 * would be six event-linked queues).  They live as long as the process.
 *
 * Verification.  Whether event-linked streams overlap on the GPU depends on which
 * hardware queues the HIP runtime mapped them to — the creation order of every
 * stream of the process, GPU_MAX_HW_QUEUES, stream priorities — and no API shows
 * it: the same three private streams ran S2 at 24.5 us per request or at 40-86 us
 * (one stream: 28.6) with nothing changed but the number of streams the process
 * had created before (profiles/r04_private_streams_queue_mapping.txt).  The first
 * request of every caller stream that would take a private stream therefore runs
 * a short synthetic probe of the request pattern behind that stream (kernels that
 * only wait; the host blocks for ~8 ms and args->stream drains once).  While no
 * live plan relies on the present mapping, other mappings are tried — the private
 * streams are re-created with the next priority (normal, low, high) and behind up to six
 * spacer streams, ~8 ms each — within a WALL-TIME budget (a request: 120 ms,
 * FCP_PRIVATE_VERIFY_BUDGET_MS; fcp_plan_verify_private_streams: its argument).
 * A caller behind which no mapping overlaps keeps
 * its requests on its own stream: the mode then costs nothing instead of a
 * multiple.  (A verdict is kept per stream HANDLE for the life of the plan: a
 * process that destroys and re-creates its streams calls this function again.)
 * FCP_PRIVATE_NO_VERIFY skips all of it (the streams are used as
 * created); FCP_DIAG=lane_priority=normal|low|high chooses the priority the search
 * starts with; FCP_DIAG=private_verify_verbose prints the search.
 * Callers with a warm-up request (every deployment of the reference has one,
 * docs/build_from_source.md:42) call fcp_plan_verify_private_streams there: no
 * serving request then ever pays for the search.
 *
 * Supervision.  A verdict is learnt once, from a synthetic probe; what the private
 * streams buy the caller's REAL traffic — its readers, its host threads, its pace —
 * and whether the mapping still overlaps later, is measured while serving: an
 * online A/B.  An evaluation runs 48 consecutive requests of the caller on
 * args->stream between two timing events there, then 48 on the private streams
 * between two timing events on one of them, and compares the time per byte of
 * work (private / stream order: streams that overlap measure 0.64-0.92; readers
 * right behind their requests 1.13-1.64; sparse traffic ~1.0 — nothing to overlap;
 * a mapping that stopped overlapping 1.1 and more:
 * profiles/r05_caller_threads_grid.txt).  Two consecutive evaluations above 0.97
 * DEMOTE the caller: verdict 0 (a line on stderr only under FCP_DIAG=lane_log), its requests stay on
 * args->stream; two consecutive ones below it re-admit a demoted caller (a trickle
 * at start-up, load later).  Evaluations run at the caller's first eligible
 * request, 256 requests later, then at doubling gaps up to 8192 requests: < 1 %
 * of the traffic runs in the mode that loses.  Results are unaffected at every
 * point.  FCP_LANE_SUPERVISE=0 turns the supervisor off; FCP_LANE_SUPERVISE_PERIOD
 * (largest gap), FCP_LANE_KEEP_RATIO tune it; fcp_plan_private_streams_stats reads
 * it; fcp_plan_verify_private_streams (a new search) starts it over. */
enum { FCP_PRIVATE_NO_CALLER_WAIT = 1u << 0, FCP_PRIVATE_ALWAYS = 1u << 1, FCP_PRIVATE_NO_VERIFY = 1u << 2 };
/* experimental */ int fcp_plan_set_private_streams(fcp_plan_t *plan, int32_t n_streams, uint32_t flags);

/* Diagnostic: do the plan's private streams overlap behind THIS caller stream, in THIS process?  Whether event-linked
 * streams overlap depends on which hardware queues the HIP runtime mapped them to — the creation order of every stream of
 * the process, GPU_MAX_HW_QUEUES, stream priorities — and cannot be told from the API: the same three private streams
 * measured 24.5 us per S2 request or 40-85 us (one stream: 28.6) with nothing changed but the number of streams the
 * process had created before (profiles/r04_private_streams_queue_mapping.txt).  The probe replays the request pattern
 * with kernels that only wait: `requests` kernels of `spin_us` microseconds on grid_blocks x 256 threads, each followed
 * — n_streams - 1 requests later — by its consumer (a stream wait + a one-thread kernel) on `stream`; once back to back
 * on `stream` (*serial_us), once through the private streams (*lanes_us, 0 without private streams); host clock, each
 * ending with a synchronisation of `stream`.  *lanes_us well below *serial_us: they overlap.  (The library runs this
 * probe itself on the first request of every caller stream, see Verification above; the entry point is for harnesses
 * and for processes that pass FCP_PRIVATE_NO_VERIFY.) */
/* experimental */ int fcp_plan_probe_private_streams(fcp_plan_t *plan, void *stream, int32_t requests, int32_t spin_us, int32_t grid_blocks,
                                   double *serial_us, double *lanes_us);
/* What the verification decided for `stream`: *verdict = 1 (its requests take the private streams), 0 (they stay on
 * `stream`: nothing overlapped behind it) or -1 (no request of that stream verified yet, or the mode is off). */
/* experimental */ int fcp_plan_private_streams_verdict(fcp_plan_t *plan, void *stream, int32_t *verdict);
/* The verification at a time of the caller's choosing (warm-up): probes the private streams behind `stream` now and, while
 * no live plan relies on the present mapping, searches another one for at most about budget_ms of wall time (<= 0: 400 ms;
 * one mapping costs ~8 ms, the whole search space ~22 of them).  A negative verdict of an earlier, cheaper look — or a
 * demotion by the supervisor — is forgotten and the search runs again (the supervisor then starts over with an evaluation at
 * the next request); a positive one is returned as it is.  Blocks the
 * host, drains `stream`.  *verdict (optional) as fcp_plan_private_streams_verdict; -1 when the mode is off, or when the
 * plan's requests so far are below the work threshold (they stay on `stream` anyway: nothing is probed).  Call it after
 * the first (warm-up) request of the plan, as the shim does. */
/* experimental */ int fcp_plan_verify_private_streams(fcp_plan_t *plan, void *stream, int32_t budget_ms, int32_t *verdict);
/* What the run-time supervisor of the plan's private streams has seen (see Supervision above). */
typedef struct fcp_private_streams_stats {
  void *supervised_stream;        /* the caller stream under supervision (the first that took the private streams), or NULL   */
  int64_t requests;               /* its requests that were eligible for a private stream so far                              */
  int64_t lane_requests;          /* ... of which ran on one                                                                  */
  int64_t evaluations;            /* completed A/B evaluations (48 requests in stream order, 48 on the private streams)       */
  double stream_order_us_per_mib; /* last evaluation: stream-order time per MiB of work (rows gathered + output written)      */
  double last_ratio;              /* last evaluation: time per byte on the private streams / in stream order                  */
  double worst_ratio;             /* the largest such ratio so far                                                            */
  double keep_ratio;              /* private streams are kept while the ratio stays at or below this (0.97)                   */
  int32_t demoted;                /* 1: the supervisor keeps this caller's requests on its own stream at present              */
  int32_t evaluation_in_progress; /* 1: an evaluation is running or waiting for its timing events                             */
} fcp_private_streams_stats_t;
/* experimental */ int fcp_plan_private_streams_stats(fcp_plan_t *plan, fcp_private_streams_stats_t *out);

/* The cheap half of the same idea, for callers that OWN their buffers: FCP_ORDER_INPUTS_READY is the caller's promise, for
 * every request of the plan, that when fcp_process_feature_columns is CALLED the blob is complete in device memory and
 * nothing still queued or running on args->stream touches the memory malloc_buff returns.  The fused kernel is then
 * launched without the queue's barrier bit, so the command processor need not wait for the queue to drain before it takes
 * the next packet: the BOUNDARY between two requests shrinks — on ONE stream, without events or extra streams (S2 back to
 * back: 28.3 -> 26.0-26.5 us per request, 0.60-0.61 of 8 TB/s; Zipf ids 25.4 -> 23.4).  What it does NOT do on this runtime
 * (round 5, scripts/probes/any_order_probe.hip): start the kernel while blocks of its predecessor still run — behind a
 * kernel whose blocks retire between 10 and 20 us the first block of the next one starts at 21.4 us without the barrier bit,
 * 22.7 us with it, wave slots free from 10 us on — so there is no overlap of a front with a predecessor's tail, only a
 * cheaper hand-over.  Everything queued BEHIND the kernel on args->stream (the consumer) still waits for it, as any
 * stream-ordered command waits for all commands before it — which is also why the gain is only there while requests
 * follow each other directly: an ordinary command between two requests (a consumer kernel, an event record) orders
 * the second request behind the first again; private streams (above) are the tool for that pattern.  NOT for
 * TensorFlow's allocator, which hands out memory that earlier, still queued kernels of the compute stream may be using.
 * Requests that queue work of their own in front of the kernel keep stream order for the kernel; of that work the
 * segment-offset pre-pass is itself launched without the barrier bit (it reads the blob and writes the new arena only:
 * RAGGED as delivered 34.4 -> 33.5 us), the inverse-map memset and the descriptor upload kernel are not.
 * Default: FCP_ORDER_STREAM. */
enum { FCP_ORDER_STREAM = 0, FCP_ORDER_INPUTS_READY = 1 };
/* experimental */ int fcp_plan_set_request_order(fcp_plan_t *plan, int32_t order);
/* experimental */ int fcp_result_wait(const void *buffer, void *stream);
/* A host reader: returns once the request has completed on the device.  What that guarantees is device-scope: copies and
 * kernels issued afterwards (hipMemcpy*, any stream) see the result.  A host that reads the arena DIRECTLY — host-mapped or
 * fine-grained memory — gets no system-scope visibility from it (the completion events carry no system fence, which is what
 * keeps them off the GPU's critical path): such a reader copies through the device or synchronises a stream of its own. */
/* experimental */ int fcp_result_synchronize(const void *buffer);

/* ---- ConcatOutputs (concat_outputs_op_gpu.cu.cc:85-140) ------------------ */
/* out[p, off_k : off_k + dims[k]] = inputs[k][p*dims[k] ...] for k < n.
 * `inputs` is a HOST array of device pointers; the pointer table is passed to
 * the kernel without a separate H2D copy when n is small.  elem_size 4 only
 * (reference registers float and int, :250-251). */
int fcp_concat_outputs(const void *const *inputs, const int32_t *dims,
                       int32_t n, int64_t prefix_size, void *out, void *stream);

/* The same with explicit destinations: input k ([prefix, dims[k]], contiguous, device)
 * goes to columns [col_offsets[k], col_offsets[k] + dims[k]) of the row-major
 * matrix `out` [prefix, out_width].  (ScatterBlock<EmbedDim, Offset>,
 * concat_outputs_op_gpu.cu.cc:85-99, with run-time offsets.) */
int fcp_concat_outputs_scatter(const void *const *inputs, const int32_t *dims,
                               const int32_t *col_offsets, int32_t n,
                               int64_t prefix_size, int32_t out_width, void *out,
                               void *stream);
/* The same for inputs that are column ranges of wider matrices: input k has row stride in_strides[k] floats
 * (>= dims[k]; NULL: contiguous).  The mixed-placement step assembles its output this way from the row-sharded
 * part and every rank's whole-column block. */
int fcp_concat_outputs_scatter_strided(const void *const *inputs, const int32_t *dims,
                                       const int32_t *in_strides, const int32_t *col_offsets,
                                       int32_t n, int64_t prefix_size, int32_t out_width,
                                       void *out, void *stream);
/* Addons>ConcatOutputs `host_inputs` (concat_outputs_op_gpu.cu.cc:186-216): the n
 * HOST tensors ([prefix, dims[k]], 4-byte elements) are packed into one pinned
 * staging buffer (library-owned, per device) and scattered into columns
 * [col_offsets[k], +dims[k]) of `out` [prefix, out_width] by one kernel on `stream`
 * — reading the pinned buffer through its device mapping when the payload is at
 * most 1 MiB (no copy at all), otherwise after ONE asynchronous H2D copy into
 * scratch obtained from `malloc_temp` (the reference's allocate_temp, :189-193;
 * called only then: it may be NULL for payloads of at most 1 MiB).  Calls from
 * several threads overlap: the per-device ring lock covers slot bookkeeping only.
 * With FCP_LAYOUT_CONCAT `out` is the group's
 * matrix inside the FeatureColumnProcess arena, whose FCP_FORM_EXTERNAL slots are
 * exactly these columns.  No stream synchronisation (the reference blocks, :234);
 * the host tensors may be reused when the call returns. */
int fcp_concat_outputs_host(const void *const *host_inputs, const int32_t *dims,
                            const int32_t *col_offsets, int32_t n,
                            int64_t prefix_size, int32_t out_width, void *out,
                            fcp_alloc_fn malloc_temp, void *malloc_temp_ctx,
                            int32_t device, void *stream);

/* ---- request staging: ConcatInputs + the H2D copy as one step -- */
/* The reference packs N host tensors with N mempcpy calls on one CPU thread into a
 * pageable TF tensor (concat_inputs_ops.cc:69-76) which TF then copies H2D.  A
 * stager owns a ring of pinned host buffers with device twins: fcp_stager_stage
 * packs the tensors (same bytes, offsets and shapes as fcp_concat_inputs) with
 * `n_threads` worker threads straight into pinned memory and enqueues ONE
 * hipMemcpyAsync on the stager's own copy stream (so the copy of request k+1
 * overlaps the kernel of request k); `stream` is made to wait for that copy.  The
 * returned pointers stay valid until the slot is reused, i.e. for the next depth-1
 * calls; the work that reads them must be enqueued on `stream` before the next call.
 * (A plan with private streams runs that work elsewhere: fcp_process_feature_columns
 * then files the blob it was given next to the arena, and the stager waits for that
 * reader — the copy stream, or the host for a zero-copy slot — before it overwrites
 * the slot.  Nothing to do for the caller.)
 *
 * fcp_stager_stage_narrow additionally converts the int64 inputs flagged in
 * `narrow_int64[n_inputs]` to int32 while packing (values outside [0, 2^31) become
 * -1, an invalid id / row either way): half the PCIe bytes for id and index tensors.
 * The plan consuming such a blob declares those inputs as 4-byte (FCP_IDS_I32 /
 * FCP_SEG_IDS_I32); results are unchanged.
 *
 * FCP_STAGER_ZERO_COPY (fcp_stager_create_ex): no copy at all — the ring is pinned
 * memory mapped into the device's address space and the returned "device blob" is
 * that mapping: the kernels fetch the ids over PCIe themselves.  Fewer runtime calls
 * per request and no copy engine in the path (S2: 65 us per request steadily, where
 * the copying form swings between 63 and 150 us with the host's load; lone-request
 * latency 100-117 vs 121-137 us), at the price of a kernel that holds its CUs for the
 * duration of the transfer.  A slot is repacked only after the work that read it has
 * finished (the call waits on the host if it must).
 *
 * (r5) Inside one request the pack and the copy overlap: the inputs are packed in up to four groups (FCP_STAGER_GROUPS) and
 * every group is shipped the moment its last byte is in the pinned buffer (by the calling thread, which watches the pack
 * workers instead of packing) — a lone request costs pack + copy / 4 + kernel instead of pack + copy + kernel.
 * (Groups are used when the caller is not issuing back to back — more than 40 us since the previous call returned: under
 * load they cost throughput, S2 58 -> 66 us per request, because the caller no longer packs.)
 * The copies themselves are KERNELS on the stager's copy stream that read the pinned ring through its device mapping
 * (FCP_STAGER_COPY_KERNEL, the default since round 5): hipMemcpyAsync's SDMA submission blocks its caller for 6-14 ms a few
 * times per thousand calls on the boxes this was measured on (profiles/r05_pcie_staging_stalls.txt), kernel copies never
 * did.  FCP_STAGER_COPY_SDMA asks for the copy engine; FCP_STAGER_COPY=kernel|sdma overrides both at run time. */
typedef struct fcp_stager fcp_stager_t;
enum { FCP_STAGER_DEFAULT = 0, FCP_STAGER_ZERO_COPY = 1, FCP_STAGER_COPY_KERNEL = 2, FCP_STAGER_COPY_SDMA = 4 };
int fcp_stager_create(int32_t device, int64_t capacity_bytes, int32_t max_inputs,
                      int32_t max_rank_sum, int32_t depth, int32_t n_threads,
                      fcp_stager_t **stager);
int fcp_stager_create_ex(int32_t device, int64_t capacity_bytes, int32_t max_inputs,
                         int32_t max_rank_sum, int32_t depth, int32_t n_threads,
                         uint32_t flags, fcp_stager_t **stager);
int fcp_stager_stage(fcp_stager_t *stager, const fcp_host_tensor_t *inputs,
                     int32_t n_inputs, void *stream, const void **device_blob,
                     int64_t *blob_bytes, const int32_t **offsets,
                     const int32_t **shapes);
int fcp_stager_stage_narrow(fcp_stager_t *stager, const fcp_host_tensor_t *inputs,
                            int32_t n_inputs, const uint8_t *narrow_int64, void *stream,
                            const void **device_blob, int64_t *blob_bytes,
                            const int32_t **offsets, const int32_t **shapes);
/* The general form: modes[n_inputs] (NULL: all FCP_STAGE_COPY) says what happens to each input while it is packed.
 * FCP_STAGE_SEG_TO_CSR turns the sorted row ids of a multi-hot feature — an int32 / int64 tensor [nnz], or the
 * SparseTensor indices [nnz, k] whose column 0 they are — into the int32 row offsets[rows + 1] the kernels want
 * (mode_args[i] = rows; the plan declares that input FCP_SEG_CSR_I32, rank 1): the host reads those bytes anyway to
 * pack them, the device then needs neither the segment-offset pre-pass nor the in-block search, and 16 bytes per id
 * shrink to 4 bytes per ROW on the wire (RAGGED, BASELINE configs[3]: 15.7 MB of blob -> 3.1 MB with narrowed ids). */
enum { FCP_STAGE_COPY = 0, FCP_STAGE_NARROW_I64 = 1, FCP_STAGE_SEG_TO_CSR = 2 };
int fcp_stager_stage_ex(fcp_stager_t *stager, const fcp_host_tensor_t *inputs, int32_t n_inputs,
                        const uint8_t *modes, const int64_t *mode_args, void *stream,
                        const void **device_blob, int64_t *blob_bytes, const int32_t **offsets,
                        const int32_t **shapes);
int fcp_stager_destroy(fcp_stager_t *stager);
/* What the stager has seen so far: staging calls, copy calls (one per group), how long the slowest copy CALL held its host
 * thread, how many held it for more than a millisecond (the "14 ms stalls" of profiles/r04_pcie_staging_memcpy_anomaly.txt),
 * requests the zero-copy fallback served. */
typedef struct fcp_stager_stats {
  int64_t calls, copy_calls, copy_calls_over_1ms, fallback_switches, requests_with_blocked_copy;
  double max_copy_call_us;
} fcp_stager_stats_t;
int fcp_stager_stats(fcp_stager_t *stager, fcp_stager_stats_t *out);

/* ---- Addons>ConcatInputs in its staged form (host only; replaces concat_inputs_ops.cc:42-77) ---- */
/* The same packing as fcp_stager_stage_ex — modes[n_inputs] / mode_args[n_inputs] as there, NULL = plain
 * fcp_concat_inputs — into a caller-owned blob, on the calling thread, without a stager or a device: what the TF
 * shim's ConcatInputsOp runs (its output 0 is the blob; TensorFlow copies it to the GPU).  The reference's op copies
 * every input byte for byte (RAGGED, BASELINE configs[3]: 15.7 MB per request); with the modes of the plan file's
 * stage section the blob carries int32 ids and int32 row offsets (3.1 MB) and the device runs neither the
 * segment-offset pre-pass nor the in-block search.  The consuming plan is the STAGED plan (PlanSpec.staged()):
 * narrowed inputs declared 4-byte, converted ones FCP_SEG_CSR_I32 of rank 1. */
int fcp_concat_inputs_ex_sizes(const fcp_host_tensor_t *inputs, int32_t n_inputs,
                               const uint8_t *modes, const int64_t *mode_args,
                               int64_t *blob_bytes, int32_t *rank_sum);
int fcp_concat_inputs_ex(const fcp_host_tensor_t *inputs, int32_t n_inputs,
                         const uint8_t *modes, const int64_t *mode_args, void *blob,
                         int64_t blob_capacity, int32_t *offsets, int32_t *shapes);
/* The same on a worker pool: the reference's op packs on one thread (concat_inputs_ops.cc:42-77), which for RAGGED is
 * 1.4 ms per request of the staged pack (0.5 ms of plain copying) — longer than everything the GPU does with it.  A
 * pool is a set of sleeping threads that split one call's inputs into ranges of about equal INPUT bytes; the calling
 * thread works too.  One call at a time uses a pool: a second concurrent caller packs on its own thread instead of
 * waiting (so an op instance shared by several serve workers never blocks on it).  Host only. */
typedef struct fcp_pack_pool fcp_pack_pool_t;
int fcp_pack_pool_create(int32_t n_threads, fcp_pack_pool_t **pool);
int fcp_pack_pool_destroy(fcp_pack_pool_t *pool);
int fcp_concat_inputs_ex_pool(fcp_pack_pool_t *pool /* NULL: this thread only */,
                              const fcp_host_tensor_t *inputs, int32_t n_inputs,
                              const uint8_t *modes, const int64_t *mode_args,
                              void *blob, int64_t blob_capacity, int32_t *offsets, int32_t *shapes);
/* The stage section of a column-plan file (version 3, written by `python -m recom_amd.graph --staged`):
 *   stage N symbols_input K / N lines "mode rows_symbol"
 * N = number of Addons>ConcatInputs inputs of the rewritten graph (= the plan's host inputs); modes[i] = FCP_STAGE_*
 * of input i; rows_symbol[i] = for FCP_STAGE_SEG_TO_CSR inputs the index into the `symbols` vector of the row count
 * (mode_args[i] = symbols[rows_symbol[i]]), else -1; *symbols_input = which ConcatInputs input IS that symbols
 * vector (int32[n_symbols]; the rewritten graph wires it in as the op's last input), or -1 when nothing is
 * converted.  *n_inputs = 0: the file has no stage section (a plain plan).  Up to `capacity` entries are written. */
int fcp_plan_file_stage_info(const char *path, int32_t *n_inputs, uint8_t *modes,
                             int32_t *rows_symbol, int32_t capacity, int32_t *symbols_input);

/* ---- plan builder from a GraphDef (replaces CudaEmitter::Optimize, cuda_emitter.cc:80-116) ---- */
/* The non-codegen half of the reference's CudaEmitter as ONE in-process call for the retained Grappler pass: where
 * the reference generates CUDA text per feature column (EmitFCCode, cuda_emitter.cc:975-1178), runs nvcc, caches the
 * .so by md5 and rewrites the graph (Rewrite, :2496-2656), the pass serialises its GraphDef, calls fcp_graph_build and
 * parses what comes back.  `graphdef`: a serialized tensorflow.GraphDef in the form the reference's emitter sees (after
 * the lookup optimizers, lookup_optimizer.cc:157-440).  The column plan is WRITTEN to `plan_path` (the file the
 * rewritten FeatureColumnProcess names in `dlpath`); *rewritten receives the serialized rewritten GraphDef
 * (malloc'd; fcp_graph_free) unless `rewritten` is NULL; *description a human-readable summary (fcp_graph_free), may
 * be NULL.  FCP_ERR_UNSUPPORTED: no ConcatV2 of embedding lookups in this graph ("nothing to fuse": the pass leaves
 * the graph alone).  No TensorFlow, no protobuf library, no Python, no device.  The same walk exists as an offline
 * tool (`python -m recom_amd.graph`); both write identical plan files (tests/test_graph_plan.py). */
enum {
  FCP_GRAPH_HOST_CONCAT_EXTERNAL = 1u << 0, /* non-lookup concat inputs stay Addons>ConcatOutputs host inputs (the
                                               reference's wiring) instead of passthrough columns                 */
  FCP_GRAPH_STAGED = 1u << 1,               /* the staged plan + stage section; ConcatInputs gets `_fcp_plan`       */
  FCP_GRAPH_NO_PRUNE = 1u << 2              /* keep the replaced subgraphs in the output graph                     */
};
int fcp_graph_build(const void *graphdef, size_t n_bytes, uint32_t flags, const char *plan_path,
                    void **rewritten, size_t *rewritten_bytes, char **description);
void fcp_graph_free(void *p);

/* ---- multi-GPU exchange (no reference counterpart) ---------- */
/* The ONE collective of the sharded path: an all-to-all partitioned along the batch,
 * issued as grouped ncclSend / ncclRecv (RCCL) to every peer at once so that all
 * xGMI links of the GPU carry traffic concurrently, on the request's stream, between
 * the partial kernel and fcp_shard_finalize / fcp_concat_outputs.  RCCL is bound at
 * run time (librccl.so.1); FCP_ERR_UNSUPPORTED where it is missing.
 *
 * A communicator belongs to one GPU of one process (one process per GPU).  Rank 0
 * calls fcp_comm_unique_id and ships the 128 bytes to the other ranks by any
 * host-side channel; every rank then calls fcp_comm_create (collective). */
#define FCP_COMM_ID_BYTES 128
#define FCP_MAX_GROUPS_ABI 16 /* fcp_plan_desc_t::n_groups is at most this */
typedef struct fcp_comm fcp_comm_t;
int fcp_comm_unique_id(uint8_t *id /* [FCP_COMM_ID_BYTES] */);
int fcp_comm_create(const uint8_t *id, int32_t rank, int32_t world, int32_t device,
                    fcp_comm_t **comm);
int fcp_comm_destroy(fcp_comm_t *comm);
int fcp_comm_rank(const fcp_comm_t *comm, int32_t *rank, int32_t *world);
/* The batch split every exchange uses: contiguous, the first rows % world ranks get
 * one row more. */
int fcp_shard_batch_slice(int64_t rows, int32_t world, int32_t rank, int64_t *begin,
                          int64_t *count);
/* Row sharding: `partial` = this rank's [rows, width] partial sums (a row-sharded
 * plan's group matrix); on return (enqueued) `slices` holds [world, row_count, width]:
 * rows [row_begin, row_begin + row_count) of every rank's partial, in rank order —
 * the input of fcp_shard_finalize. */
int fcp_shard_exchange(fcp_comm_t *comm, const void *partial, int64_t rows,
                       int64_t width, void *slices, int64_t *row_begin,
                       int64_t *row_count, void *stream);
/* Column sharding: `block` = this rank's [rows, widths[rank]] final column block;
 * `recv` receives, back to back, the [row_count, widths[g]] blocks of g = 0..world-1
 * (fcp_concat_outputs[_scatter] puts them side by side). */
int fcp_shard_exchange_columns(fcp_comm_t *comm, const void *block, int64_t rows,
                               const int32_t *widths, void *recv, int64_t *row_begin,
                               int64_t *row_count, void *stream);
/* One native call per sharded request: partial kernel -> exchange -> finalize (row
 * mode, FCP_PLACE_ROW_SHARD) or concat (FCP_PLACE_COLUMN_SHARD) of concat group
 * `group`, all enqueued on args->stream, from buffers the object owns (a ring of 3;
 * args' allocator callbacks are ignored).  max_rows / max_arena_bytes bound the
 * requests it will see (fcp_plan_arena_bytes); col_widths[world]: column mode only.
 * *out: device [row_count, width] of this rank's batch slice, valid for the next two
 * calls.  Calls may come on different streams and host threads: they are serialised on
 * the step object for the duration of the ENQUEUE (a ring entry waits for its previous
 * use; RCCL wants its collectives issued one at a time and in the same order on every
 * rank), the enqueued work of different streams still overlaps on the GPU. */
typedef struct fcp_shard_step fcp_shard_step_t;
int fcp_shard_step_create(fcp_plan_t *plan, fcp_comm_t *comm, int32_t mode,
                          int32_t group, int64_t max_rows, int64_t max_arena_bytes,
                          const int32_t *col_widths, fcp_shard_step_t **step);
int fcp_shard_step_run(fcp_shard_step_t *step, const fcp_process_args_t *args,
                       void **out, int64_t *row_begin, int64_t *row_count);
int fcp_shard_step_destroy(fcp_shard_step_t *step);

/* ---- multi-GPU finalize (no reference counterpart) --------- */
/* After the all-to-all of per-rank partial sums: out = sum over `world`
 * slices in rank order; for MEAN columns divide by the segment length read
 * from the same blob.  partial_slices: device [world, rows, width]. */
int fcp_shard_finalize(fcp_plan_t *plan, const fcp_process_args_t *args,
                       int32_t group, const void *partial_slices,
                       int32_t world, int64_t row_begin, int64_t row_count,
                       void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FCP_HIP_H_ */
