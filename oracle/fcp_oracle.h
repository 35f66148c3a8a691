/*
 * fcp_oracle.h — CPU oracle for the fused feature-column path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under recom_amd/ may include, link, load
 * or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY PINNING: the reference (AlibabaResearch/recom) ships no tests, golden
 * vectors or fixtures for this path and its implementation
 * cannot be built here (device code exists only as C++ string fragments that
 * need TensorFlow 2.6.2 + SymEngine + nvcc + CUB 1.8 to assemble).  The oracle
 * is therefore pinned against independent implementations of the same TF op
 * semantics that ARE importable in the build container — PyTorch-CPU
 * `embedding_bag` / `bucketize(right=True)` / `index_select` and a NumPy
 * float64 restatement — through tests/test_oracle.py and the committed
 * fixtures in tests/golden/, and TensorFlow's documented examples of every
 * fused op (tests/golden/tf_doc_examples.py).  Two fragments of the reference
 * ARE plain C++ inside their string literals and are compiled from the
 * reference's own source, where it lies, into oracle/_ref/ by
 * oracle/ref_extract.py: `Bucketize` (cuda_emitter.cc:233-247) and the arena
 * alignment `alignmem` (:967-969); orc_bucketize and the per-column arena
 * layout are held to them.  For everything else parity relative to the
 * reference itself is "unpinned by reference-owned vectors" and DESIGN.md
 * says so.
 */
#ifndef FCP_ORACLE_H_
#define FCP_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same numeric values as include/fcp_hip.h (kept separate on purpose). */
enum { ORC_FORM_GATHER = 1, ORC_FORM_SEGMENT_REDUCE = 2, ORC_FORM_GATHER_SCATTER = 3,
       ORC_FORM_PASSTHROUGH = 4, ORC_FORM_BATCH_COL_REDUCTION = 5,
       /* a concat slot filled by Addons>ConcatOutputs `host_inputs` (concat_outputs_op_gpu.cu.cc:186-216):
          ProcessFeatureColumns leaves it untouched */
       ORC_FORM_EXTERNAL = 6 };
enum { ORC_COMBINER_NONE = 0, ORC_COMBINER_SUM = 1, ORC_COMBINER_MEAN = 2 };
enum { ORC_IDS_I32 = 0, ORC_IDS_I64 = 1, ORC_IDS_F32_BUCKETIZE = 2 };
enum { ORC_XFORM_NONE = 0, ORC_XFORM_SELECT = 1, ORC_XFORM_FILTER = 2 };
enum { ORC_SEG_NONE = 0, ORC_SEG_IDS_I32 = 1, ORC_SEG_IDS_I64 = 2, ORC_SEG_CSR_I32 = 3 };
enum { ORC_ROWS_FROM_IDS = 0, ORC_ROWS_FROM_SYMBOL = 1, ORC_ROWS_FROM_INPUT_DIM0 = 2,
       ORC_ROWS_FROM_GROUP = 3 /* external slots: the row count of the other columns of the group */ };

typedef struct orc_column {
  int32_t form, combiner, dim, id_source;
  int64_t vocab;
  int32_t table_input, ids_input, seg_input, seg_kind, seg_stride;
  int32_t rows_source, rows_arg;
  int32_t n_boundaries;
  const float *boundaries;
  int32_t concat_group, concat_slot;
  /* id transform in front of the lookup: the reference's CPU ops Addons>SelectValue (mode 1,
   * custom_ops/select_value/select_value_ops.cc:33-56) and Addons>GatherIndiceValue /
   * Addons>GatherValueGenIndice (mode 2, gather_indice_value_ops.cc:33-78,
   * gather_value_gen_indice_ops.cc:33-67) over closed intervals [lo_i, hi_i] =
   * left_boundaries / right_boundaries.  Restated with the INTENDED test lo <= x && x <= hi: the
   * reference's `x >= l || x <= r` is true for every x. */
  int32_t xform_mode, xform_n;
  const int64_t *xform_lo, *xform_hi;
  int64_t xform_substitute;
  /* > 0: id = Fingerprint64(AsString(id)) % hash_buckets first (TensorFlow AsString ->
   * StringToHashBucketFast, the graph of categorical_column_with_hash_bucket over integer features;
   * reference models: examples/python/dlrm.py "hash-int" columns) */
  int64_t hash_buckets;
  /* seg_map_n > 0: the segment id of element i is a function of its first seg_map_n index coordinates — a
   * SparseReshape between the SparseTensor and the lookup, which the reference folds into the generated index
   * expression (EmitInputInline, cuda_emitter.cc:1874-1916: flat index over the input shape, "/" by the trailing
   * output dims, "%" by the dim itself; for the row coordinate, offset 0, the "%" is absent):
   *   seg = (sum_k idx[i*seg_stride + k] * seg_map_mul[k]) / seg_map_div,
   * one factor (seg_map_sym_slot: 0..3 = mul, 4 = div) times symbols[seg_map_sym] when seg_map_sym >= 0. */
  int32_t seg_map_n, seg_map_sym, seg_map_sym_slot, seg_map_pad;
  int64_t seg_map_mul[4];
  int64_t seg_map_div;
} orc_column_t;

/* TensorFlow 2.6.2's Fingerprint64 (core/platform/fingerprint.h -> FarmHash farmhashna::Hash64, third_party
 * farmhash 816a4ae6; un-vendored here) for strings of at most 32 bytes, restated from the published
 * algorithm.  Known answers it is pinned to (tests/test_oracle.py): "" -> k2; "abc" -> 0x24a5b3a074e7f369
 * (the CityHash64 v1.1 / FarmHash na test value for short strings, the same code path);
 * tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) == [0, 2, 2] (TensorFlow API docs). */
uint64_t orc_fingerprint64(const char *s, size_t len);
/* StringToHashBucketFast(AsString(value), num_buckets) */
int64_t orc_hash_bucket_int64(int64_t value, int64_t num_buckets);

typedef struct orc_plan {
  int32_t n_columns;
  const orc_column_t *columns;
  int32_t n_host_inputs;
  const int32_t *host_input_ranks;
  const int32_t *host_input_elem_sizes;
  int32_t n_groups;
  int32_t shard_rank, shard_world;
} orc_plan_t;

/* a5  cuda_emitter.cc:233-247 */
int32_t orc_bucketize(const float *boundaries, int32_t n, float value);
void orc_bucketize_array(const float *boundaries, int32_t n, const float *values,
                         int64_t count, int32_t *out);

/* a1  custom_ops/concat_inputs/concat_inputs_ops.cc:42-77 */
int64_t orc_concat_inputs(const void *const *datas, const int64_t *nbytes,
                          const int32_t *ranks, const int64_t *const *dims,
                          int32_t n, int8_t *blob, int32_t *offsets,
                          int32_t *shapes);

/* a6  cuda_emitter.cc:250-293, driver :1305-1327 */
int64_t orc_gather_rows(const float *table, int64_t vocab, int32_t dim,
                        const int64_t *ids, int64_t n, float *out,
                        int64_t out_stride);

/* a9  cuda_emitter.cc:296-345, zero-fill :1351-1359 */
int64_t orc_gather_scatter_rows(const float *table, int64_t vocab, int32_t dim,
                                const int64_t *ids, const int64_t *rows,
                                int64_t n, int64_t num_rows, float *out,
                                int64_t out_stride);

/* a8  cuda_emitter.cc:768-818 (ComputeSegmentOffsets) */
void orc_segment_offsets(const int64_t *seg_ids, int64_t n,
                         int64_t num_segments, int32_t *offsets);

/* a7/a8  cuda_emitter.cc:402-501, :564-661, :831-909.  Sequential fp32
 * accumulation in id order; mean = sum / count (:625, :903); empty = 0. */
int64_t orc_sparse_segment_reduce(const float *table, int64_t vocab, int32_t dim,
                                  const int64_t *ids, const int32_t *offsets,
                                  int64_t num_segments, int32_t mean, float *out,
                                  int64_t out_stride);

/* a8 in the reference's own summation order (8 `ty` lanes stride the rows of a
 * segment, then an LDS pairwise tree, cuda_emitter.cc:868-906).  Used only to
 * bound the fp32 reordering error against the sequential order. */
void orc_sparse_segment_reduce_ref8x8(const float *table, int32_t dim,
                                      const int64_t *ids, const int32_t *offsets,
                                      int64_t num_segments, int32_t mean,
                                      float *out, int64_t out_stride);

/* a7 / a8 in the addition order of TensorFlow 2.6.2's CPU kernel (SparseSegmentReductionOpBase::Reduce,
 * tensorflow/core/kernels/segment_reduction_ops_impl.h — third party, restated from its published source, unpinned):
 * first num & 7 rows (8 for 0, 9 for 1) left to right, divided by num at once when mean && num < 10, then every further
 * 8 rows summed among themselves and added; mean && num >= 10 divides at the end.  Equal to the sequential order for
 * bags of up to 9 ids.  Used to bound the HIP path against the north star's "vs TF-CPU" tolerance. */
void orc_sparse_segment_reduce_tfcpu(const float *table, int32_t dim, const int64_t *ids, const int32_t *offsets,
                                     int64_t num_segments, int32_t mean, float *out, int64_t out_stride);

/* a7 in the reference's own summation order for dim <= 20 (64-id tiles, CUB 1.8 BlockScan with
 * BLOCK_SCAN_WARP_SCANS over (row vector, head flag) pairs, carry across tiles; cuda_emitter.cc:348-661,
 * :1542-1618).  row_ids[nnz] sorted ascending.  Used only to bound the fp32 reordering error. */
void orc_sparse_segment_reduce_refscan(const float *table, int32_t dim, const int64_t *ids,
                                       const int64_t *row_ids, int64_t nnz, int64_t num_segments,
                                       int32_t mean, float *out, int64_t out_stride);
/* the same with the order INSIDE the 64-item scan selectable: CUB 1.8's (the function above) or rocPRIM's one-wavefront
 * scan, which is what the reference's template runs on when compiled against hipCUB (oracle/_ref/libref_device_scan.so) */
enum { ORC_SCAN_CUB18 = 0, ORC_SCAN_ROCPRIM64 = 1 };
void orc_sparse_segment_reduce_refscan_assoc(const float *table, int32_t dim, const int64_t *ids,
                                             const int64_t *row_ids, int64_t nnz, int64_t num_segments,
                                             int32_t mean, int32_t assoc, float *out, int64_t out_stride);

/* a11 cuda_emitter.cc:1216-1241 */
void orc_batch_col_reduction(const float *x, int64_t batch, int64_t rows,
                             int64_t cols, float *out, int64_t out_stride);

/* a10 concat_outputs_op_gpu.cu.cc:85-99 */
void orc_concat_outputs(const float *const *inputs, const int32_t *dims,
                        int32_t n, int64_t prefix, float *out);

/* Whole path a2..a10: blob + tables -> one [rows_g, width_g] matrix per group.
 * group_out[g] must hold rows_g*width_g floats.  Returns the number of ids
 * outside [0,vocab) (their rows read as zeros), or -1 on a malformed plan.
 * n_threads > 1 parallelises over columns with OpenMP (CPU baseline). */
int64_t orc_process_feature_columns(const orc_plan_t *plan, const int8_t *blob,
                                    const int32_t *offsets, const int32_t *shapes,
                                    const float *const *tables,
                                    const int32_t *symbols, float *const *group_out,
                                    int32_t n_threads);

/* Rows / width of a group for these run-time shapes (-1 on error). */
/* n_threads independent single-threaded workers (serve_workers), calls_per_thread requests
 * each; returns elapsed seconds (bench.py cpu_baseline). */
int64_t orc_process_feature_columns_unfused(const orc_plan_t *p, const int8_t *blob, const int32_t *offsets,
                                            const int32_t *shapes, const float *const *tables, const int32_t *symbols,
                                            float *const *group_out, float *scratch);
int64_t orc_serve_for_dataflow(const orc_plan_t *p, const int8_t *const *blobs, int32_t n_blobs,
                               const int32_t *const *offsets, const int32_t *const *shapes,
                               const float *const *tables, const int32_t *symbols, int32_t n_threads,
                               double seconds, int32_t dataflow, double *elapsed);
int64_t orc_serve_for(const orc_plan_t *p, const int8_t *const *blobs, int32_t n_blobs,
                      const int32_t *const *offsets, const int32_t *const *shapes,
                      const float *const *tables, const int32_t *symbols, int32_t n_threads,
                      double seconds, double *elapsed);
/* parallel first touch (multi-socket hosts: spreads a table's pages over the sockets) */
void orc_fill_f32(float *p, int64_t n, float v);

double orc_serve_throughput(const orc_plan_t *plan, const int8_t *const *blobs, int32_t n_blobs,
                            const int32_t *const *offsets, const int32_t *const *shapes,
                            const float *const *tables, const int32_t *symbols, int32_t n_threads,
                            int32_t calls_per_thread);
int64_t orc_group_rows(const orc_plan_t *plan, int32_t group,
                       const int32_t *shapes, const int32_t *symbols);
int32_t orc_group_width(const orc_plan_t *plan, int32_t group);
int32_t orc_column_offset(const orc_plan_t *plan, int32_t column);

#ifdef __cplusplus
}
#endif
#endif
