/*
 * fcp_oracle.c — plain-C CPU restatement of the reference's fused
 * feature-column path (AlibabaResearch/recom).  TEST INFRASTRUCTURE ONLY — see
 * fcp_oracle.h for who may use it and for the parity-pinning statement.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference root, tensorflow_addons/ prefix dropped where unambiguous).
 * Integer / index / copy work is restated exactly; floating-point pooling uses
 * sequential fp32 accumulation in id order (the order of the reference-owned
 * plain-C++ segment mean, custom_ops/extended_sparse_segment_reduce/
 * extended_sparse_segment_reduce_ops.cc:53-107; TF-CPU's SparseSegmentReduction
 * functor adds in the same order for bags of up to 9 ids and sums every further 8
 * rows among themselves first — orc_sparse_segment_reduce_tfcpu restates that
 * order), NOT the CUB block-scan tree order of the
 * generated CUDA (graph_optimizers/cuda_emitter.cc:452-455), which cannot be
 * pinned without CUB 1.8 and is only tolerance-comparable anyway.
 *
 * Build: gcc -O2 -fPIC -shared [-fopenmp] fcp_oracle.c -o libfcp_oracle.so
 * (no -ffast-math, no FMA contraction: additions and one division only).
 */
#include "fcp_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* --------------------------------------------------------------------------
 * a5 — Bucketize.  graph_optimizers/cuda_emitter.cc:233-247: binary search,
 * returns r+1 = number of boundaries <= value (upper bound; TF Bucketize).
 * NaN compares false with `<`, so it walks right and lands in bucket n, as the
 * generated code does.
 * ------------------------------------------------------------------------ */
int32_t orc_bucketize(const float *boundaries, int32_t n, float value) {
  int32_t l = 0, r = n - 1;
  while (l <= r) {
    int32_t mid = (l + r) >> 1;
    if (value < boundaries[mid]) {
      r = mid - 1;
    } else {
      l = mid + 1;
    }
  }
  return r + 1;
}

void orc_bucketize_array(const float *boundaries, int32_t n, const float *values,
                         int64_t count, int32_t *out) {
  for (int64_t i = 0; i < count; ++i) out[i] = orc_bucketize(boundaries, n, values[i]);
}

/* --------------------------------------------------------------------------
 * a1 — ConcatInputs.  custom_ops/concat_inputs/concat_inputs_ops.cc:42-77:
 * offsets[i] = running byte size (int32), shapes = all dims in order, blob =
 * byte concatenation (mempcpy per input, :74).  Returns total bytes.
 * ------------------------------------------------------------------------ */
int64_t orc_concat_inputs(const void *const *datas, const int64_t *nbytes,
                          const int32_t *ranks, const int64_t *const *dims,
                          int32_t n, int8_t *blob, int32_t *offsets,
                          int32_t *shapes) {
  int64_t size = 0;
  int32_t *shape_itr = shapes;
  for (int32_t i = 0; i < n; ++i) {
    offsets[i] = (int32_t)size;
    size += nbytes[i];
    for (int32_t j = 0; j < ranks[i]; ++j) *(shape_itr++) = (int32_t)dims[i][j];
  }
  if (blob) {
    int8_t *itr = blob;
    for (int32_t i = 0; i < n; ++i) {
      if (nbytes[i]) memcpy(itr, datas[i], (size_t)nbytes[i]);
      itr += nbytes[i];
    }
  }
  return size;
}

/* --------------------------------------------------------------------------
 * a6 — GatherRows.  cuda_emitter.cc:250-293 (+ driver loop :1305-1327):
 * out[i, :] = table[ids[i], :].  An id outside [0, vocab) is undefined in the
 * reference (out-of-bounds read); here, as in the product, the row reads as
 * zeros and is counted.
 * ------------------------------------------------------------------------ */
int64_t orc_gather_rows(const float *table, int64_t vocab, int32_t dim,
                        const int64_t *ids, int64_t n, float *out,
                        int64_t out_stride) {
  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    float *o = out + i * out_stride;
    int64_t id = ids[i];
    if (id < 0 || id >= vocab) {
      memset(o, 0, sizeof(float) * (size_t)dim);
      ++bad;
    } else {
      memcpy(o, table + id * dim, sizeof(float) * (size_t)dim);
    }
  }
  return bad;
}

/* --------------------------------------------------------------------------
 * a9 — GatherScatterRows.  cuda_emitter.cc:296-345; output pre-zeroed
 * (:1351-1359); out[rows[i], :] = table[ids[i], :].  The matcher guarantees at
 * most one id per row (lookup_optimizer.cc:150-155); if a row repeats, the
 * reference races — here (and in the product) the highest i wins.
 * ------------------------------------------------------------------------ */
int64_t orc_gather_scatter_rows(const float *table, int64_t vocab, int32_t dim,
                                const int64_t *ids, const int64_t *rows,
                                int64_t n, int64_t num_rows, float *out,
                                int64_t out_stride) {
  int64_t bad = 0;
  for (int64_t r = 0; r < num_rows; ++r)
    memset(out + r * out_stride, 0, sizeof(float) * (size_t)dim);
  for (int64_t i = 0; i < n; ++i) {
    int64_t r = rows[i];
    if (r < 0 || r >= num_rows) continue; /* TF ScatterNd-GPU drops bad rows */
    float *o = out + r * out_stride;
    int64_t id = ids[i];
    if (id < 0 || id >= vocab) {
      memset(o, 0, sizeof(float) * (size_t)dim);
      ++bad;
    } else {
      memcpy(o, table + id * dim, sizeof(float) * (size_t)dim);
    }
  }
  return bad;
}

/* --------------------------------------------------------------------------
 * a8 — ComputeSegmentOffsets.  cuda_emitter.cc:768-818.  For sorted segment
 * ids it writes offsets[id] = first position whose segment id >= id, for
 * id in (seg[idx-1], seg[idx]] (inner loop :792-795), with seg[-1] = -1
 * (:785) and seg[n] = num_segments (:786).  Result: CSR offsets[0..S].
 * Ids >= num_segments are clamped into the tail (positions past the last
 * valid segment belong to no output row), ids < 0 behave like segment 0's
 * predecessor exactly as the reference's loop bounds do.
 * ------------------------------------------------------------------------ */
void orc_segment_offsets(const int64_t *seg_ids, int64_t n,
                         int64_t num_segments, int32_t *offsets) {
  int64_t prev = -1;
  for (int64_t idx = 0; idx <= n; ++idx) {
    int64_t cur = idx < n ? seg_ids[idx] : num_segments;
    if (cur > num_segments) cur = num_segments;
    for (int64_t id = prev + 1; id <= cur; ++id) {
      if (id >= 0 && id <= num_segments) offsets[id] = (int32_t)idx;
    }
    if (cur > prev) prev = cur;
  }
}

/* --------------------------------------------------------------------------
 * a7 / a8 — SparseSegment{Sum,Mean}WithNumSegments.
 *   dim <= 20: cuda_emitter.cc:402-501 (sum), :564-661 (mean; sum / counter,
 *              :625), driver :1542-1618, arena pre-zeroed :1460-1475;
 *   dim  > 20: cuda_emitter.cc:831-909 (mean: / float(end-begin), :903).
 * Semantics restated: out[s,:] = sum_{i in [off[s], off[s+1])} W[ids[i],:],
 * divided by the count for mean; a segment with no ids is all zeros (TF-CPU
 * SparseSegment*WithNumSegments; the dim>20 template would produce 0/0 there,
 * the dim<=20 one and TF produce 0 — the TF behaviour is the contract).
 * ------------------------------------------------------------------------ */
int64_t orc_sparse_segment_reduce(const float *table, int64_t vocab, int32_t dim,
                                  const int64_t *ids, const int32_t *offsets,
                                  int64_t num_segments, int32_t mean, float *out,
                                  int64_t out_stride) {
  int64_t bad = 0;
  for (int64_t s = 0; s < num_segments; ++s) {
    float *o = out + s * out_stride;
    int32_t lo = offsets[s], hi = offsets[s + 1];
    for (int32_t e = 0; e < dim; ++e) o[e] = 0.0f;
    for (int32_t i = lo; i < hi; ++i) {
      int64_t id = ids[i];
      if (id < 0 || id >= vocab) {
        ++bad;
        continue;
      }
      const float *w = table + id * dim;
      for (int32_t e = 0; e < dim; ++e) o[e] = o[e] + w[e];
    }
    if (mean && hi > lo) {
      const float cnt = (float)(hi - lo);
      for (int32_t e = 0; e < dim; ++e) o[e] = o[e] / cnt;
    }
  }
  return bad;
}

/* a7 / a8 in the addition order of TENSORFLOW's CPU kernel — what the north star's "vs TF-CPU" tolerance is measured
 * against.  THIRD-PARTY arithmetic, not under /root/reference and not installed here: TensorFlow 2.6.2 (the reference's
 * pin, docs/build_from_source.md:8), tensorflow/core/kernels/segment_reduction_ops_impl.h,
 * SparseSegmentReductionOpBase::Reduce, restated from its published source; PARITY UNPINNED for this function (no
 * TensorFlow to run it against).  For a segment of `num` rows l0 .. l(num-1), per output coefficient:
 *   num == 1:  out = l0                                   (no division, mean or not)
 *   else       r = num & 7;  m = (mean && num < 10) ? float(num) : 1
 *              first chunk = the first r rows for r in 2..7, the first 8 rows for r == 0, the first 9 rows for r == 1:
 *                out = ((..(l0 + l1) + l2 ..) + l(k-1)) / m        (one left-to-right Eigen expression, then the division)
 *              then chunks of 8:  out = out + (((((((la + lb) + lc) + ld) + le) + lf) + lg) + lh)
 *              mean && num >= 10:  out = out / float(num)
 * i.e. bags of up to 9 ids are summed strictly left to right — the same additions as orc_sparse_segment_reduce — and
 * from 10 ids on every further 8 rows are summed among themselves first.  Segments without ids are zeros
 * (SparseSegment*WithNumSegments fills them with the default value 0).  Ids are assumed valid (TF raises otherwise). */
void orc_sparse_segment_reduce_tfcpu(const float *table, int32_t dim, const int64_t *ids, const int32_t *offsets,
                                     int64_t num_segments, int32_t mean, float *out, int64_t out_stride) {
  for (int64_t s = 0; s < num_segments; ++s) {
    float *o = out + s * out_stride;
    const int64_t lo = offsets[s], num = (int64_t)offsets[s + 1] - lo;
    if (num <= 0) {
      for (int32_t e = 0; e < dim; ++e) o[e] = 0.0f;
      continue;
    }
    if (num == 1) {
      const float *w = table + ids[lo] * dim;
      for (int32_t e = 0; e < dim; ++e) o[e] = w[e];
      continue;
    }
    int64_t r = num & 7;
    const float m = (mean && num < 10) ? (float)num : 1.0f;
    if (r == 0) r = 8;
    if (r == 1) r = 9;
    for (int32_t e = 0; e < dim; ++e) {
      float acc = table[ids[lo] * dim + e];
      for (int64_t k = 1; k < r; ++k) acc = acc + table[ids[lo + k] * dim + e];
      acc = acc / m;
      for (int64_t c = r; c < num; c += 8) {
        float t = table[ids[lo + c] * dim + e];
        for (int64_t k = 1; k < 8; ++k) t = t + table[ids[lo + c + k] * dim + e];
        acc = acc + t;
      }
      if (mean && num >= 10) acc = acc / (float)num;
      o[e] = acc;
    }
  }
}

/* a8 in the reference's own order: VBLOCK_DIM_Y = 8 `ty` lanes take rows
 * begin+ty, begin+ty+8, ... (cuda_emitter.cc:868-883), then partial sums are
 * combined by the stride loop :885-899 — (8+1)/2 = 4, 2, 1 — i.e.
 * t[y] += t[y+stride] for y+stride < last_stride. */
void orc_sparse_segment_reduce_ref8x8(const float *table, int32_t dim,
                                      const int64_t *ids, const int32_t *offsets,
                                      int64_t num_segments, int32_t mean,
                                      float *out, int64_t out_stride) {
  enum { Y = 8 };
  float *part = (float *)malloc(sizeof(float) * Y * (size_t)dim);
  for (int64_t s = 0; s < num_segments; ++s) {
    int32_t lo = offsets[s], hi = offsets[s + 1];
    for (int32_t k = 0; k < Y * dim; ++k) part[k] = 0.0f;
    for (int32_t ty = 0; ty < Y; ++ty)
      for (int32_t i = lo + ty; i < hi; i += Y) {
        const float *w = table + ids[i] * dim;
        for (int32_t e = 0; e < dim; ++e) part[ty * dim + e] += w[e];
      }
    for (int32_t stride = (Y + 1) / 2, last = Y; last != stride;
         last = stride, stride = (stride + 1) / 2) {
      for (int32_t ty = 0; ty + stride < last; ++ty)
        for (int32_t e = 0; e < dim; ++e) part[ty * dim + e] += part[(ty + stride) * dim + e];
    }
    float *o = out + s * out_stride;
    for (int32_t e = 0; e < dim; ++e)
      o[e] = (mean && hi > lo) ? part[e] / (float)(hi - lo) : part[e];
  }
  free(part);
}

/* a7 in the reference's OWN summation order for dim <= 20: tiles of 64 ids, a segmented
 * inclusive scan with cub::BlockScan<..., 64, BLOCK_SCAN_WARP_SCANS> (CUB 1.8.0, WORKSPACE:5-14) over
 * (row-vector, head-flag) pairs, cuda_emitter.cc:348-501 (Sum) / :504-661 (Mean), driven tile by tile
 * by EmitSparseSegmentReduce :1542-1618:
 *   - head flag of element i = row[i] - row[i-1] (row[-1] = 0, `last_row_id`, :1576), tail flag =
 *     row[i+1] - row[i] (INT_MAX after the end, :1589-1590);
 *   - ScanVecScanOp (:381-400): op(a, b) = b if b opens a segment, else a + b;
 *   - CUB 1.8 WarpScanShfl::InclusiveScan on each 32-lane warp: for d = 1, 2, 4, 8, 16:
 *     x[i] = op(x[i-d], x[i]) for i >= d (Kogge-Stone); BLOCK_SCAN_WARP_SCANS then folds the first
 *     warp's aggregate into every item of the second: x[i] = op(aggregate0, x[i]);
 *   - a tail lane writes op(last_aggregate, x[i]) (:470-479); the tile's last lane keeps the carry
 *     (:482-490).
 * The slabs of SCAN_DIM = 8 floats (:1609-1618) only partition the row: every float sees the same
 * order, so whole rows are scanned here.  Used only to bound the fp32 reordering error of the HIP path
 * (sequential id order) against what the reference's kernel computes; rows without ids stay zero. */
typedef struct { int32_t key; int32_t cnt; } orc_scan_head_t;

static void refscan_op(const float *av, const orc_scan_head_t *ah, const float *bv, const orc_scan_head_t *bh,
                       float *rv, orc_scan_head_t *rh, int32_t dim) {
  /* may be called with rv == bv (in-place on the right operand) or rv == av */
  orc_scan_head_t h;
  h.key = (int32_t)((uint32_t)ah->key + (uint32_t)bh->key);
  if (bh->key) {
    h.cnt = bh->cnt;
    for (int32_t e = 0; e < dim; ++e) rv[e] = bv[e];
  } else {
    h.cnt = ah->cnt + bh->cnt;
    for (int32_t e = 0; e < dim; ++e) rv[e] = av[e] + bv[e];
  }
  *rh = h;
}

/* assoc selects ONLY the order inside the 64-item inclusive scan; flags, operator, carry and write-out are shared:
 *   ORC_SCAN_CUB18     the reference's: CUB 1.8 WarpScanShfl on 32-lane warps + the first warp's aggregate (above);
 *   ORC_SCAN_ROCPRIM64 rocPRIM's one-wavefront scan (rocprim/warp/detail/warp_scan_dpp.hpp: Kogge-Stone with d = 1, 2, 4, 8
 *                      inside rows of 16 lanes, then lane 15 of each 32 folded into lanes 16..31 (row_bcast:15), then lane 31
 *                      into lanes 32..63 (row_bcast:31)) — what hipCUB's BlockScan<.., 64, BLOCK_SCAN_WARP_SCANS> runs on
 *                      gfx950.  It exists so that the reference's template compiled against hipCUB (oracle/_ref/
 *                      libref_device_scan.so) can be held to this restatement BIT FOR BIT: everything here but the dozen
 *                      lines of the CUB 1.8 scan order is then pinned by the reference's own text. */
void orc_sparse_segment_reduce_refscan_assoc(const float *table, int32_t dim, const int64_t *ids,
                                             const int64_t *row_ids, int64_t nnz, int64_t num_segments,
                                             int32_t mean, int32_t assoc, float *out, int64_t out_stride) {
  enum { BT = 64, WARP = 32, ROW = 16 };
  float *x = (float *)malloc(sizeof(float) * BT * (size_t)dim);
  float *t = (float *)malloc(sizeof(float) * BT * (size_t)dim);
  float *carry = (float *)calloc((size_t)dim, sizeof(float));
  float *tmp = (float *)malloc(sizeof(float) * (size_t)dim);
  orc_scan_head_t h[BT], th[BT], carry_h = {0, 0};
  for (int64_t s = 0; s < num_segments; ++s)
    for (int32_t e = 0; e < dim; ++e) out[s * out_stride + e] = 0.0f; /* the arena is pre-zeroed, :1460-1475 */
  int64_t last_row = 0;
  for (int64_t base = 0; base < nnz; base += BT) {
    int64_t row[BT + 2];
    row[0] = last_row;
    for (int i = 0; i < BT; ++i) row[i + 1] = base + i < nnz ? row_ids[base + i] : INT32_MAX;
    row[BT + 1] = base + BT < nnz ? row_ids[base + BT] : INT32_MAX;
    last_row = row[BT];
    for (int i = 0; i < BT; ++i) {
      h[i].key = (int32_t)(row[i + 1] - row[i]);
      h[i].cnt = 1;
      for (int32_t e = 0; e < dim; ++e)
        x[i * dim + e] = base + i < nnz ? table[ids[base + i] * dim + e] : 0.0f; /* lanes past the end: not read */
    }
    const int group = assoc == ORC_SCAN_ROCPRIM64 ? ROW : WARP;
    for (int w = 0; w < BT / group; ++w) /* Kogge-Stone inside each warp (CUB) / each row of 16 lanes (rocPRIM) */
      for (int d = 1; d < group; d <<= 1) {
        memcpy(t, x, sizeof(float) * BT * (size_t)dim);
        memcpy(th, h, sizeof(h));
        for (int i = w * group + d; i < (w + 1) * group; ++i)
          refscan_op(t + (i - d) * dim, &th[i - d], t + i * dim, &th[i], x + i * dim, &h[i], dim);
      }
    /* fold the aggregate of lane `src` into lanes [lo, hi): x[i] = op(x[src], x[i]) */
#define ORC_FOLD(src, lo, hi)                                         \
  do {                                                                \
    memcpy(tmp, x + (src) * dim, sizeof(float) * (size_t)dim);        \
    const orc_scan_head_t agg = h[(src)];                             \
    for (int i = (lo); i < (hi); ++i) {                               \
      memcpy(t, x + i * dim, sizeof(float) * (size_t)dim);            \
      const orc_scan_head_t hi_ = h[i];                               \
      refscan_op(tmp, &agg, t, &hi_, x + i * dim, &h[i], dim);        \
    }                                                                 \
  } while (0)
    if (assoc == ORC_SCAN_ROCPRIM64) {
      ORC_FOLD(15, 16, 32); /* row_bcast:15, lanes with lane % 32 >= 16 */
      ORC_FOLD(47, 48, 64);
      ORC_FOLD(31, 32, 64); /* row_bcast:31 */
    } else {
      for (int w = 1; w < BT / WARP; ++w) ORC_FOLD(w * WARP - 1, w * WARP, (w + 1) * WARP); /* warp prefixes, in order */
    }
#undef ORC_FOLD
    for (int i = 0; i < BT; ++i) {
      const int tail = row[i + 2] != row[i + 1];
      if (base + i >= nnz) break;
      if (tail || i == BT - 1) {
        orc_scan_head_t rh;
        refscan_op(carry, &carry_h, x + i * dim, &h[i], tmp, &rh, dim);
        if (tail && row[i + 1] >= 0 && row[i + 1] < num_segments) {
          float *o = out + row[i + 1] * out_stride;
          for (int32_t e = 0; e < dim; ++e) o[e] = mean ? tmp[e] / (float)rh.cnt : tmp[e]; /* sum / counter, :625 */
        }
        if (i == BT - 1) { /* :482-490: the pair itself if its segment ended here, else carry (+) pair — the same value */
          memcpy(carry, tmp, sizeof(float) * (size_t)dim);
          carry_h = rh;
        }
      }
    }
  }
  free(x);
  free(t);
  free(carry);
  free(tmp);
}

void orc_sparse_segment_reduce_refscan(const float *table, int32_t dim, const int64_t *ids,
                                       const int64_t *row_ids, int64_t nnz, int64_t num_segments,
                                       int32_t mean, float *out, int64_t out_stride) {
  orc_sparse_segment_reduce_refscan_assoc(table, dim, ids, row_ids, nnz, num_segments, mean, ORC_SCAN_CUB18, out, out_stride);
}

/* --------------------------------------------------------------------------
 * a11 — BatchColReduction.  cuda_emitter.cc:1216-1241 (Sum(x, axis=1) on a
 * rank-3 input): out[b,c] = sum_r x[b,r,c], r ascending (:1231-1236).
 * ------------------------------------------------------------------------ */
void orc_batch_col_reduction(const float *x, int64_t batch, int64_t rows,
                             int64_t cols, float *out, int64_t out_stride) {
  for (int64_t b = 0; b < batch; ++b)
    for (int64_t c = 0; c < cols; ++c) {
      float acc = 0.0f;
      for (int64_t r = 0; r < rows; ++r) acc += x[(b * rows + r) * cols + c];
      out[b * out_stride + c] = acc;
    }
}

/* --------------------------------------------------------------------------
 * a10 — ConcatOutputs.  custom_ops/concat_outputs/concat_outputs_op_gpu.cu.cc
 * :85-99 (ScatterBlock): out[p*sum + off_k + e] = in_k[p*dim_k + e], off_k =
 * prefix sum of dims (:74-79).
 * ------------------------------------------------------------------------ */
void orc_concat_outputs(const float *const *inputs, const int32_t *dims,
                        int32_t n, int64_t prefix, float *out) {
  int64_t sum = 0;
  for (int32_t k = 0; k < n; ++k) sum += dims[k];
  int64_t off = 0;
  for (int32_t k = 0; k < n; ++k) {
    for (int64_t p = 0; p < prefix; ++p)
      memcpy(out + p * sum + off, inputs[k] + p * dims[k], sizeof(float) * (size_t)dims[k]);
    off += dims[k];
  }
}

/* ------------------------------ whole path ------------------------------- */

static int64_t shape_offset(const orc_plan_t *p, int32_t input) {
  int64_t o = 0;
  for (int32_t i = 0; i < input; ++i) o += p->host_input_ranks[i];
  return o;
}

static int64_t input_numel(const orc_plan_t *p, int32_t input, const int32_t *shapes) {
  int64_t o = shape_offset(p, input), n = 1;
  for (int32_t j = 0; j < p->host_input_ranks[input]; ++j) n *= shapes[o + j];
  return n;
}

int64_t orc_group_rows(const orc_plan_t *p, int32_t group, const int32_t *shapes, const int32_t *symbols);

static int64_t column_rows(const orc_plan_t *p, const orc_column_t *c,
                           const int32_t *shapes, const int32_t *symbols) {
  switch (c->rows_source) {
  case ORC_ROWS_FROM_IDS:
    return input_numel(p, c->ids_input, shapes);
  case ORC_ROWS_FROM_SYMBOL:
    return symbols ? symbols[c->rows_arg] : -1;
  case ORC_ROWS_FROM_INPUT_DIM0:
    return shapes[shape_offset(p, c->rows_arg)];
  case ORC_ROWS_FROM_GROUP:
    return orc_group_rows(p, c->concat_group, shapes, symbols);
  default:
    return -1;
  }
}

int64_t orc_group_rows(const orc_plan_t *p, int32_t group, const int32_t *shapes,
                       const int32_t *symbols) {
  int64_t rows = -1;
  for (int32_t k = 0; k < p->n_columns; ++k) {
    if (p->columns[k].concat_group != group || p->columns[k].rows_source == ORC_ROWS_FROM_GROUP) continue;
    int64_t r = column_rows(p, &p->columns[k], shapes, symbols);
    if (r < 0 || (rows >= 0 && r != rows)) return -1;
    rows = r;
  }
  return rows;
}

int32_t orc_group_width(const orc_plan_t *p, int32_t group) {
  int32_t w = 0;
  for (int32_t k = 0; k < p->n_columns; ++k)
    if (p->columns[k].concat_group == group) w += p->columns[k].dim;
  return w;
}

int32_t orc_column_offset(const orc_plan_t *p, int32_t column) {
  const orc_column_t *c = &p->columns[column];
  int32_t off = 0;
  for (int32_t k = 0; k < p->n_columns; ++k)
    if (p->columns[k].concat_group == c->concat_group &&
        p->columns[k].concat_slot < c->concat_slot)
      off += p->columns[k].dim;
  return off;
}

/* Materialise the id stream of a column as int64 (the index expression the
 * reference inlines: Cast :1788-1797, Bucketize :1798-1835). */
static void load_ids(const orc_column_t *c, const int8_t *src, int64_t n, int64_t *ids) {
  for (int64_t i = 0; i < n; ++i) {
    if (c->id_source == ORC_IDS_I32) {
      int32_t v;
      memcpy(&v, src + 4 * i, 4);
      ids[i] = v;
    } else if (c->id_source == ORC_IDS_I64) {
      int64_t v;
      memcpy(&v, src + 8 * i, 8);
      ids[i] = v;
    } else {
      float v;
      memcpy(&v, src + 4 * i, 4);
      ids[i] = orc_bucketize(c->boundaries, c->n_boundaries, v);
    }
  }
}

/* ---- Fingerprint64 (FarmHash farmhashna::Hash64) for len <= 32 ------------------------------------- */
static uint64_t fp_fetch64(const char *p) { uint64_t v; memcpy(&v, p, 8); return v; }   /* little endian hosts */
static uint32_t fp_fetch32(const char *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t fp_rot(uint64_t v, int s) { return s ? (v >> s) | (v << (64 - s)) : v; }
static uint64_t fp_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul;
  a ^= a >> 47;
  uint64_t b = (v ^ a) * mul;
  b ^= b >> 47;
  return b * mul;
}
uint64_t orc_fingerprint64(const char *s, size_t len) {
  const uint64_t k0 = 0xc3a5c85c97cb3127ULL, k1 = 0xb492b66fbe98f273ULL, k2 = 0x9ae16a3b2f90404fULL;
  if (len <= 16) { /* HashLen0to16 */
    if (len >= 8) {
      uint64_t mul = k2 + len * 2, a = fp_fetch64(s) + k2, b = fp_fetch64(s + len - 8);
      return fp_len16(fp_rot(b, 37) * mul + a, (fp_rot(a, 25) + b) * mul, mul);
    }
    if (len >= 4) {
      uint64_t mul = k2 + len * 2, a = fp_fetch32(s);
      return fp_len16(len + (a << 3), fp_fetch32(s + len - 4), mul);
    }
    if (len > 0) {
      uint8_t a = (uint8_t)s[0], b = (uint8_t)s[len >> 1], c = (uint8_t)s[len - 1];
      uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)len + ((uint32_t)c << 2);
      uint64_t h = (uint64_t)y * k2 ^ (uint64_t)z * k0;
      h ^= h >> 47;
      return h * k2;
    }
    return k2;
  }
  if (len <= 32) { /* HashLen17to32 */
    uint64_t mul = k2 + len * 2, a = fp_fetch64(s) * k1, b = fp_fetch64(s + 8), c = fp_fetch64(s + len - 8) * mul,
             d = fp_fetch64(s + len - 16) * k2;
    return fp_len16(fp_rot(a + b, 43) + fp_rot(c, 30) + d, a + fp_rot(b + k2, 18) + c, mul);
  }
  return 0; /* longer strings are not on this path (decimal int64: at most 20 bytes) */
}

int64_t orc_hash_bucket_int64(int64_t value, int64_t num_buckets) {
  char buf[32];
  int n = snprintf(buf, sizeof buf, "%lld", (long long)value); /* AsString: plain decimal */
  return (int64_t)(orc_fingerprint64(buf, (size_t)n) % (uint64_t)num_buckets);
}

/* The interval test of the id-filter ops (select_value_ops.cc:35-41 and siblings), as intended. */
static int xform_in(const orc_column_t *c, int64_t x) {
  for (int32_t i = 0; i < c->xform_n; ++i)
    if (x >= c->xform_lo[i] && x <= c->xform_hi[i]) return 1;
  return 0;
}

static int64_t load_index(const orc_column_t *c, const int8_t *src, int64_t e) {
  if (c->seg_kind == ORC_SEG_IDS_I32) {
    int32_t v;
    memcpy(&v, src + 4 * e, 4);
    return v;
  }
  int64_t v;
  memcpy(&v, src + 8 * e, 8);
  return v;
}

/* Segment ids: element i*seg_stride of the tensor (SparseTensor indices[:, 0]: cuda_emitter.cc:1836-1873), or — with a
 * SparseReshape folded in (cuda_emitter.cc:1874-1916) — the row coordinate of the reshaped element. */
static void load_seg(const orc_column_t *c, const int8_t *src, int64_t n, const int32_t *symbols, int64_t *seg) {
  for (int64_t i = 0; i < n; ++i) {
    if (c->seg_map_n <= 0) {
      seg[i] = load_index(c, src, i * c->seg_stride);
      continue;
    }
    const int64_t sym = c->seg_map_sym >= 0 ? symbols[c->seg_map_sym] : 1;
    int64_t lin = 0, neg = 0;
    for (int32_t k = 0; k < c->seg_map_n; ++k) {
      const int64_t v = load_index(c, src, i * c->seg_stride + k);
      if (v < 0) neg = 1;
      lin += v * (c->seg_map_sym_slot == k ? c->seg_map_mul[k] * sym : c->seg_map_mul[k]);
    }
    const int64_t div = c->seg_map_sym_slot == 4 ? c->seg_map_div * sym : c->seg_map_div;
    seg[i] = neg ? -1 : lin / div;
  }
}

/* Row sharding (no reference counterpart): this rank owns ids
 * with id % world == rank, stored at local row id / world.  Non-owned ids
 * contribute nothing; pooled columns emit partial SUMS (the mean division
 * happens after the cross-rank reduction). */
static void shard_ids(const orc_plan_t *p, int64_t *ids, int64_t n, int64_t vocab_global) {
  if (p->shard_world <= 1) return;
  for (int64_t i = 0; i < n; ++i) {
    int64_t id = ids[i];
    if (id < 0 || id >= vocab_global) continue; /* stays bad */
    ids[i] = (id % p->shard_world == p->shard_rank) ? id / p->shard_world : INT64_MIN;
  }
}

static int64_t process_column(const orc_plan_t *p, int32_t k, const int8_t *blob,
                              const int32_t *offsets, const int32_t *shapes,
                              const float *const *tables, const int32_t *symbols,
                              float *const *group_out, const int32_t *widths,
                              const int32_t *col_offs) {
  const orc_column_t *c = &p->columns[k];
  if (c->form == ORC_FORM_EXTERNAL) return 0; /* filled by ConcatOutputs from a host input: not ours to write */
  const int32_t width = widths[c->concat_group];
  const int32_t off = col_offs[k];
  const int64_t rows = column_rows(p, c, shapes, symbols);
  float *out = group_out[c->concat_group] + off;
  int64_t bad = 0;
  const int sharded = p->shard_world > 1;
  /* local vocab of the shard */
  const int64_t vocab_local =
      sharded ? (c->vocab - p->shard_rank + p->shard_world - 1) / p->shard_world : c->vocab;

  if (sharded && p->shard_rank != 0 &&
      (c->form == ORC_FORM_PASSTHROUGH || c->form == ORC_FORM_BATCH_COL_REDUCTION)) {
    /* table-free columns are owned by rank 0; the others contribute zeros */
    for (int64_t r = 0; r < rows; ++r) memset(out + r * width, 0, 4 * (size_t)c->dim);
    return 0;
  }
  if (c->form == ORC_FORM_PASSTHROUGH) {
    const int8_t *src = blob + offsets[c->ids_input];
    for (int64_t r = 0; r < rows; ++r)
      memcpy(out + r * width, src + 4 * r * c->dim, 4 * (size_t)c->dim);
    return 0;
  }
  if (c->form == ORC_FORM_BATCH_COL_REDUCTION) {
    const int64_t so = shape_offset(p, c->ids_input);
    const int64_t R = shapes[so + 1];
    float *x = (float *)malloc(sizeof(float) * (size_t)(rows * R * c->dim) + 4);
    memcpy(x, blob + offsets[c->ids_input], sizeof(float) * (size_t)(rows * R * c->dim));
    orc_batch_col_reduction(x, rows, R, c->dim, out, width);
    free(x);
    return 0;
  }

  int64_t nnz = input_numel(p, c->ids_input, shapes);
  int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
  load_ids(c, blob + offsets[c->ids_input], nnz, ids);
  /* the CPU id ops that sit in front of the lookup, applied here as they are in the graph.
   * SelectValue: elementwise.  GatherIndiceValue / GatherValueGenIndice: the (index, value) pairs that fail
   * the test are removed — `keep` marks the survivors; the segment structure is compacted below. */
  if (c->hash_buckets > 0)
    for (int64_t i = 0; i < nnz; ++i) ids[i] = orc_hash_bucket_int64(ids[i], c->hash_buckets);
  uint8_t *keep = NULL;
  if (c->xform_mode == ORC_XFORM_SELECT) {
    for (int64_t i = 0; i < nnz; ++i)
      if (!xform_in(c, ids[i])) ids[i] = c->xform_substitute;
  } else if (c->xform_mode == ORC_XFORM_FILTER) {
    keep = (uint8_t *)malloc((size_t)nnz + 1);
    for (int64_t i = 0; i < nnz; ++i) keep[i] = (uint8_t)xform_in(c, ids[i]);
  }
  /* ScatterNd columns (form 3) count differently: of the ids written to one row only the one that stays (the last) is
   * "an id that reached the output", so only winners outside the vocabulary are counted, plus the row ids outside
   * [0, rows) that ScatterNd drops — `gbad` remembers which ids are globally bad before the shard mapping. */
  const int scatter = c->form == ORC_FORM_GATHER_SCATTER;
  uint8_t *gbad = NULL;
  if (scatter) {
    gbad = (uint8_t *)malloc((size_t)nnz + 1);
    for (int64_t i = 0; i < nnz; ++i) gbad[i] = (uint8_t)(ids[i] < 0 || ids[i] >= c->vocab);
  }
  /* count globally-bad ids once, then map to the local shard */
  if (sharded) {
    if (!scatter)
      for (int64_t i = 0; i < nnz; ++i)
        if (ids[i] < 0 || ids[i] >= c->vocab) ++bad;
    shard_ids(p, ids, nnz, c->vocab);
  }
  const float *table = tables[c->table_input];

  if (c->form == ORC_FORM_GATHER) {
    /* GatherValueGenIndice emits index [i] for every surviving value i and the lookup scatters the rows
     * back (form 3); with one value per row that is: a dropped value leaves a zero row */
    int64_t b = orc_gather_rows(table, vocab_local, c->dim, ids, nnz, out, width);
    if (!sharded) bad += b;
    if (keep)
      for (int64_t i = 0; i < nnz; ++i)
        if (!keep[i]) {
          memset(out + i * width, 0, 4 * (size_t)c->dim);
          if (!sharded && (ids[i] < 0 || ids[i] >= vocab_local)) --bad; /* never reached the lookup */
        }
  } else {
    int64_t *seg = NULL;
    int32_t *offs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(rows + 1));
    if (c->seg_kind == ORC_SEG_CSR_I32) {
      memcpy(offs, blob + offsets[c->seg_input], sizeof(int32_t) * (size_t)(rows + 1));
    } else {
      seg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
      load_seg(c, blob + offsets[c->seg_input], nnz, symbols, seg);
      orc_segment_offsets(seg, nnz, rows, offs);
    }
    if (keep && scatter && seg) {
      /* ScatterNd takes its row ids in any order (GatherScatterRows, cuda_emitter.cc:296-345): the filter op in front
       * of it removes (row id, id) pairs by position, nothing is assumed sorted */
      int64_t w = 0;
      for (int64_t i = 0; i < nnz; ++i)
        if (keep[i]) {
          ids[w] = ids[i];
          seg[w] = seg[i];
          gbad[w] = gbad[i];
          ++w;
        }
      nnz = w;
    } else if (keep) { /* compact ids (and row ids) in place; the row offsets follow the survivors */
      int64_t w = 0, src = 0;
      for (int64_t r = 0; r < rows; ++r) {
        int64_t lo = offs[r] < 0 ? 0 : offs[r], hi = offs[r + 1] > nnz ? nnz : offs[r + 1];
        if (lo < src) lo = src;
        offs[r] = (int32_t)w;
        for (int64_t i = lo; i < hi; ++i)
          if (keep[i]) {
            ids[w] = ids[i];
            if (seg) seg[w] = seg[i];
            if (gbad) gbad[w] = gbad[i];
            ++w;
          }
        if (hi > src) src = hi;
      }
      offs[rows] = (int32_t)w;
      nnz = w;
    }
    if (c->form == ORC_FORM_SEGMENT_REDUCE) {
      const int32_t mean = (c->combiner == ORC_COMBINER_MEAN) && !sharded;
      int64_t b = orc_sparse_segment_reduce(table, vocab_local, c->dim, ids, offs, rows, mean,
                                            out, width);
      if (!sharded) bad += b;
    } else { /* ORC_FORM_GATHER_SCATTER */
      if (seg) {
        (void)orc_gather_scatter_rows(table, vocab_local, c->dim, ids, seg, nnz, rows, out, width);
        /* the counter: rows ScatterNd drops, and winners (the last write to a row) outside the vocabulary */
        int64_t *winner = (int64_t *)malloc(sizeof(int64_t) * (size_t)(rows + 1));
        for (int64_t r = 0; r < rows; ++r) winner[r] = -1;
        for (int64_t i = 0; i < nnz; ++i) {
          if (seg[i] < 0 || seg[i] >= rows) ++bad;
          else winner[seg[i]] = i;
        }
        for (int64_t r = 0; r < rows; ++r)
          if (winner[r] >= 0 && gbad[winner[r]]) ++bad;
        free(winner);
      } else {
        /* CSR form of "at most one id per row": last id of the row wins */
        for (int64_t r = 0; r < rows; ++r) {
          float *o = out + r * width;
          memset(o, 0, 4 * (size_t)c->dim);
          if (offs[r + 1] > offs[r]) {
            int64_t id = ids[offs[r + 1] - 1];
            if (id >= 0 && id < vocab_local) memcpy(o, table + id * c->dim, 4 * (size_t)c->dim);
            if (gbad[offs[r + 1] - 1]) ++bad;
          }
        }
      }
    }
    free(seg);
    free(offs);
  }
  free(keep);
  free(gbad);
  free(ids);
  return bad;
}

/* (group, slot)-ordered prefix sums of dims: O(n log n) once per call.  The sort keys are copied next to the column
 * index so that the comparator needs no context: until round 6 it read the plan through a global under
 * `omp critical`, i.e. every request of every serving worker queued for one lock — bench.py's cpu_baseline fell from
 * 516 k inferences/s at 16 workers to 151 k at 256. */
typedef struct { int32_t group, slot, k; } orc_slot_key_t;
static int cmp_group_slot(const void *a, const void *b) {
  const orc_slot_key_t *x = (const orc_slot_key_t *)a, *y = (const orc_slot_key_t *)b;
  if (x->group != y->group) return x->group < y->group ? -1 : 1;
  if (x->slot != y->slot) return x->slot < y->slot ? -1 : 1;
  return x->k < y->k ? -1 : x->k > y->k;
}

static void layout_columns(const orc_plan_t *p, int32_t *widths, int32_t *col_offs) {
  orc_slot_key_t *order = (orc_slot_key_t *)malloc(sizeof(orc_slot_key_t) * (size_t)(p->n_columns + 1));
  for (int32_t k = 0; k < p->n_columns; ++k) {
    order[k].group = p->columns[k].concat_group;
    order[k].slot = p->columns[k].concat_slot;
    order[k].k = k;
  }
  qsort(order, (size_t)p->n_columns, sizeof(orc_slot_key_t), cmp_group_slot);
  for (int32_t g = 0; g < p->n_groups; ++g) widths[g] = 0;
  for (int32_t j = 0; j < p->n_columns; ++j) {
    const orc_column_t *c = &p->columns[order[j].k];
    col_offs[order[j].k] = widths[c->concat_group];
    widths[c->concat_group] += c->dim;
  }
  free(order);
}

int64_t orc_process_feature_columns(const orc_plan_t *p, const int8_t *blob,
                                    const int32_t *offsets, const int32_t *shapes,
                                    const float *const *tables,
                                    const int32_t *symbols, float *const *group_out,
                                    int32_t n_threads) {
  for (int32_t g = 0; g < p->n_groups; ++g)
    if (orc_group_rows(p, g, shapes, symbols) < 0) return -1;
  int32_t *widths = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->n_groups + 1));
  int32_t *col_offs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->n_columns + 1));
  layout_columns(p, widths, col_offs);
  int64_t bad = 0;
#ifdef _OPENMP
  if (n_threads > 1) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : bad)
    for (int32_t k = 0; k < p->n_columns; ++k)
      bad += process_column(p, k, blob, offsets, shapes, tables, symbols, group_out, widths,
                            col_offs);
    free(widths);
    free(col_offs);
    return bad;
  }
#endif
  (void)n_threads;
  for (int32_t k = 0; k < p->n_columns; ++k)
    bad += process_column(p, k, blob, offsets, shapes, tables, symbols, group_out, widths,
                          col_offs);
  free(widths);
  free(col_offs);
  return bad;
}

/* The same result through TENSORFLOW-CPU'S DATAFLOW for the unrewritten graph: every column's op (GatherV2 / Bucketize +
 * GatherV2 / SparseSegment* / ScatterNd; lookup_optimizer.cc:157-440 are the patterns) produces its own contiguous
 * [rows, dim] tensor, then ConcatV2 copies the tensors row by row into [rows, sum dim].  process_column runs with a
 * one-column "group" of width dim; the values are those of orc_process_feature_columns bit for bit (the concat is a copy).
 * What bench.py's cpu_baseline times since round 6: writing every column straight into the concat matrix — the FUSED
 * layout, this oracle's checker form — visits each 64-byte output line from several columns at a 4 x sum(dim) stride,
 * which is fine while a worker's matrix stays in its share of the L3 and collapses beyond it (2 x 64-core host, S2:
 * 516 k inferences/s at 16 workers, one per CCD, 151 k at 256; the same with cache-resident tables:
 * profiles/r06_cpu_baseline_collapse.txt).  `scratch`: sum over columns of rows x dim floats.  Single-threaded. */
int64_t orc_process_feature_columns_unfused(const orc_plan_t *p, const int8_t *blob, const int32_t *offsets,
                                            const int32_t *shapes, const float *const *tables, const int32_t *symbols,
                                            float *const *group_out, float *scratch) {
  for (int32_t g = 0; g < p->n_groups; ++g)
    if (orc_group_rows(p, g, shapes, symbols) < 0) return -1;
  int32_t *widths = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->n_groups + 1));
  int32_t *col_offs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->n_columns + 1));
  int32_t *w1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(p->n_groups + 1));
  int32_t *o1 = (int32_t *)calloc((size_t)(p->n_columns + 1), sizeof(int32_t));
  float **t1 = (float **)malloc(sizeof(float *) * (size_t)(p->n_groups + 1));
  float **col_t = (float **)malloc(sizeof(float *) * (size_t)(p->n_columns + 1));
  layout_columns(p, widths, col_offs);
  int64_t bad = 0;
  float *cur = scratch;
  for (int32_t k = 0; k < p->n_columns; ++k) { /* the column ops */
    const orc_column_t *c = &p->columns[k];
    const int64_t rows = orc_group_rows(p, c->concat_group, shapes, symbols);
    col_t[k] = cur;
    cur += rows * c->dim;
    w1[c->concat_group] = c->dim;
    t1[c->concat_group] = col_t[k];
    bad += process_column(p, k, blob, offsets, shapes, tables, symbols, t1, w1, o1);
  }
  for (int32_t g = 0; g < p->n_groups; ++g) { /* ConcatV2: row by row, inputs in slot order */
    const int64_t rows = orc_group_rows(p, g, shapes, symbols);
    for (int64_t r = 0; r < rows; ++r) {
      float *dst = group_out[g] + r * widths[g];
      for (int32_t k = 0; k < p->n_columns; ++k) {
        const orc_column_t *c = &p->columns[k];
        if (c->concat_group != g || c->form == ORC_FORM_EXTERNAL) continue;
        memcpy(dst + col_offs[k], col_t[k] + r * c->dim, 4 * (size_t)c->dim);
      }
    }
  }
  free(widths);
  free(col_offs);
  free(w1);
  free(o1);
  free(t1);
  free(col_t);
  return bad;
}

/* Serving-style CPU throughput (what TF-CPU does with many Session::Run threads, the
 * reference harness' serve_workers, examples/cc/recom_examples.patch:210-216): n_threads
 * workers each process whole requests on their own, single-threaded, into private
 * outputs; no synchronisation inside the timed region.  Requests rotate over `n_blobs`
 * requests (blob + offsets + shapes; equal row counts, nnz may differ) so that the touched
 * table rows are not cache-resident.
 * Returns the elapsed seconds for n_threads * calls_per_thread requests, or -1. */
/* The same protocol for a fixed DURATION: every worker serves requests until `seconds` have passed (the clock is read
 * between requests), so that every worker count of a sweep is measured for the same time.  Returns the number of
 * requests all workers completed, *elapsed = the wall time from the common start to the last worker's finish, or -1. */
int64_t orc_serve_for(const orc_plan_t *p, const int8_t *const *blobs, int32_t n_blobs,
                      const int32_t *const *offsets, const int32_t *const *shapes,
                      const float *const *tables, const int32_t *symbols, int32_t n_threads,
                      double seconds, double *elapsed) {
  return orc_serve_for_dataflow(p, blobs, n_blobs, offsets, shapes, tables, symbols, n_threads, seconds, 0, elapsed);
}

/* dataflow 0: every column straight into the concat matrix (orc_process_feature_columns); 1: TF-CPU's dataflow, column
 * tensors + ConcatV2 (orc_process_feature_columns_unfused). */
int64_t orc_serve_for_dataflow(const orc_plan_t *p, const int8_t *const *blobs, int32_t n_blobs,
                               const int32_t *const *offsets, const int32_t *const *shapes,
                               const float *const *tables, const int32_t *symbols, int32_t n_threads,
                               double seconds, int32_t dataflow, double *elapsed) {
#ifdef _OPENMP
  if (n_threads < 1 || !(seconds > 0.0) || n_blobs < 1 || !elapsed) return -1;
  for (int32_t r = 0; r < n_blobs; ++r)
    for (int32_t g = 0; g < p->n_groups; ++g)
      if (orc_group_rows(p, g, shapes[r], symbols) < 0 ||
          orc_group_rows(p, g, shapes[r], symbols) != orc_group_rows(p, g, shapes[0], symbols))
        return -1; /* requests may differ in nnz, not in rows */
  double t0 = 0.0, t1 = 0.0;
  int64_t total = 0;
  int failed = 0;
#pragma omp parallel num_threads(n_threads) reduction(+ : total)
  {
    float **out = (float **)malloc(sizeof(float *) * (size_t)p->n_groups);
    int ok = out != NULL;
    for (int32_t g = 0; ok && g < p->n_groups; ++g) {
      const size_t n = (size_t)orc_group_rows(p, g, shapes[0], symbols) * (size_t)orc_group_width(p, g);
      out[g] = (float *)calloc(n ? n : 1, sizeof(float));
      if (!out[g]) ok = 0;
    }
    float *scratch = NULL;
    if (ok && dataflow) {
      size_t n = 0;
      for (int32_t k = 0; k < p->n_columns; ++k)
        n += (size_t)orc_group_rows(p, p->columns[k].concat_group, shapes[0], symbols) * (size_t)p->columns[k].dim;
      scratch = (float *)calloc(n ? n : 1, sizeof(float));
      if (!scratch) ok = 0;
    }
    int32_t r = (int32_t)(((int64_t)omp_get_thread_num() * 7) % n_blobs);
    if (ok) { /* warm */
      if (dataflow) (void)orc_process_feature_columns_unfused(p, blobs[r], offsets[r], shapes[r], tables, symbols, out, scratch);
      else (void)orc_process_feature_columns(p, blobs[r], offsets[r], shapes[r], tables, symbols, out, 1);
    }
#pragma omp barrier
#pragma omp master
    t0 = omp_get_wtime();
#pragma omp barrier
    if (ok) {
      const double stop = t0 + seconds;
      do {
        r = (r + 1) % n_blobs;
        if (dataflow) (void)orc_process_feature_columns_unfused(p, blobs[r], offsets[r], shapes[r], tables, symbols, out, scratch);
        else (void)orc_process_feature_columns(p, blobs[r], offsets[r], shapes[r], tables, symbols, out, 1);
        ++total;
      } while (omp_get_wtime() < stop);
    }
#pragma omp barrier
#pragma omp master
    t1 = omp_get_wtime();
    if (!ok) {
#pragma omp atomic write
      failed = 1;
    }
    if (out) {
      for (int32_t g = 0; g < p->n_groups; ++g) free(out[g]);
      free(out);
    }
    free(scratch);
  }
  *elapsed = t1 - t0;
  return failed ? -1 : total;
#else
  (void)p; (void)blobs; (void)n_blobs; (void)offsets; (void)shapes; (void)tables; (void)symbols; (void)n_threads; (void)seconds; (void)dataflow; (void)elapsed;
  return -1;
#endif
}

/* First-touch helper for the CPU baseline: fills p[0..n) from all cores at once (static schedule), so that on a multi-socket
 * host the pages of a table are spread over the sockets' memory in contiguous shares instead of all landing next to the
 * one thread that wrote them (bench.py cpu_baseline: 24 GB of tables read at random by up to 256 workers). */
void orc_fill_f32(float *p, int64_t n, float v) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) p[i] = v;
}

double orc_serve_throughput(const orc_plan_t *p, const int8_t *const *blobs, int32_t n_blobs,
                            const int32_t *const *offsets, const int32_t *const *shapes,
                            const float *const *tables, const int32_t *symbols, int32_t n_threads,
                            int32_t calls_per_thread) {
#ifdef _OPENMP
  if (n_threads < 1 || calls_per_thread < 1 || n_blobs < 1) return -1.0;
  for (int32_t r = 0; r < n_blobs; ++r)
    for (int32_t g = 0; g < p->n_groups; ++g)
      if (orc_group_rows(p, g, shapes[r], symbols) < 0 ||
          orc_group_rows(p, g, shapes[r], symbols) != orc_group_rows(p, g, shapes[0], symbols))
        return -1.0; /* requests may differ in nnz, not in rows */
  double t0 = 0.0, t1 = 0.0;
  int failed = 0;
#pragma omp parallel num_threads(n_threads)
  {
    float **out = (float **)malloc(sizeof(float *) * (size_t)p->n_groups);
    int ok = out != NULL;
    for (int32_t g = 0; ok && g < p->n_groups; ++g) {
      const size_t n = (size_t)orc_group_rows(p, g, shapes[0], symbols) * (size_t)orc_group_width(p, g);
      out[g] = (float *)calloc(n ? n : 1, sizeof(float));
      if (!out[g]) ok = 0;
    }
    const int32_t first = (int32_t)(((int64_t)omp_get_thread_num() * calls_per_thread) % n_blobs);
    if (ok) (void)orc_process_feature_columns(p, blobs[first], offsets[first], shapes[first], tables, symbols, out, 1); /* warm */
#pragma omp barrier
#pragma omp master
    t0 = omp_get_wtime();
    if (ok)
      for (int32_t i = 0; i < calls_per_thread; ++i)
      {
          const int32_t r = (first + 1 + i) % n_blobs;
          (void)orc_process_feature_columns(p, blobs[r], offsets[r], shapes[r], tables, symbols, out, 1);
        }
#pragma omp barrier
#pragma omp master
    t1 = omp_get_wtime();
    if (!ok) {
#pragma omp atomic write
      failed = 1;
    }
    if (out) {
      for (int32_t g = 0; g < p->n_groups; ++g) free(out[g]);
      free(out);
    }
  }
  return failed ? -1.0 : t1 - t0;
#else
  (void)p; (void)blobs; (void)n_blobs; (void)offsets; (void)shapes; (void)tables; (void)symbols; (void)n_threads; (void)calls_per_thread;
  return -1.0;
#endif
}

