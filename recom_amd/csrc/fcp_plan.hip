// fcp_plan.hip — plan creation (the non-codegen half of the reference's CudaEmitter), const buffers (CreateConstBuffers,
// cuda_emitter.cc:2260-2301), shape-dependent column records + launch geometry, plan files, the placement gate, accessors.
// Carved out of fcp_api.hip in round 6 (see fcp_host.h); the original header comment follows.
//
// (was fcp_api.hip) — host side of libfcp_hip.so: the C ABI declared in
// include/fcp_hip.h.  Plan building (the non-codegen half of the reference's
// CudaEmitter), const buffers (CreateConstBuffers, cuda_emitter.cc:2260-2301),
// the per-request entry (ProcessFeatureColumns, :2303-2494, and its kernel
// caller :2139-2258) and the host packer of Addons>ConcatInputs
// (custom_ops/concat_inputs/concat_inputs_ops.cc:42-77).
//
// Differences from the reference's per-request host work, on purpose:
//   * no blocking stream synchronisation anywhere on the request path (the
//     reference blocks at :2246 and twice more in the ops);
//   * no per-request H2D of a pageable argument struct (:2216): the shape-
//     dependent descriptors are cached on the device keyed by the request's
//     (offsets, shapes, symbols) and re-uploaded from pinned memory only when
//     the shapes change;
//   * 64-bit byte offsets and row offsets (the reference's int arithmetic
//     overflows beyond 2^31).
#include "fcp_host.h"

namespace fcph {

thread_local std::string g_last_error;

int hip_fail(const char *what, hipError_t e) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorNoBinaryForGpu ||
          e == hipErrorInsufficientDriver)
             ? FCP_ERR_NO_DEVICE
             : FCP_ERR_HIP;
}

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

} // namespace fcph

// failure reporting for the library's other translation units (fcp_shard.hip)
int fcp_internal_fail(int code, const std::string &msg) { return fail(code, msg); }

namespace fcph {

int validate_desc(const fcp_plan_desc_t *d) {
  if (!d) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan descriptor");
  if (d->abi_version != FCP_ABI_VERSION) return fail(FCP_ERR_INVALID_ARGUMENT, "abi_version mismatch");
  if (d->n_columns <= 0 || !d->columns) return fail(FCP_ERR_INVALID_ARGUMENT, "plan has no columns");
  if (d->n_host_inputs < 0 || (d->n_host_inputs > 0 && (!d->host_input_ranks || !d->host_input_elem_sizes)))
    return fail(FCP_ERR_INVALID_ARGUMENT, "host input attrs missing");
  if (d->n_groups <= 0 || d->n_groups > FCP_MAX_GROUPS)
    return fail(FCP_ERR_INVALID_ARGUMENT, "n_groups must be in [1, 16]");
  if (d->layout != FCP_LAYOUT_CONCAT && d->layout != FCP_LAYOUT_PER_COLUMN)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad layout");
  if (d->shard_world < 1 || d->shard_rank < 0 || d->shard_rank >= d->shard_world)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad shard rank/world");
  for (int i = 0; i < d->n_host_inputs; ++i) {
    if (d->host_input_ranks[i] < 0 || d->host_input_ranks[i] > 8)
      return fail(FCP_ERR_INVALID_ARGUMENT, "host input rank out of range");
    if (d->host_input_elem_sizes[i] <= 0) return fail(FCP_ERR_INVALID_ARGUMENT, "bad element size");
  }
  for (int k = 0; k < d->n_columns; ++k) {
    const fcp_column_desc_t &c = d->columns[k];
    const std::string where = "column " + std::to_string(k) + ": ";
    if (c.form < FCP_FORM_GATHER || c.form > FCP_FORM_EXTERNAL)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad form");
    if (c.dim <= 0) return fail(FCP_ERR_INVALID_ARGUMENT, where + "dim must be positive");
    if (c.concat_group < 0 || c.concat_group >= d->n_groups)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "concat_group out of range");
    if (c.form == FCP_FORM_EXTERNAL) {
      // a slot reserved for an Addons>ConcatOutputs host input: no inputs of its own
      if (d->layout != FCP_LAYOUT_CONCAT) return fail(FCP_ERR_INVALID_ARGUMENT, where + "external slots need FCP_LAYOUT_CONCAT");
      if (c.rows_source != FCP_ROWS_FROM_GROUP) return fail(FCP_ERR_INVALID_ARGUMENT, where + "external slot takes its rows from its group");
      for (int j = 0; j < k; ++j)
        if (d->columns[j].concat_group == c.concat_group && d->columns[j].concat_slot == c.concat_slot)
          return fail(FCP_ERR_INVALID_ARGUMENT, where + "duplicate concat slot");
      continue;
    }
    if (c.rows_source == FCP_ROWS_FROM_GROUP) return fail(FCP_ERR_INVALID_ARGUMENT, where + "only external slots take their rows from the group");
    if (c.ids_input < 0 || c.ids_input >= d->n_host_inputs)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "ids_input out of range");
    const bool lookup = c.form == FCP_FORM_GATHER || c.form == FCP_FORM_SEGMENT_REDUCE ||
                        c.form == FCP_FORM_GATHER_SCATTER;
    if (lookup) {
      if (c.vocab <= 0) return fail(FCP_ERR_INVALID_ARGUMENT, where + "vocab must be positive");
      if (c.table_input < 0 || c.table_input >= d->n_device_inputs)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "table_input out of range");
      if (c.id_source < FCP_IDS_I32 || c.id_source > FCP_IDS_F32_BUCKETIZE)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad id_source");
      if (c.id_source == FCP_IDS_F32_BUCKETIZE && (c.n_boundaries <= 0 || !c.boundaries))
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "bucketize needs boundaries");
      const int esz = d->host_input_elem_sizes[c.ids_input];
      if (esz != (c.id_source == FCP_IDS_I64 ? 8 : 4))
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "ids element size does not match id_source");
    } else if (d->host_input_elem_sizes[c.ids_input] != 4) {
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "payload must be a 4-byte type");
    }
    if (c.hash_buckets != 0) {
      if (!lookup) return fail(FCP_ERR_INVALID_ARGUMENT, where + "id transforms apply to lookup columns only");
      if (c.hash_buckets < 0) return fail(FCP_ERR_INVALID_ARGUMENT, where + "hash_buckets must be positive");
      if (c.id_source == FCP_IDS_F32_BUCKETIZE) return fail(FCP_ERR_INVALID_ARGUMENT, where + "hash_buckets applies to integer ids");
    }
    if (c.xform_mode != FCP_XFORM_NONE) {
      if (!lookup) return fail(FCP_ERR_INVALID_ARGUMENT, where + "id transforms apply to lookup columns only");
      if (c.xform_mode != FCP_XFORM_SELECT && c.xform_mode != FCP_XFORM_FILTER)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad xform_mode");
      if (c.xform_n < 0 || c.xform_n > (1 << 20) || (c.xform_n > 0 && (!c.xform_lo || !c.xform_hi)))
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad id transform intervals");
      for (int i = 0; i < c.xform_n; ++i)
        if (c.xform_lo[i] > c.xform_hi[i]) return fail(FCP_ERR_INVALID_ARGUMENT, where + "empty id transform interval");
    }
    if (c.form == FCP_FORM_SEGMENT_REDUCE || c.form == FCP_FORM_GATHER_SCATTER) {
      if (c.seg_kind < FCP_SEG_IDS_I32 || c.seg_kind > FCP_SEG_CSR_I32)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad seg_kind");
      if (c.seg_input < 0 || c.seg_input >= d->n_host_inputs)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_input out of range");
      if (c.seg_stride < 1) return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_stride must be >= 1");
      if (d->host_input_elem_sizes[c.seg_input] != (c.seg_kind == FCP_SEG_IDS_I64 ? 8 : 4))
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "segment element size does not match seg_kind");
      if (c.rows_source == FCP_ROWS_FROM_IDS)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "pooled column needs an explicit row source");
    }
    if (c.form == FCP_FORM_SEGMENT_REDUCE && c.combiner != FCP_COMBINER_SUM &&
        c.combiner != FCP_COMBINER_MEAN)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "segment-reduce needs sum or mean");
    if (c.form == FCP_FORM_BATCH_COL_REDUCTION && d->host_input_ranks[c.ids_input] != 3)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "BatchColReduction input must be rank 3");
    if (c.rows_source < FCP_ROWS_FROM_IDS || c.rows_source > FCP_ROWS_FROM_INPUT_DIM0)
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad rows_source");
    if (c.rows_source == FCP_ROWS_FROM_SYMBOL && (c.rows_arg < 0 || c.rows_arg >= d->n_symbols))
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "symbol index out of range");
    if (c.rows_source == FCP_ROWS_FROM_INPUT_DIM0 &&
        (c.rows_arg < 0 || c.rows_arg >= d->n_host_inputs || d->host_input_ranks[c.rows_arg] < 1))
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "rows_arg host input out of range");
    for (int j = 0; j < k; ++j)
      if (d->columns[j].concat_group == c.concat_group && d->columns[j].concat_slot == c.concat_slot)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "duplicate concat slot");
  }
  return FCP_OK;
}

int validate_ext(const fcp_plan_desc_t *d, const fcp_column_ext_t *ext) {
  for (int k = 0; k < d->n_columns; ++k) {
    const fcp_column_ext_t &e = ext[k];
    if (e.seg_map_n == 0) continue;
    const fcp_column_desc_t &c = d->columns[k];
    const std::string where = "column " + std::to_string(k) + ": ";
    if (e.seg_map_n < 0 || e.seg_map_n > FCP_SEG_MAP_MAX) return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_map_n out of range");
    if (c.form != FCP_FORM_SEGMENT_REDUCE || (c.seg_kind != FCP_SEG_IDS_I32 && c.seg_kind != FCP_SEG_IDS_I64))
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "a segment-id map needs a pooled column with segment ids");
    if (c.seg_stride < e.seg_map_n) return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_stride is smaller than the number of mapped coordinates");
    if (e.seg_map_div < 1) return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_map_div must be >= 1");
    for (int i = 0; i < e.seg_map_n; ++i)
      if (e.seg_map_mul[i] < 0) return fail(FCP_ERR_INVALID_ARGUMENT, where + "negative seg_map_mul");
    if (e.seg_map_sym >= d->n_symbols || e.seg_map_sym < -1) return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_map_sym out of range");
    if (e.seg_map_sym >= 0 && !(e.seg_map_sym_slot == 4 || (e.seg_map_sym_slot >= 0 && e.seg_map_sym_slot < e.seg_map_n)))
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "seg_map_sym_slot out of range");
  }
  return FCP_OK;
}

// Run-time shapes -> per-column dynamic records, arena layout and launch
// geometry.  Mirrors what the generated host code evaluates per call from
// SymEngine expressions (cuda_emitter.cc:2151-2179, :2410-2455).
int finish_geometry(const fcp_plan *p, DynMeta *m);
void find_regular_csr(const fcp_plan *p, const FcpColDyn *dyn, DynMeta *m);

int compute_dyn_slow(const fcp_plan *p, const int32_t *offsets, const int32_t *shapes,
                     const int32_t *symbols, int64_t blob_bytes, FcpColDyn *dyn, DynMeta *m) {
  const int nc = (int)p->cols.size();
  const int ng = p->desc.n_groups;
  const int nh = (int)p->ranks.size();
  // element counts of all host inputs, one pass (this function is on the request
  // path whenever shapes change: no allocation, no string building unless it fails)
  thread_local std::vector<int64_t> numel_v, col_rows;
  numel_v.resize(nh);
  col_rows.resize(nc);
  for (int i = 0; i < nh; ++i) {
    int64_t n = 1;
    const int32_t *d = shapes + p->shape_off[i];
    for (int j = 0; j < p->ranks[i]; ++j) {
      if (d[j] < 0) return fail(FCP_ERR_SHAPE_MISMATCH, "negative dimension in concated_shapes");
      n *= d[j];
    }
    numel_v[i] = n;
    if (offsets[i] < 0) return fail(FCP_ERR_SHAPE_MISMATCH, "negative blob offset (int32 overflow?)");
    if (blob_bytes >= 0 && (int64_t)offsets[i] + n * p->elem_sizes[i] > blob_bytes)
      return fail(FCP_ERR_SHAPE_MISMATCH, "host input " + std::to_string(i) + " exceeds the blob");
  }
  const int64_t *numel = numel_v.data();
  m->group_rows.assign(ng, -1);
  for (int k = 0; k < nc; ++k) {
    const fcp_column_desc_t &c = p->cols[k].d;
    int64_t rows;
    if (c.form == FCP_FORM_EXTERNAL) continue; // rows of its group, below
    if (c.rows_source == FCP_ROWS_FROM_IDS) {
      rows = numel[c.ids_input];
    } else if (c.rows_source == FCP_ROWS_FROM_SYMBOL) {
      if (!symbols) return fail(FCP_ERR_INVALID_ARGUMENT, "plan needs the symbols tensor");
      rows = symbols[c.rows_arg];
    } else {
      rows = shapes[p->shape_off[c.rows_arg]];
    }
    if (rows < 0 || rows > 0x7fffffff) return fail(FCP_ERR_SHAPE_MISMATCH, "row count out of range");
    col_rows[k] = rows;
    int32_t &gr = m->group_rows[c.concat_group];
    if (gr >= 0 && gr != rows)
      return fail(FCP_ERR_SHAPE_MISMATCH, "columns of concat group " + std::to_string(c.concat_group) +
                                              " disagree on the row count");
    gr = (int32_t)rows;
  }
  for (int g = 0; g < ng; ++g)
    if (m->group_rows[g] < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "concat group without columns");
  for (int k = 0; k < nc; ++k)
    if (p->cols[k].d.form == FCP_FORM_EXTERNAL) col_rows[k] = m->group_rows[p->cols[k].d.concat_group];

  // arena: outputs, then CSR scratch (one malloc_buff, cuda_emitter.cc:2151-2163)
  int64_t cursor = 0;
  m->group_base.assign(ng, 0);
  if (p->desc.layout == FCP_LAYOUT_CONCAT) {
    for (int g = 0; g < ng; ++g) {
      m->group_base[g] = cursor;
      cursor += align128((int64_t)m->group_rows[g] * p->group_width[g] * 4);
    }
  }
  m->max_seg_nnz = 0;
  m->seg_pairs = 0;
  for (int k = 0; k < nc; ++k) {
    const HostColumn &hc = p->cols[k];
    const fcp_column_desc_t &c = hc.d;
    FcpColDyn &d = dyn[p->pos_of[k]];
    d.seg_off = 0;
    d.seg_sym = 1;
    const int64_t rows = col_rows[k];
    d.rows = (int32_t)rows;
    d.ids_off = c.form == FCP_FORM_EXTERNAL ? 0 : offsets[c.ids_input];
    if (d.ids_off % 4) return fail(FCP_ERR_UNSUPPORTED, "blob tensor not 4-byte aligned");
    const int64_t n_ids = c.form == FCP_FORM_EXTERNAL ? 0 : numel[c.ids_input];
    d.csr_base = -1;
    d.inner = 1;
    if (c.form == FCP_FORM_EXTERNAL) {
      d.nnz = 0;
    } else if (c.form == FCP_FORM_PASSTHROUGH) {
      if (n_ids != rows * c.dim) return fail(FCP_ERR_SHAPE_MISMATCH, "passthrough tensor size != rows*dim");
      if (n_ids / p->vec >= 0xFFFFFFFDLL) return fail(FCP_ERR_UNSUPPORTED, "passthrough tensor exceeds 2^32 slots");
      d.nnz = (int32_t)rows;
    } else if (c.form == FCP_FORM_BATCH_COL_REDUCTION) {
      const int32_t *s = shapes + p->shape_off[c.ids_input];
      if (s[0] != rows || s[2] != c.dim) return fail(FCP_ERR_SHAPE_MISMATCH, "BatchColReduction shape mismatch");
      d.inner = s[1];
      d.nnz = (int32_t)rows;
    } else {
      if (n_ids > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "more than 2^31 ids in one column");
      d.nnz = (int32_t)n_ids;
      if (c.form == FCP_FORM_GATHER && n_ids != rows)
        return fail(FCP_ERR_SHAPE_MISMATCH, "gather column: ids count != rows");
      if (c.form != FCP_FORM_GATHER) {
        d.seg_off = offsets[c.seg_input];
        if (d.seg_off % 4) return fail(FCP_ERR_UNSUPPORTED, "blob tensor not 4-byte aligned");
        const int64_t n_seg = numel[c.seg_input];
        if (c.seg_kind == FCP_SEG_CSR_I32) {
          if (n_seg != rows + 1) return fail(FCP_ERR_SHAPE_MISMATCH, "CSR offsets must have rows+1 entries");
        } else {
          if (n_seg < n_ids * c.seg_stride - (c.seg_stride - 1) && n_ids > 0)
            return fail(FCP_ERR_SHAPE_MISMATCH, "segment id tensor shorter than the id stream");
          if (hc.ext.seg_map_n > 0) {
            if (n_seg < n_ids * c.seg_stride) return fail(FCP_ERR_SHAPE_MISMATCH, "index matrix shorter than the id stream");
            if (hc.ext.seg_map_sym >= 0) {
              if (!symbols) return fail(FCP_ERR_INVALID_ARGUMENT, "plan needs the symbols tensor");
              d.seg_sym = symbols[hc.ext.seg_map_sym];
              if (d.seg_sym < (hc.ext.seg_map_sym_slot == 4 ? 1 : 0))
                return fail(FCP_ERR_SHAPE_MISMATCH, "segment-id map: symbol value out of range");
              // the pre-pass multiplies the factor by the symbol in 64 bits (load_seg_mapped): the product — and with it
              // every idx * factor for idx < 2^31 — must stay inside int64
              const int64_t factor = hc.ext.seg_map_sym_slot == 4 ? hc.ext.seg_map_div : hc.ext.seg_map_mul[hc.ext.seg_map_sym_slot];
              if (d.seg_sym > 0 && factor > (INT64_MAX >> 32) / d.seg_sym)
                return fail(FCP_ERR_UNSUPPORTED, "segment-id map: factor x symbol exceeds 2^31 (the reshaped row index would overflow)");
            }
          }
          if (d.nnz > m->max_seg_nnz) m->max_seg_nnz = d.nnz;
          m->seg_pairs += rows;
        }
      }
    }
    if (p->desc.layout == FCP_LAYOUT_CONCAT) {
      d.out_base = m->group_base[c.concat_group] + (int64_t)hc.out_off * 4;
      d.out_stride = p->group_width[c.concat_group];
    } else {
      d.out_base = cursor;
      d.out_stride = c.dim;
      cursor += align128(rows * c.dim * 4);
    }
  }
  // A few segment-id columns (the reference's models E / F: 10-20 multi-hot columns next to ~1000
  // one-hot ones): searching inside the blocks beats a second, dependent launch (E 15.6 -> 13.5 us).
  // Hundreds of them (RAGGED with SparseTensor indices): every row block would repeat the search on
  // the same arrays, and the one coalesced pre-pass scan wins (39.8 us vs 42.7-58.9 us).
  m->seg_search = p->seg_search && m->seg_pairs <= p->env.seg_search_max_pairs;
  m->csr_arena_off = cursor;
  int64_t csr_cursor = 0; // in int32 elements
  if (p->csr_by_pos) {
    const int64_t stride = ((int64_t)m->group_rows[0] + 1 + 31) / 32 * 32;
    for (int k : p->seg_cols) dyn[p->pos_of[k]].csr_base = (int32_t)(p->pos_of[k] * stride);
    csr_cursor = stride * (int64_t)p->cols.size();
    if (csr_cursor > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "CSR scratch exceeds 2^31 entries");
  } else {
    for (int k : p->seg_cols) {
      dyn[p->pos_of[k]].csr_base = (int32_t)csr_cursor;
      csr_cursor += (col_rows[k] + 1 + 31) / 32 * 32;
      if (csr_cursor > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "CSR scratch exceeds 2^31 entries");
    }
  }
  cursor += csr_cursor * 4;
  m->arena_bytes = cursor;
  find_regular_csr(p, dyn, m);
  return finish_geometry(p, m);
}

// Are this request's row-offset arrays regular (FcpLaunch::csr_reg)?  Mode 1: the plan lays its CSR scratch out by column
// position and the pre-pass fills it (not when the blocks search the segment ids themselves: then there is no scratch to
// read).  Mode 2: EVERY column of a one-group plan brings CSR offsets in the blob and the arrays lie one constant stride
// apart in position order — what a packer that keeps the converted inputs together produces; checked per descriptor
// install (n_columns compares), never assumed.
void find_regular_csr(const fcp_plan *p, const FcpColDyn *dyn, DynMeta *m) {
  m->csr_reg_mode = 0;
  m->csr_reg_stride = 0;
  m->csr_reg_base = 0;
  if (p->desc.n_groups != 1 || p->cols.empty() || m->group_rows[0] <= 0) return;
  const int nc = (int)p->cols.size();
  if (p->csr_by_pos) {
    if (p->seg_search && m->seg_pairs <= p->env.seg_search_max_pairs) return; // (m->seg_search is decided from the same two facts)
    m->csr_reg_mode = 1;
    m->csr_reg_stride = (int32_t)(((int64_t)m->group_rows[0] + 1 + 31) / 32 * 32);
    m->csr_reg_base = 0; // relative to the scratch (csr_arena_off)
    return;
  }
  int64_t stride = 0;
  for (int i = 0; i < nc; ++i) {
    const fcp_column_desc_t &c = p->cols[p->order[i]].d;
    if (c.seg_kind != FCP_SEG_CSR_I32 || (c.form != FCP_FORM_SEGMENT_REDUCE && c.form != FCP_FORM_GATHER_SCATTER)) return;
    if (i == 1) stride = dyn[1].seg_off - dyn[0].seg_off;
    if (i >= 1 && dyn[i].seg_off - dyn[i - 1].seg_off != stride) return;
  }
  if (nc == 1) stride = 4 * ((int64_t)m->group_rows[0] + 1);
  if (stride < 4 * ((int64_t)m->group_rows[0] + 1) || (stride & 3) || stride / 4 > 0x7fffffff || (dyn[0].seg_off & 3)) return;
  m->csr_reg_mode = 2;
  m->csr_reg_stride = (int32_t)(stride / 4);
  m->csr_reg_base = dyn[0].seg_off;
}

// launch geometry, one set per kernel kind (shared by both compute_dyn variants)
int finish_geometry(const fcp_plan *p, DynMeta *m) {
  const int ng = p->desc.n_groups;
  int32_t max_rows = 0;
  for (int g = 0; g < ng; ++g) max_rows = std::max(max_rows, m->group_rows[g]);
  for (int kind = 0; kind < 2; ++kind) {
    DynMeta::Geo &G2 = m->geo[kind];
    int rpw = 1;
    if (kind == 0) {
      // 4 rows per wave (16 per block) measured best on S2 at batch 512 and 2048: ~2 rounds of blocks
      // de-phase the read and write bursts; 8 rows lose to the tail.  Choosing fewer rows per wave for
      // narrow plans so that the grid reaches 8 blocks per CU (round 2: DLRM 896 -> 3584 blocks) made
      // them SLOWER (DLRM 5.0 -> 7.0 us, S2 at batch 128 11.0 -> 11.9 us): a block's fixed staging chain
      // costs more than the idle CUs, see profiles/HISTORY.md, round 2.
      while (rpw < 4 && max_rows >= 32 * rpw) rpw *= 2; // >= 64 rows -> 4, 32..63 -> 2, < 32 -> 1
      static const int forced = [] { // tuning aid: FCP_DIAG=rows_per_wave=1|2|4
        const int v = (int)fcp::diag_ll("rows_per_wave", 0);
        return (v == 1 || v == 2 || v == 4) ? v : 0;
      }();
      if (forced) rpw = forced;
    } // ragged kernel: one row per wave (2 interleaved rows measured slower: 33.7 vs 31.6 us)
    G2.rows_per_wave = rpw;
    int32_t blocks = 0;
    for (int g = 0; g < ng; ++g) {
      FcpGroupLaunch &G = G2.groups[g];
      G.rows = m->group_rows[g];
      G.nslots = p->group_nslots[g];
      G.nlist = p->list_n[kind][g];
      // a list that names every span in order is the identity: -1 spares the blocks a dependent load
      const int nspans_g = (p->group_nslots[g] + FCP_WAVE - 1) / FCP_WAVE;
      G.span_list_off = G.nlist == nspans_g ? -1 : p->list_off[kind][g];
      // listed spans are dealt to XCDs in groups of 8; fewer than 8 are not padded
      // (nsp8 = -nlist selects the plain mapping in the kernels)
      static const bool no_xcd_map = fcp::diag_on("no_xcd_map"); // tuning aid: plain span order
      G.nsp8 = (G.nlist >= 8 && !no_xcd_map) ? (G.nlist + 7) / 8 : -std::max(G.nlist, 1);
      G.block_begin = blocks;
      G.slot_map_off = p->group_map_off[g];
      G.csr_reg_stride = 0; // (fill_launch sets groups[0]'s per request)
      const int rows_per_block = FCP_WAVES_PER_BLOCK * rpw;
      const int64_t ntiles = ((int64_t)G.rows + rows_per_block - 1) / rows_per_block;
      const int64_t nb = G.nlist == 0 ? 0 : (G.nsp8 > 0 ? 8ll * G.nsp8 : (int64_t)G.nlist) * ntiles;
      if (blocks + nb > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "grid too large");
      blocks += (int32_t)nb;
    }
    G2.grid_blocks = blocks;
  }
  return FCP_OK;
}

// The same records for FCP_LAYOUT_CONCAT plans in ONE pass over compact per-column facts
// (fcp_plan::fast_cols, concat order = the order of `dyn`): this is the host's critical path when
// every request brings new shapes (1000 columns: ~13 us with the general routine).  Any
// irregularity returns -1 and the general routine runs instead, so every error message comes from
// one place.
int compute_dyn_fast(const fcp_plan *p, const int32_t *offsets, const int32_t *shapes, const int32_t *symbols,
                     int64_t blob_bytes, FcpColDyn *dyn, DynMeta *m) {
  const int nc = (int)p->fast_cols.size();
  const int ng = p->desc.n_groups;
  const int nh = (int)p->ranks.size();
  thread_local std::vector<int64_t> numel_v;
  numel_v.resize(nh);
  int64_t *numel = numel_v.data();
  for (int i = 0; i < nh; ++i) {
    int64_t n = 1;
    const int32_t *d = shapes + p->shape_off[i];
    const int rank = p->ranks[i];
    for (int j = 0; j < rank; ++j) {
      if (d[j] < 0) return -1;
      n *= d[j];
    }
    numel[i] = n;
    if (offsets[i] < 0 || (offsets[i] & 3)) return -1;
    if (blob_bytes >= 0 && (int64_t)offsets[i] + n * p->elem_sizes[i] > blob_bytes) return -1;
  }
  auto rows_of = [&](const fcp_plan::FastCol &c) -> int64_t {
    if (c.rows_source == FCP_ROWS_FROM_GROUP) return m->group_rows[c.group];
    if (c.rows_source == FCP_ROWS_FROM_IDS) return numel[c.ids_input];
    if (c.rows_source == FCP_ROWS_FROM_SYMBOL) return symbols ? (int64_t)symbols[c.rows_arg] : -1;
    return shapes[p->shape_off[c.rows_arg]];
  };
  m->group_rows.resize(ng);
  m->group_base.resize(ng);
  int64_t cursor = 0;
  for (int g = 0; g < ng; ++g) {
    const int64_t rows = rows_of(p->fast_cols[p->group_rep[g]]);
    if (rows < 0 || rows > 0x7fffffff) return -1;
    m->group_rows[g] = (int32_t)rows;
    m->group_base[g] = cursor;
    cursor += align128(rows * p->group_width[g] * 4);
  }
  int32_t max_seg_nnz = 0;
  int64_t seg_pairs = 0;
  int64_t gathered = 0; // floats
  for (int i = 0; i < nc; ++i) {
    const fcp_plan::FastCol &c = p->fast_cols[i];
    const int64_t rows = rows_of(c);
    if (rows != m->group_rows[c.group]) return -1;
    FcpColDyn &d = dyn[i];
    const bool external = c.form == FCP_FORM_EXTERNAL;
    const int64_t n_ids = external ? 0 : numel[c.ids_input];
    d.ids_off = external ? 0 : offsets[c.ids_input];
    d.seg_off = 0;
    d.out_base = m->group_base[c.group] + c.out_off_bytes;
    d.out_stride = p->group_width[c.group];
    d.csr_base = -1;
    d.inner = 1;
    d.rows = (int32_t)rows;
    d.seg_sym = 1;
    if (c.form == FCP_FORM_GATHER) {
      if (n_ids != rows) return -1;
      d.nnz = (int32_t)n_ids;
    } else if (external) {
      d.nnz = 0;
    } else if (c.form == FCP_FORM_PASSTHROUGH) {
      if (n_ids != rows * c.dim || n_ids / p->vec >= 0xFFFFFFFDLL) return -1;
      d.nnz = (int32_t)rows;
    } else if (c.form == FCP_FORM_BATCH_COL_REDUCTION) {
      const int32_t *sh = shapes + p->shape_off[c.ids_input];
      if (sh[0] != rows || sh[2] != c.dim) return -1;
      d.inner = sh[1];
      d.nnz = (int32_t)rows;
    } else {
      if (n_ids > 0x7fffffff) return -1;
      d.nnz = (int32_t)n_ids;
      d.seg_off = offsets[c.seg_input];
      const int64_t n_seg = numel[c.seg_input];
      if (c.seg_kind == FCP_SEG_CSR_I32) {
        if (n_seg != rows + 1) return -1;
      } else {
        if (n_seg < n_ids * c.seg_stride - (c.seg_stride - 1) && n_ids > 0) return -1;
        if (d.nnz > max_seg_nnz) max_seg_nnz = d.nnz;
        seg_pairs += rows;
      }
    }
    gathered += (int64_t)d.nnz * c.dim;
  }
  m->work_bytes = gathered * 4 + cursor;
  m->max_seg_nnz = max_seg_nnz;
  m->seg_pairs = seg_pairs;
  m->seg_search = p->seg_search && seg_pairs <= p->env.seg_search_max_pairs;
  m->csr_arena_off = cursor;
  int64_t csr_cursor = 0; // in int32 elements
  if (p->csr_by_pos) {
    const int64_t stride = ((int64_t)m->group_rows[0] + 1 + 31) / 32 * 32;
    for (int k : p->seg_cols) dyn[p->pos_of[k]].csr_base = (int32_t)(p->pos_of[k] * stride);
    csr_cursor = stride * nc;
    if (csr_cursor > 0x7fffffff) return -1;
  } else {
    for (int k : p->seg_cols) {
      FcpColDyn &d = dyn[p->pos_of[k]];
      d.csr_base = (int32_t)csr_cursor;
      csr_cursor += ((int64_t)d.rows + 1 + 31) / 32 * 32;
      if (csr_cursor > 0x7fffffff) return -1;
    }
  }
  m->arena_bytes = cursor + csr_cursor * 4;
  find_regular_csr(p, dyn, m);
  return finish_geometry(p, m);
}

int compute_dyn(const fcp_plan *p, const int32_t *offsets, const int32_t *shapes, const int32_t *symbols,
                int64_t blob_bytes, FcpColDyn *dyn, DynMeta *m) {
  const bool slow_only = fcp::diag_on("dyn_general"); // test aid (looked up per call: only on a descriptor miss)
  if (!slow_only && !p->fast_cols.empty()) {
    const int rc = compute_dyn_fast(p, offsets, shapes, symbols, blob_bytes, dyn, m);
    if (rc >= 0) return rc;
  }
  const int rc = compute_dyn_slow(p, offsets, shapes, symbols, blob_bytes, dyn, m);
  if (rc == FCP_OK) {
    int64_t gathered = 0;
    for (size_t i = 0; i < p->cols.size(); ++i) gathered += (int64_t)dyn[i].nnz * p->cols[p->order[i]].d.dim;
    m->work_bytes = gathered * 4 + m->csr_arena_off;
  }
  return rc;
}

// Are the boundaries evenly spaced closely enough that floor((x - b0) * inv) + 1 names the right bucket
// for x at every boundary and just below it (the places where rounding could push the guess over)?  Then
// the kernels take the guess and verify it with two reads instead of a 7-step binary search.  The guess is
// only a starting point — a failed verification falls back to the search — so this is a speed decision.
void uniform_boundaries(const std::vector<float> &b, float *b0, float *inv, float *step_out) {
  const int n = (int)b.size();
  if (n < 2 || !(b[n - 1] > b[0])) return;
  const float lo = b[0];
  const float scale = (float)((double)(n - 1) / ((double)b[n - 1] - (double)b[0]));
  if (!(scale > 0.0f) || !std::isfinite(scale)) return;
  auto guess = [&](float x) {
    float t = (x - lo) * scale;
    t = std::fmin(std::fmax(t, -1.0f), (float)n);
    int g = (int)std::floor(t) + 1;
    return std::min(std::max(g, 0), n);
  };
  auto exact = [&](float x) { return (int)(std::upper_bound(b.begin(), b.end(), x) - b.begin()); };
  int misses = 0;
  for (int i = 0; i < n; ++i) {
    if (i && !(b[i] > b[i - 1])) return; // not strictly increasing
    const float below = std::nextafter(b[i], -INFINITY);
    misses += guess(b[i]) != exact(b[i]);
    misses += guess(below) != exact(below);
  }
  if (misses * 8 > n) return; // an occasional miss costs one fallback search; many mean the spacing is not even
  *b0 = lo;
  *inv = scale;
  // Reproducible boundaries: b[i] == fma(i, step, b0) bit for bit (one correctly rounded operation on the host
  // and on the GPU alike).  Holds for the reference's 0, 5, ..., 495 and for any integer / dyadic grid.
  const float step = b[1] - b[0];
  if (!(step > 0.0f) || !std::isfinite(step)) return;
  for (int i = 0; i < n; ++i)
    if (std::fmaf((float)i, step, lo) != b[i]) return;
  *step_out = step;
}

void destroy_device(fcp_plan *p) {
  if (p->host_only) return;
  for (auto &s : p->slots) {
    if (s.h_dyn) (void)hipHostFree(s.h_dyn);
    if (s.d_dyn) (void)hipFree(s.d_dyn);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    for (auto &e : s.done_pool) (void)hipEventDestroy(e.second);
  }
  if (p->d_slot_map) (void)hipFree(p->d_slot_map);
  if (p->d_span_list) (void)hipFree(p->d_span_list);
  if (p->d_cols) (void)hipFree(p->d_cols);
  if (p->d_xforms) (void)hipFree(p->d_xforms);
  if (p->d_segmaps) (void)hipFree(p->d_segmaps);
  if (p->d_const) (void)hipFree(p->d_const);
  if (p->d_seg_cols) (void)hipFree(p->d_seg_cols);
  if (p->d_bad) (void)hipFree(p->d_bad);
  if (p->d_zeros) (void)hipFree(p->d_zeros);
  if (p->d_stamps) (void)hipFree(p->d_stamps);
}

int init_device(fcp_plan *p) {
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  const int nc = (int)p->cols.size();
  // slot map: slot -> position in the concat-ordered column arrays
  std::vector<uint32_t> map;
  for (int g = 0; g < p->desc.n_groups; ++g) {
    p->group_map_off[g] = (int32_t)map.size();
    for (int pos = 0; pos < nc; ++pos) {
      const HostColumn &hc = p->cols[p->order[pos]];
      if (hc.d.concat_group != g) continue;
      for (int s = 0; s < hc.d.dim / p->vec; ++s) map.push_back((uint32_t)pos);
    }
  }
  HIP_TRY(hipMalloc(&p->d_span_list, std::max<size_t>(p->span_list.size(), 1) * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(p->d_span_list, p->span_list.data(), p->span_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&p->d_slot_map, std::max<size_t>(map.size(), 1) * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(p->d_slot_map, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  // const buffers (bucketize boundaries), each 128-byte aligned as the reference's
  // identical boundary arrays (the usual case: hundreds of bucketized features with one
  // boundary list) are stored once, so that neighbouring columns can share one LDS copy
  int64_t const_bytes = 0;
  std::vector<int> owners; // indices of columns whose array was stored
  for (size_t k = 0; k < p->cols.size(); ++k) {
    HostColumn &hc = p->cols[k];
    if (hc.boundaries.empty()) continue;
    for (int o : owners)
      if (p->cols[o].boundaries == hc.boundaries) {
        hc.const_off = p->cols[o].const_off;
        break;
      }
    if (hc.const_off < 0) {
      hc.const_off = const_bytes;
      const_bytes += align128((int64_t)hc.boundaries.size() * 4);
      owners.push_back((int)k);
    }
  }
  for (HostColumn &hc : p->cols) { // id transform intervals beyond the first (the first one travels in the record)
    if (hc.xf_lo.size() <= 1) continue;
    hc.xf_const_off = const_bytes;
    const_bytes += align128((int64_t)(hc.xf_lo.size() - 1) * 16);
  }
  if (const_bytes) {
    HIP_TRY(hipMalloc(&p->d_const, const_bytes));
    for (const HostColumn &hc : p->cols) {
      if (hc.xf_const_off < 0) continue;
      std::vector<int64_t> pairs;
      for (size_t i = 1; i < hc.xf_lo.size(); ++i) {
        pairs.push_back(hc.xf_lo[i]);
        pairs.push_back(hc.xf_hi[i]);
      }
      HIP_TRY(hipMemcpy(p->d_const + hc.xf_const_off, pairs.data(), pairs.size() * 8, hipMemcpyHostToDevice));
    }
    for (int o : owners)
      HIP_TRY(hipMemcpy(p->d_const + p->cols[o].const_off, p->cols[o].boundaries.data(),
                        p->cols[o].boundaries.size() * 4, hipMemcpyHostToDevice));
  }
  // static column records (tables are bound on the first request)
  std::vector<FcpXform> h_xforms; // filled only if some column has an id transform
  p->h_cols.resize(nc);
  for (int pos = 0; pos < nc; ++pos) {
    const HostColumn &hc = p->cols[p->order[pos]];
    FcpColStatic &s = p->h_cols[pos];
    s.table = nullptr;
    s.boundaries = hc.const_off >= 0 ? reinterpret_cast<const float *>(p->d_const + hc.const_off) : nullptr;
    s.vocab = hc.d.vocab;
    s.dim = hc.d.dim;
    s.out_off = hc.out_off;
    s.flags = FCP_F_PACK(hc.d.form, hc.d.combiner, hc.d.id_source, hc.d.seg_kind);
    s.n_boundaries = (int32_t)hc.boundaries.size();
    s.seg_stride = hc.d.seg_stride < 1 ? 1 : hc.d.seg_stride;
    s.bnd_b0 = 0.0f;
    s.bnd_inv = 0.0f;
    s.bnd_step = 0.0f;
    uniform_boundaries(hc.boundaries, &s.bnd_b0, &s.bnd_inv, &s.bnd_step);
    // id transform: an empty interval set (nothing is "in") is encoded as one impossible interval
    s.bnd_off = -1;
    s.xform = 0;
    if (hc.d.xform_mode != FCP_XFORM_NONE || hc.d.hash_buckets > 0) {
      if (h_xforms.empty()) {
        FcpXform none;
        none.lo0 = 1;
        none.hi0 = 0;
        none.sub = 0;
        none.extra = nullptr;
        none.hash_buckets = 0;
        none.pad_ = 0;
        h_xforms.assign(nc, none);
      }
      FcpXform &x = h_xforms[pos];
      if (hc.d.hash_buckets > 0) {
        s.xform |= FCP_XFORM_HASH_BIT;
        x.hash_buckets = hc.d.hash_buckets;
      }
      if (hc.d.xform_mode != FCP_XFORM_NONE) {
        const uint32_t n = (uint32_t)std::max<size_t>(hc.xf_lo.size(), 1);
        s.xform |= (n << 2) | (uint32_t)hc.d.xform_mode;
        x.sub = hc.d.xform_substitute;
        if (!hc.xf_lo.empty()) {
          x.lo0 = hc.xf_lo[0];
          x.hi0 = hc.xf_hi[0];
        }
        if (hc.xf_const_off >= 0) x.extra = reinterpret_cast<const int64_t *>(p->d_const + hc.xf_const_off);
      }
    }
  }
  HIP_TRY(hipMalloc(&p->d_cols, nc * sizeof(FcpColStatic)));
  HIP_TRY(hipMemcpy(p->d_cols, p->h_cols.data(), nc * sizeof(FcpColStatic), hipMemcpyHostToDevice));
  if (!h_xforms.empty()) {
    HIP_TRY(hipMalloc(&p->d_xforms, nc * sizeof(FcpXform)));
    HIP_TRY(hipMemcpy(p->d_xforms, h_xforms.data(), nc * sizeof(FcpXform), hipMemcpyHostToDevice));
  }
  if (p->has_seg_map) {
    std::vector<FcpSegMap> h_maps(nc);
    for (int pos = 0; pos < nc; ++pos) {
      const fcp_column_ext_t &e = p->cols[p->order[pos]].ext;
      FcpSegMap &sm = h_maps[pos];
      for (int i = 0; i < 4; ++i) sm.mul[i] = i < e.seg_map_n ? e.seg_map_mul[i] : 0;
      sm.div = e.seg_map_n > 0 ? e.seg_map_div : 1;
      sm.n = e.seg_map_n;
      sm.sym_slot = e.seg_map_n > 0 && e.seg_map_sym >= 0 ? e.seg_map_sym_slot : -1;
    }
    HIP_TRY(hipMalloc(&p->d_segmaps, nc * sizeof(FcpSegMap)));
    HIP_TRY(hipMemcpy(p->d_segmaps, h_maps.data(), nc * sizeof(FcpSegMap), hipMemcpyHostToDevice));
  }
  if (!p->seg_cols.empty()) {
    std::vector<int32_t> seg_pos;
    for (int k : p->seg_cols) seg_pos.push_back(p->pos_of[k]);
    HIP_TRY(hipMalloc(&p->d_seg_cols, seg_pos.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p->d_seg_cols, seg_pos.data(), seg_pos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(&p->d_zeros, 256));
  HIP_TRY(hipMemset(p->d_zeros, 0, 256));
  if (p->desc.flags & FCP_FLAG_COUNT_BAD_IDS) {
    HIP_TRY(hipMalloc(&p->d_bad, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(p->d_bad, 0, sizeof(unsigned long long)));
  }
  {
    int large_bar = 0;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, p->desc.device);
    p->host_writes_dyn = large_bar != 0 && !p->env.dyn_upload_kernel; // (FCP_DYN_UPLOAD=kernel)
  }
#if defined(FCP_STAMPS)
  HIP_TRY(hipMalloc(&p->d_stamps, 8 * sizeof(unsigned long long) * 65536));
  HIP_TRY(hipMemset(p->d_stamps, 0, 8 * sizeof(unsigned long long) * 65536));
#endif
  for (auto &s : p->slots) {
    // rounded up to 16 bytes: the upload kernel moves uint4s
    const size_t dyn_bytes = (nc * sizeof(FcpColDyn) + 15) / 16 * 16;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.h_dyn), dyn_bytes, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(&s.h_dyn_dev, s.h_dyn, 0));
    if (p->host_writes_dyn) {
      if (hipExtMallocWithFlags(reinterpret_cast<void **>(&s.d_dyn), dyn_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        p->host_writes_dyn = false; // fall back for every slot: nothing has been used yet
        for (auto &t : p->slots)
          if (t.d_dyn) {
            (void)hipFree(t.d_dyn);
            t.d_dyn = nullptr;
          }
      }
    }
  }
  for (auto &s : p->slots) {
    const size_t dyn_bytes = (nc * sizeof(FcpColDyn) + 15) / 16 * 16;
    if (!s.d_dyn) HIP_TRY(hipMalloc(&s.d_dyn, dyn_bytes));
    HIP_TRY(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming));
    s.done = nullptr; // created per stream on first use (done_event_for)
  }
  p->bound_tables.assign(p->desc.n_device_inputs, nullptr);
  {
    // the gate at plan level (the reference's check_table_size, cuda_emitter.cc:1080-1094, decides per table
    // against 256 MiB): the tables this plan reads on this device must fit the device at all
    int64_t shard_bytes = 0;
    (void)fcp_plan_table_bytes(p, &shard_bytes, nullptr);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0 && (uint64_t)shard_bytes > total_b)
      return fail(FCP_ERR_UNSUPPORTED, "this plan's tables need " + std::to_string(shard_bytes) + " bytes on a device with " +
                                           std::to_string(total_b) + ": shard them over more GPUs (fcp_placement_decide)");
  }
  return FCP_OK;
}

} // namespace fcph

// =============================== C ABI ======================================
extern "C" {

int fcp_abi_version(void) { return FCP_ABI_VERSION; }

const char *fcp_status_string(int status) {
  switch (status) {
  case FCP_OK: return "ok";
  case FCP_ERR_INVALID_ARGUMENT: return "invalid argument";
  case FCP_ERR_SHAPE_MISMATCH: return "run-time shapes do not match the plan";
  case FCP_ERR_ALLOC: return "allocator callback failed";
  case FCP_ERR_HIP: return "HIP runtime error";
  case FCP_ERR_UNSUPPORTED: return "unsupported";
  case FCP_ERR_NO_DEVICE: return "no usable gfx950 device";
  default: return "unknown status";
  }
}

const char *fcp_last_error(void) { return g_last_error.c_str(); }

// ---- plan ---------------------------------------------------------------------
int fcp_plan_create(const fcp_plan_desc_t *desc, fcp_plan_t **out) { return fcp_plan_create_ex(desc, nullptr, out); }

int fcp_plan_create_ex(const fcp_plan_desc_t *desc, const fcp_column_ext_t *ext, fcp_plan_t **out) {
  if (!out) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan out pointer");
  *out = nullptr;
  int rc = validate_desc(desc);
  if (rc) return rc;
  if (ext && (rc = validate_ext(desc, ext))) return rc;
  fcp_plan *p = new (std::nothrow) fcp_plan();
  if (!p) return fail(FCP_ERR_ALLOC, "out of host memory");
  p->env = fcp::read_env(); // the library's shipping switches, read here and nowhere on the request path (fcp_env.h)
  p->desc = *desc;
  p->desc.columns = nullptr;
  p->desc.host_input_ranks = nullptr;
  p->desc.host_input_elem_sizes = nullptr;
  p->host_only = (desc->flags & kFlagHostOnly) != 0;
  p->ranks.assign(desc->host_input_ranks, desc->host_input_ranks + desc->n_host_inputs);
  p->elem_sizes.assign(desc->host_input_elem_sizes, desc->host_input_elem_sizes + desc->n_host_inputs);
  p->shape_off.resize(desc->n_host_inputs);
  int32_t acc = 0;
  for (int i = 0; i < desc->n_host_inputs; ++i) {
    p->shape_off[i] = acc;
    acc += p->ranks[i];
  }
  p->rank_sum = acc;
  p->cols.resize(desc->n_columns);
  int gcd4 = 4;
  for (int k = 0; k < desc->n_columns; ++k) {
    HostColumn &hc = p->cols[k];
    hc.d = desc->columns[k];
    if (hc.d.id_source == FCP_IDS_F32_BUCKETIZE && hc.d.boundaries && hc.d.n_boundaries > 0 &&
        hc.d.form != FCP_FORM_PASSTHROUGH && hc.d.form != FCP_FORM_BATCH_COL_REDUCTION && hc.d.form != FCP_FORM_EXTERNAL)
      hc.boundaries.assign(hc.d.boundaries, hc.d.boundaries + hc.d.n_boundaries);
    hc.d.boundaries = nullptr;
    if (hc.d.xform_mode != FCP_XFORM_NONE && hc.d.xform_n > 0) {
      hc.xf_lo.assign(hc.d.xform_lo, hc.d.xform_lo + hc.d.xform_n);
      hc.xf_hi.assign(hc.d.xform_hi, hc.d.xform_hi + hc.d.xform_n);
    }
    hc.d.xform_lo = hc.d.xform_hi = nullptr;
    if (ext && ext[k].seg_map_n > 0) {
      hc.ext = ext[k];
      p->has_seg_map = true;
    }
    if (hc.d.dim % 4) gcd4 = (hc.d.dim % 2) ? 1 : std::min(gcd4, 2);
    const int f = hc.d.form;
    if ((f == FCP_FORM_SEGMENT_REDUCE || f == FCP_FORM_GATHER_SCATTER) && hc.d.seg_kind != FCP_SEG_CSR_I32)
      p->seg_cols.push_back(k);
  }
  p->vec = gcd4;
  // any-order ScatterNd columns last: their part of the CSR scratch (the inverse maps, built with atomic max from zero)
  // is then ONE range at the tail, the only one a request has to clear
  std::stable_partition(p->seg_cols.begin(), p->seg_cols.end(), [&](int32_t k) { return p->cols[k].d.form != FCP_FORM_GATHER_SCATTER; });
  p->n_seg_plain = 0;
  for (int32_t k : p->seg_cols) p->n_seg_plain += p->cols[k].d.form != FCP_FORM_GATHER_SCATTER;
  // Segment-id columns (SparseTensor indices / row ids): unsharded plans let every block find its rows'
  // ranges with a 16-ary search (fcp_kernels.hip::seg_lower_bound) instead of running the
  // ComputeSegmentOffsets pre-pass as a second, dependent launch.  Row-sharded plans keep the
  // pre-pass: fcp_shard_finalize needs the row lengths as CSR.  FCP_SEG_PREPASS=1: tuning aid.
  p->seg_search = desc->shard_world <= 1 && !p->env.seg_prepass;
  if (p->has_seg_map) p->seg_search = false; // mapped segment ids are evaluated by the pre-pass only
  for (const HostColumn &hc : p->cols) {
    if (hc.d.seg_kind != FCP_SEG_NONE && hc.d.seg_stride > 0xffff) p->seg_search = false; // stride rides in 16 flag bits
    // ScatterNd columns take their row ids in any order (cuda_emitter.cc:296-345): nothing to search, the pre-pass
    // builds the row -> position map
    if (hc.d.form == FCP_FORM_GATHER_SCATTER && hc.d.seg_kind != FCP_SEG_CSR_I32) {
      p->has_inverse = true;
      p->seg_search = false;
    }
  }
  p->csr_by_pos = desc->layout == FCP_LAYOUT_CONCAT && desc->n_groups == 1 && !p->has_inverse && !p->seg_cols.empty() &&
                  2 * p->seg_cols.size() >= p->cols.size();
  // (a column that brings its row offsets in the blob reads them there: the kernel's regular-CSR shortcut serves every pooled /
  // scatter column of the launch from ONE matrix, so a plan that mixes the two encodings keeps the packed scratch)
  for (const HostColumn &hc : p->cols)
    if ((hc.d.form == FCP_FORM_SEGMENT_REDUCE || hc.d.form == FCP_FORM_GATHER_SCATTER) && hc.d.seg_kind == FCP_SEG_CSR_I32) p->csr_by_pos = false;
  if (fcp::diag_ll("csr_by_pos", 1) == 0) p->csr_by_pos = false; // tuning aid: FCP_DIAG=csr_by_pos=0 = packed scratch (the round-5 layout)
  // The kernels park a table row as one 32-bit number (the three largest values are their sentinels): a table — or
  // one shard of it — may hold up to 2^32 - 3 ROWS, of any width: the byte offset is formed in 64 bits where the row
  // is read.  The dense body keeps round 2's pre-scaled 32-bit slot offsets (row * dim / vec) while every table of
  // the plan stays below 2^32 - 3 slots (64 GB at vec 4), which saves it a 64-bit multiply per read.  (The
  // reference's int arithmetic stops at 2^31 elements = 8 GB, cuda_emitter.cc:270-271.)
  const bool wide_rows_forced = fcp::diag_on("wide_rows"); // test aid: the 64-bit row arithmetic on small tables
  for (int k = 0; k < desc->n_columns; ++k) {
    const fcp_column_desc_t &c = p->cols[k].d;
    if (c.form == FCP_FORM_PASSTHROUGH || c.form == FCP_FORM_BATCH_COL_REDUCTION || c.form == FCP_FORM_EXTERNAL) continue;
    const int64_t local_vocab = (c.vocab - desc->shard_rank + desc->shard_world - 1) / desc->shard_world;
    if (local_vocab >= 0xFFFFFFFDLL) {
      delete p;
      return fail(FCP_ERR_UNSUPPORTED, "column " + std::to_string(k) + ": table shard exceeds 2^32 - 3 rows");
    }
    if (local_vocab * (c.dim / p->vec) >= 0xFFFFFFFDLL || wide_rows_forced) p->wide_rows = true;
  }
  // concat layout: offsets = prefix sums of dims in slot order
  // (concat_outputs_op_gpu.cu.cc:74-79)
  const int ng = desc->n_groups;
  p->group_width.assign(ng, 0);
  p->group_nslots.assign(ng, 0);
  p->group_map_off.assign(ng, 0);
  for (int g = 0; g < ng; ++g) {
    std::vector<int> members;
    for (int k = 0; k < desc->n_columns; ++k)
      if (p->cols[k].d.concat_group == g) members.push_back(k);
    if (members.empty()) {
      delete p;
      return fail(FCP_ERR_INVALID_ARGUMENT, "concat group without columns");
    }
    std::sort(members.begin(), members.end(),
              [&](int a, int b) { return p->cols[a].d.concat_slot < p->cols[b].d.concat_slot; });
    int32_t off = 0;
    for (int k : members) {
      p->cols[k].out_off = off;
      off += p->cols[k].d.dim;
    }
    p->group_width[g] = off;
    p->group_nslots[g] = off / p->vec;
    for (int k : members) p->order.push_back(k);
  }
  p->pos_of.assign(desc->n_columns, 0);
  for (int pos = 0; pos < desc->n_columns; ++pos) p->pos_of[p->order[pos]] = pos;
  if (desc->layout == FCP_LAYOUT_CONCAT && desc->n_groups <= 255 && !p->has_seg_map) { // (maps: the general routine resolves their symbol)
    p->fast_cols.resize(desc->n_columns);
    p->group_rep.assign(desc->n_groups, -1);
    for (int pos = 0; pos < desc->n_columns; ++pos) {
      const HostColumn &hc = p->cols[p->order[pos]];
      fcp_plan::FastCol &f = p->fast_cols[pos];
      f.ids_input = hc.d.ids_input;
      f.seg_input = hc.d.seg_input;
      f.rows_arg = hc.d.rows_arg;
      f.dim = hc.d.dim;
      f.seg_stride = hc.d.seg_stride < 1 ? 1 : hc.d.seg_stride;
      f.out_off_bytes = (int64_t)hc.out_off * 4;
      f.form = (uint8_t)hc.d.form;
      f.rows_source = (uint8_t)hc.d.rows_source;
      f.seg_kind = (uint8_t)hc.d.seg_kind;
      f.group = (uint8_t)hc.d.concat_group;
      if (p->group_rep[hc.d.concat_group] < 0 && hc.d.form != FCP_FORM_EXTERNAL) p->group_rep[hc.d.concat_group] = pos;
    }
    for (int g = 0; g < desc->n_groups; ++g)
      if (p->group_rep[g] < 0) p->fast_cols.clear(); // a group without columns: let the general routine report it
  }
  // hybrid dispatch: classify every 64-slot span of every group
  for (int kind = 0; kind < 2; ++kind) {
    p->list_off[kind].assign(ng, -1);
    p->list_n[kind].assign(ng, 0);
  }
  p->dense_only = true;
  for (int g = 0; g < ng; ++g) {
    const int nspans = (p->group_nslots[g] + FCP_WAVE - 1) / FCP_WAVE;
    std::vector<char> ragged(nspans, 0);
    for (int k = 0; k < desc->n_columns; ++k) {
      const HostColumn &hc = p->cols[k];
      if (hc.d.concat_group != g) continue;
      if (hc.d.form == FCP_FORM_GATHER || hc.d.form == FCP_FORM_PASSTHROUGH || hc.d.form == FCP_FORM_EXTERNAL) continue;
      const int s0 = hc.out_off / p->vec / FCP_WAVE, s1 = (hc.out_off + hc.d.dim - 1) / p->vec / FCP_WAVE;
      for (int sp = s0; sp <= s1 && sp < nspans; ++sp) ragged[sp] = 1;
    }
    for (int kind = 0; kind < 2; ++kind) {
      p->list_off[kind][g] = (int32_t)p->span_list.size();
      for (int sp = 0; sp < nspans; ++sp)
        if (ragged[sp] == kind) p->span_list.push_back((uint32_t)sp);
      p->list_n[kind][g] = (int32_t)p->span_list.size() - p->list_off[kind][g];
    }
    if (p->list_n[1][g] > 0) p->dense_only = false;
  }
  if (!p->host_only) {
    rc = init_device(p);
    if (rc) {
      destroy_device(p);
      delete p;
      return rc;
    }
  } else {
    int32_t off = 0;
    for (int g = 0; g < ng; ++g) {
      p->group_map_off[g] = off;
      off += p->group_nslots[g];
    }
  }
  *out = p;
  return FCP_OK;
}

namespace {
// A column-plan file in memory (see include/fcp_hip.h for the format).
struct ParsedPlanFile {
  fcp_plan_desc_t d;
  std::vector<int32_t> ranks, esz;
  std::vector<fcp_column_desc_t> cols;
  std::vector<std::vector<float>> bnd;
  std::vector<std::vector<int64_t>> xlo, xhi;
  std::vector<fcp_column_ext_t> ext; // "segmaps" section (version 4); empty = no column has extensions
  // "stage" section (version 3): what Addons>ConcatInputs does to each of ITS inputs while packing
  std::vector<uint8_t> stage_modes;
  std::vector<int32_t> stage_rows_symbol;
  int32_t stage_symbols_input = -1;
  bool has_stage = false;
};

int parse_plan_file(const char *path, ParsedPlanFile &P) {
  std::FILE *f = std::fopen(path, "r");
  if (!f) return fail(FCP_ERR_INVALID_ARGUMENT, std::string("cannot open column plan ") + path);
  struct Closer {
    std::FILE *f;
    ~Closer() { std::fclose(f); }
  } closer{f};
  const std::string where = std::string("column plan ") + path + ": ";
  char tag[32], t2[32], t3[32];
  int version = 0, n_host = 0, n_cols = 0;
  fcp_plan_desc_t &d = P.d;
  std::memset(&d, 0, sizeof(d));
  if (std::fscanf(f, "%31s %d", tag, &version) != 2 || std::strcmp(tag, "fcp_plan") || version < 1 || version > 4)
    return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad header");
  if (std::fscanf(f, "%31s %d", tag, &d.layout) != 2 || std::strcmp(tag, "layout"))
    return fail(FCP_ERR_INVALID_ARGUMENT, where + "expected 'layout'");
  if (std::fscanf(f, "%31s %d %31s %d %31s %d", tag, &d.n_groups, t2, &d.n_symbols, t3, &d.n_device_inputs) != 6 ||
      std::strcmp(tag, "groups") || std::strcmp(t2, "symbols") || std::strcmp(t3, "device_inputs"))
    return fail(FCP_ERR_INVALID_ARGUMENT, where + "expected 'groups G symbols S device_inputs D'");
  if (std::fscanf(f, "%31s %d", tag, &n_host) != 2 || std::strcmp(tag, "host_inputs") || n_host < 0 || n_host > (1 << 24))
    return fail(FCP_ERR_INVALID_ARGUMENT, where + "expected 'host_inputs N'");
  P.ranks.resize(n_host);
  P.esz.resize(n_host);
  for (int i = 0; i < n_host; ++i)
    if (std::fscanf(f, "%d %d", &P.ranks[i], &P.esz[i]) != 2) return fail(FCP_ERR_INVALID_ARGUMENT, where + "truncated host input list");
  if (std::fscanf(f, "%31s %d", tag, &n_cols) != 2 || std::strcmp(tag, "columns") || n_cols < 0 || n_cols > (1 << 24))
    return fail(FCP_ERR_INVALID_ARGUMENT, where + "expected 'columns C'");
  P.cols.resize(n_cols);
  P.bnd.resize(n_cols);
  P.xlo.resize(n_cols);
  P.xhi.resize(n_cols);
  for (int k = 0; k < n_cols; ++k) {
    fcp_column_desc_t &c = P.cols[k];
    std::memset(&c, 0, sizeof(c));
    long long vocab = 0;
    if (std::fscanf(f, "%d %d %d %d %lld %d %d %d %d %d %d %d %d %d %d", &c.form, &c.combiner, &c.dim, &c.id_source, &vocab,
                    &c.table_input, &c.ids_input, &c.seg_input, &c.seg_kind, &c.seg_stride, &c.rows_source, &c.rows_arg,
                    &c.concat_group, &c.concat_slot, &c.n_boundaries) != 15 ||
        c.n_boundaries < 0 || c.n_boundaries > (1 << 24))
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "truncated or malformed column " + std::to_string(k));
    c.vocab = vocab;
    P.bnd[k].resize(c.n_boundaries);
    for (int b = 0; b < c.n_boundaries; ++b)
      if (std::fscanf(f, "%f", &P.bnd[k][b]) != 1) return fail(FCP_ERR_INVALID_ARGUMENT, where + "truncated boundary list");
    c.boundaries = c.n_boundaries ? P.bnd[k].data() : nullptr;
    if (version >= 2) { // id transform: mode, number of intervals, substitute, (lo, hi) pairs
      long long sub = 0, hb = 0;
      if (std::fscanf(f, "%d %d %lld %lld", &c.xform_mode, &c.xform_n, &sub, &hb) != 4 || c.xform_n < 0 || c.xform_n > (1 << 20))
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "truncated or malformed id transform of column " + std::to_string(k));
      c.xform_substitute = sub;
      c.hash_buckets = hb;
      P.xlo[k].resize(c.xform_n);
      P.xhi[k].resize(c.xform_n);
      for (int i = 0; i < c.xform_n; ++i) {
        long long lo = 0, hi = 0;
        if (std::fscanf(f, "%lld %lld", &lo, &hi) != 2) return fail(FCP_ERR_INVALID_ARGUMENT, where + "truncated interval list");
        P.xlo[k][i] = lo;
        P.xhi[k][i] = hi;
      }
      c.xform_lo = c.xform_n ? P.xlo[k].data() : nullptr;
      c.xform_hi = c.xform_n ? P.xhi[k].data() : nullptr;
    }
  }
  // optional trailing sections: "segmaps M" + M x "column n sym slot mul0 mul1 mul2 mul3 div" (version 4), then
  // "stage N symbols_input K" + N x "mode rows_symbol" (version 3)
  bool seen_maps = false;
  for (;;) {
    int count = 0;
    const int got = std::fscanf(f, "%31s %d", tag, &count);
    if (got == EOF || got == 0) break;
    if (got != 2) return fail(FCP_ERR_INVALID_ARGUMENT, where + "malformed trailing section");
    if (version >= 4 && !std::strcmp(tag, "segmaps") && !seen_maps && !P.has_stage) {
      if (count < 0 || count > n_cols) return fail(FCP_ERR_INVALID_ARGUMENT, where + "bad 'segmaps M'");
      seen_maps = true;
      P.ext.assign(n_cols, fcp_column_ext_t{});
      for (int i = 0; i < count; ++i) {
        int col = -1, n = 0, sym = -1, slot = 0;
        long long mul[4] = {0, 0, 0, 0}, div = 1;
        if (std::fscanf(f, "%d %d %d %d %lld %lld %lld %lld %lld", &col, &n, &sym, &slot, &mul[0], &mul[1], &mul[2], &mul[3], &div) != 9 ||
            col < 0 || col >= n_cols || n < 1 || n > FCP_SEG_MAP_MAX || P.ext[col].seg_map_n != 0)
          return fail(FCP_ERR_INVALID_ARGUMENT, where + "malformed segmaps entry " + std::to_string(i));
        fcp_column_ext_t &e = P.ext[col];
        e.seg_map_n = n;
        e.seg_map_sym = sym;
        e.seg_map_sym_slot = slot;
        for (int j = 0; j < 4; ++j) e.seg_map_mul[j] = mul[j];
        e.seg_map_div = div;
      }
    } else if (version >= 3 && !std::strcmp(tag, "stage") && !P.has_stage) {
      const int n_stage = count;
      int sym_in = -1;
      if (std::fscanf(f, "%31s %d", t2, &sym_in) != 2 || std::strcmp(t2, "symbols_input") || n_stage < 0 || n_stage > (1 << 24) ||
          sym_in < -1 || sym_in >= n_stage)
        return fail(FCP_ERR_INVALID_ARGUMENT, where + "expected 'stage N symbols_input K'");
      P.stage_modes.resize(n_stage);
      P.stage_rows_symbol.resize(n_stage);
      for (int i = 0; i < n_stage; ++i) {
        int mode = 0, sym = -1;
        if (std::fscanf(f, "%d %d", &mode, &sym) != 2 || mode < FCP_STAGE_COPY || mode > FCP_STAGE_SEG_TO_CSR || sym < -1 ||
            sym >= d.n_symbols || (mode == FCP_STAGE_SEG_TO_CSR && (sym < 0 || sym_in < 0)))
          return fail(FCP_ERR_INVALID_ARGUMENT, where + "malformed stage entry " + std::to_string(i));
        P.stage_modes[i] = (uint8_t)mode;
        P.stage_rows_symbol[i] = sym;
      }
      P.stage_symbols_input = sym_in;
      P.has_stage = true;
    } else {
      return fail(FCP_ERR_INVALID_ARGUMENT, where + "unexpected section '" + tag + "'");
    }
  }
  d.abi_version = FCP_ABI_VERSION;
  d.n_columns = n_cols;
  d.columns = P.cols.data();
  d.n_host_inputs = n_host;
  d.host_input_ranks = P.ranks.data();
  d.host_input_elem_sizes = P.esz.data();
  d.shard_rank = 0;
  d.shard_world = 1;
  return FCP_OK;
}
} // namespace

int fcp_plan_create_from_file(const char *path, int32_t device, uint32_t flags, fcp_plan_t **out) {
  if (!path || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  ParsedPlanFile P;
  const int rc = parse_plan_file(path, P);
  if (rc) return rc;
  if (P.has_stage && (int32_t)P.stage_modes.size() != P.d.n_host_inputs)
    return fail(FCP_ERR_INVALID_ARGUMENT, std::string("column plan ") + path + ": the stage section lists " +
                                              std::to_string(P.stage_modes.size()) + " inputs, the plan has " +
                                              std::to_string(P.d.n_host_inputs) + " host inputs");
  P.d.device = device;
  P.d.flags = flags;
  return fcp_plan_create_ex(&P.d, P.ext.empty() ? nullptr : P.ext.data(), out);
}

int fcp_plan_file_stage_info(const char *path, int32_t *n_inputs, uint8_t *modes, int32_t *rows_symbol, int32_t capacity,
                             int32_t *symbols_input) {
  if (!path) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  ParsedPlanFile P;
  const int rc = parse_plan_file(path, P);
  if (rc) return rc;
  const int32_t n = P.has_stage ? (int32_t)P.stage_modes.size() : 0;
  if (n_inputs) *n_inputs = n;
  if (symbols_input) *symbols_input = P.has_stage ? P.stage_symbols_input : -1;
  for (int32_t i = 0; i < n && i < capacity; ++i) {
    if (modes) modes[i] = P.stage_modes[i];
    if (rows_symbol) rows_symbol[i] = P.stage_rows_symbol[i];
  }
  return FCP_OK;
}

int fcp_plan_counts(const fcp_plan_t *p, int32_t *n_columns, int32_t *n_groups, int32_t *n_host_inputs,
                    int32_t *n_device_inputs, int32_t *n_symbols) {
  if (!p) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan");
  if (n_columns) *n_columns = p->desc.n_columns;
  if (n_groups) *n_groups = p->desc.n_groups;
  if (n_host_inputs) *n_host_inputs = p->desc.n_host_inputs;
  if (n_device_inputs) *n_device_inputs = p->desc.n_device_inputs;
  if (n_symbols) *n_symbols = p->desc.n_symbols;
  return FCP_OK;
}

int fcp_plan_output_columns(const fcp_plan_t *p, int32_t *n, int32_t *indices, int32_t capacity) {
  if (!p) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan");
  int32_t count = 0;
  for (int32_t k = 0; k < (int32_t)p->cols.size(); ++k) {
    if (p->cols[k].d.form == FCP_FORM_EXTERNAL) continue;
    if (indices && count < capacity) indices[count] = k;
    ++count;
  }
  if (n) *n = count;
  return FCP_OK;
}

int fcp_plan_table_bytes(const fcp_plan_t *p, int64_t *shard_bytes, int64_t *max_table_bytes_unsharded) {
  if (!p) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan");
  // a table input may feed several columns (shared embeddings): count each once
  std::vector<int64_t> local(p->desc.n_device_inputs, 0), whole(p->desc.n_device_inputs, 0);
  for (const HostColumn &hc : p->cols) {
    const int f = hc.d.form;
    if (f != FCP_FORM_GATHER && f != FCP_FORM_SEGMENT_REDUCE && f != FCP_FORM_GATHER_SCATTER) continue;
    const int64_t local_vocab =
        (hc.d.vocab - p->desc.shard_rank + p->desc.shard_world - 1) / p->desc.shard_world;
    local[hc.d.table_input] = std::max(local[hc.d.table_input], local_vocab * hc.d.dim * 4);
    whole[hc.d.table_input] = std::max(whole[hc.d.table_input], hc.d.vocab * hc.d.dim * 4);
  }
  int64_t sum = 0, mx = 0;
  for (int t = 0; t < p->desc.n_device_inputs; ++t) {
    sum += local[t];
    mx = std::max(mx, whole[t]);
  }
  if (shard_bytes) *shard_bytes = sum;
  if (max_table_bytes_unsharded) *max_table_bytes_unsharded = mx;
  return FCP_OK;
}

int fcp_placement_assign(const int64_t *table_bytes, int32_t n_tables, int64_t hbm_bytes, int64_t reserve_bytes, int32_t world,
                         int32_t prefer_mode, int32_t *owner, fcp_placement_t *out) {
  if (!out || n_tables < 0 || (n_tables > 0 && !table_bytes) || hbm_bytes <= 0 || reserve_bytes < 0 || world < 1)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad placement arguments");
  if (prefer_mode != FCP_PLACE_COLUMN_SHARD && prefer_mode != FCP_PLACE_ROW_SHARD && prefer_mode != FCP_PLACE_MIXED)
    return fail(FCP_ERR_INVALID_ARGUMENT, "prefer_mode must be column sharding, row sharding or mixed");
  const int64_t budget = hbm_bytes - reserve_bytes;
  if (budget <= 0) return fail(FCP_ERR_INVALID_ARGUMENT, "reserve_bytes leaves no room for tables");
  int64_t total = 0, largest = 0;
  for (int32_t t = 0; t < n_tables; ++t) {
    if (table_bytes[t] < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "negative table size");
    total += table_bytes[t];
    largest = std::max(largest, table_bytes[t]);
  }
  out->min_world = (int32_t)std::max<int64_t>(1, (total + budget - 1) / budget);
  out->mode = FCP_PLACE_REPLICATE;
  out->bytes_per_gpu = total;
  if (owner)
    for (int32_t t = 0; t < n_tables; ++t) owner[t] = 0;
  if (total <= budget) return FCP_OK; // fits one GPU: replicas, no collective
  // row sharding: every table contributes ceil(rows / world) rows to every GPU (at most one row's worth of
  // rounding per table, ignored here: tables are >> one row)
  const int64_t row_share = (total + world - 1) / world;
  const bool row_ok = world > 1 && row_share <= budget;
  // whole tables, longest first onto the least loaded rank (longest-processing-time packing), on top of the row
  // share of the tables that are spread: `spread_over` = the threshold above which a table is spread by rows
  std::vector<int32_t> assign(n_tables, -1);
  int32_t n_whole = 0; // tables the last pack() left whole
  auto pack = [&](int64_t spread_over, int64_t *share) {
    int64_t spread = 0;
    std::vector<int32_t> order;
    for (int32_t t = 0; t < n_tables; ++t) {
      if (table_bytes[t] > spread_over) {
        spread += table_bytes[t];
        assign[t] = -1;
      } else {
        order.push_back(t);
      }
    }
    std::vector<int64_t> load(world, (spread + world - 1) / world);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return table_bytes[a] > table_bytes[b]; });
    for (int32_t t : order) {
      const int32_t r = (int32_t)(std::min_element(load.begin(), load.end()) - load.begin());
      load[r] += table_bytes[t];
      assign[t] = r;
    }
    *share = *std::max_element(load.begin(), load.end());
    n_whole = (int32_t)order.size();
    return *share <= budget;
  };
  int64_t col_share = 0, mixed_share = 0;
  const bool col_ok = world > 1 && largest <= budget && pack(INT64_MAX, &col_share);
  int mode;
  if (prefer_mode == FCP_PLACE_MIXED) {
    if (col_ok) mode = FCP_PLACE_COLUMN_SHARD;                                       // every table fits a GPU: no rows spread at all
    // (MIXED needs a whole table for every rank — the whole-column step gives every rank a block; with fewer, the few
    // small tables are spread by rows like the large ones: their partial sums are a rounding error on the wire)
    else if (world > 1 && largest > budget && pack(budget, &mixed_share) && n_whole >= world) mode = FCP_PLACE_MIXED;
    else if (row_ok) mode = FCP_PLACE_ROW_SHARD;
    else mode = -1;
  } else {
    if (!row_ok && !col_ok) mode = -1;
    else mode = (col_ok && (prefer_mode == FCP_PLACE_COLUMN_SHARD || !row_ok)) ? FCP_PLACE_COLUMN_SHARD : FCP_PLACE_ROW_SHARD;
  }
  if (mode < 0)
    return fail(FCP_ERR_UNSUPPORTED, "tables of " + std::to_string(total) + " bytes do not fit " + std::to_string(world) +
                                         " GPU(s) with " + std::to_string(budget) + " bytes each: needs at least " +
                                         std::to_string(out->min_world));
  out->mode = mode;
  if (mode == FCP_PLACE_COLUMN_SHARD) {
    (void)pack(INT64_MAX, &col_share); // (the mixed attempt may have run after it)
    out->bytes_per_gpu = col_share;
  } else if (mode == FCP_PLACE_MIXED) {
    out->bytes_per_gpu = mixed_share;
  } else {
    out->bytes_per_gpu = row_share;
    std::fill(assign.begin(), assign.end(), -1);
  }
  if (owner)
    for (int32_t t = 0; t < n_tables; ++t) owner[t] = assign[t];
  return FCP_OK;
}

int fcp_placement_decide(const int64_t *table_bytes, int32_t n_tables, int64_t hbm_bytes, int64_t reserve_bytes,
                         int32_t world, int32_t prefer_mode, fcp_placement_t *out) {
  return fcp_placement_assign(table_bytes, n_tables, hbm_bytes, reserve_bytes, world, prefer_mode, nullptr, out);
}

int fcp_plan_release_captures(fcp_plan_t *p) {
  if (!p) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan");
  std::lock_guard<std::mutex> lock(p->mu);
  for (auto &s : p->slots) {
    if (!s.captured) continue;
    s.captured = false;
    s.done_valid = false; // its readers were graph replays: the next installation drains the stream / device
  }
  return FCP_OK;
}

int fcp_plan_destroy(fcp_plan_t *p) {
  if (!p) return FCP_OK;
  if (!p->host_only) {
    DeviceGuard guard;
    if (guard.enter(p->desc.device) == FCP_OK) {
      (void)hipDeviceSynchronize();
      pending_forget(p); // (the lanes belong to the device's pool and stay)
      if (p->pool && p->lane_relies.exchange(false)) p->pool->n_relying.fetch_sub(1, std::memory_order_acq_rel);
      for (hipEvent_t *ev : {&p->sup.b0, &p->sup.b1, &p->sup.w0, &p->sup.w1}) // the supervisor's timing events
        if (*ev) {
          (void)hipEventDestroy(*ev);
          *ev = nullptr;
        }
      destroy_device(p);
    }
  }
  delete p;
  return FCP_OK;
}

int fcp_plan_group_width(const fcp_plan_t *p, int32_t group, int32_t *width) {
  if (!p || !width || group < 0 || group >= p->desc.n_groups)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad group");
  *width = p->group_width[group];
  return FCP_OK;
}

int fcp_plan_column_offset(const fcp_plan_t *p, int32_t column, int32_t *offset) {
  if (!p || !offset || column < 0 || column >= (int32_t)p->cols.size())
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad column");
  *offset = p->cols[column].out_off;
  return FCP_OK;
}

int fcp_plan_arena_bytes(fcp_plan_t *p, const int32_t *concated_shapes, const int32_t *symbols,
                         int64_t *bytes) {
  if (!p || !concated_shapes || !bytes) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<FcpColDyn> dyn(p->cols.size());
  std::vector<int32_t> offsets(p->ranks.size(), 0);
  DynMeta m;
  int rc = compute_dyn(p, offsets.data(), concated_shapes, symbols, -1, dyn.data(), &m);
  if (rc) return rc;
  *bytes = m.arena_bytes;
  return FCP_OK;
}

int fcp_plan_read_bad_ids(fcp_plan_t *p, void *stream, int64_t *count) {
  if (!p || !count) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  *count = 0;
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan");
  if (!p->d_bad) return FCP_OK;
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  unsigned long long v = 0;
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  if (p->pool && p->lane_count > 0) { // requests of that stream may have run on a private lane
    std::lock_guard<std::mutex> cal_lock(p->pool->cal_mu);
    for (auto &l : p->pool->lanes)
      if (l->stream) HIP_TRY(hipStreamSynchronize(l->stream));
  }
  HIP_TRY(hipMemcpy(&v, p->d_bad, sizeof(v), hipMemcpyDeviceToHost));
  *count = (int64_t)v;
  return FCP_OK;
}

} // extern "C"

#if defined(FCP_STAMPS)
// diagnostic builds only: per-block timestamps of the LAST dense launch (8 x u64 per block, 100 MHz ticks)
extern "C" int fcp_debug_read_stamps(fcp_plan_t *p, unsigned long long *out, int n_blocks) {
  if (!p || !p->d_stamps || n_blocks > 65536) return FCP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, p->d_stamps, 8 * sizeof(unsigned long long) * (size_t)n_blocks, hipMemcpyDeviceToHost));
  return FCP_OK;
}
#endif

