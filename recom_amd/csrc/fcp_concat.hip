// fcp_concat.hip — Addons>ConcatOutputs[NoHost] (concat_outputs_op_gpu.cu.cc:133-140, host inputs :186-216) behind the ABI and
// fcp_shard_finalize.  Carved out of fcp_api.hip in round 6 (see fcp_host.h).
#include "fcp_host.h"

extern "C" {

// ---- ConcatOutputs ----------------------------------------------------------------
int fcp_concat_outputs(const void *const *inputs, const int32_t *dims, int32_t n, int64_t prefix_size,
                       void *out, void *stream) {
  if (n < 0 || prefix_size < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "negative size");
  if (n == 0 || prefix_size == 0) return FCP_OK;
  if (!inputs || !dims || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  int64_t width = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (dims[k] <= 0 || !inputs[k]) return fail(FCP_ERR_INVALID_ARGUMENT, "bad concat input");
    width += dims[k];
  }
  if (width > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "concat width exceeds 2^31");
  // per-column layout: the inputs are columns of FeatureColumnProcess arenas (output_ptrs); with private streams the
  // kernels that fill them run elsewhere
  if (int rc = wait_for_inputs(inputs, n, stream)) return rc;
  const int e = fcp_launch_concat_outputs(inputs, dims, nullptr, nullptr, n, prefix_size, (int32_t)width, 0, out,
                                          static_cast<hipStream_t>(stream));
  if (e) return hip_fail("concat-outputs launch", (hipError_t)e);
  return FCP_OK;
}

namespace {
int check_scatter_args(const void *const *inputs, const int32_t *dims, const int32_t *col_offsets, int32_t n,
                       int64_t prefix_size, int32_t out_width, const void *out) {
  if (n < 0 || prefix_size < 0 || out_width < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "negative size");
  if (n == 0 || prefix_size == 0) return FCP_OK;
  if (!inputs || !dims || !col_offsets || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  for (int32_t k = 0; k < n; ++k) {
    if (dims[k] <= 0 || !inputs[k]) return fail(FCP_ERR_INVALID_ARGUMENT, "bad concat input");
    if (col_offsets[k] < 0 || (int64_t)col_offsets[k] + dims[k] > out_width)
      return fail(FCP_ERR_INVALID_ARGUMENT, "concat input " + std::to_string(k) + " does not fit the output row");
  }
  return FCP_OK;
}

// Pinned staging for Addons>ConcatOutputs host inputs: a small ring per device, grown on demand.  A slot is
// reused once the copy that read it has completed (its event), so the call never waits for the GPU in the
// steady state and never hands pageable memory to an asynchronous copy.
struct HostStageSlot {
  char *buf = nullptr;
  char *buf_dev = nullptr; // device mapping of buf
  size_t cap = 0;
  hipEvent_t copied = nullptr;
  bool busy = false; // reserved by a call that is packing into it / enqueueing its reader (guarded by the ring mutex)
};
struct HostStageRing {
  std::mutex mu; // covers `next` and the slots' `busy` flags only: callers pack and launch outside it
  HostStageSlot slots[4];
  size_t next = 0;
};
HostStageRing *host_stage_ring(int device) {
  static std::mutex mu;
  static std::vector<HostStageRing *> rings;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)rings.size() <= device) rings.resize(device + 1, nullptr);
  if (!rings[device]) rings[device] = new HostStageRing(); // lives as long as the process (pinned memory is freed at exit)
  return rings[device];
}
} // namespace

int fcp_concat_outputs_scatter_strided(const void *const *inputs, const int32_t *dims, const int32_t *in_strides,
                                       const int32_t *col_offsets, int32_t n, int64_t prefix_size, int32_t out_width, void *out,
                                       void *stream) {
  int rc = check_scatter_args(inputs, dims, col_offsets, n, prefix_size, out_width, out);
  if (rc || n == 0 || prefix_size == 0) return rc;
  if (in_strides)
    for (int32_t k = 0; k < n; ++k)
      if (in_strides[k] < dims[k]) return fail(FCP_ERR_INVALID_ARGUMENT, "input row stride smaller than its width");
  // inputs (and an `out` that lies in an arena with external slots) may be results of private-stream requests
  if ((rc = wait_for_inputs(inputs, n, stream))) return rc;
  if ((rc = wait_for_inputs(&out, 1, stream))) return rc;
  const int e = fcp_launch_concat_outputs(inputs, dims, col_offsets, in_strides, n, prefix_size, out_width, 0, out,
                                          static_cast<hipStream_t>(stream));
  if (e) return hip_fail("concat-outputs launch", (hipError_t)e);
  return FCP_OK;
}

int fcp_concat_outputs_scatter(const void *const *inputs, const int32_t *dims, const int32_t *col_offsets, int32_t n,
                               int64_t prefix_size, int32_t out_width, void *out, void *stream) {
  return fcp_concat_outputs_scatter_strided(inputs, dims, nullptr, col_offsets, n, prefix_size, out_width, out, stream);
}

int fcp_concat_outputs_host(const void *const *host_inputs, const int32_t *dims, const int32_t *col_offsets, int32_t n,
                            int64_t prefix_size, int32_t out_width, void *out, fcp_alloc_fn malloc_temp,
                            void *malloc_temp_ctx, int32_t device, void *stream_) {
  int rc = check_scatter_args(host_inputs, dims, col_offsets, n, prefix_size, out_width, out);
  if (rc || n == 0 || prefix_size == 0) return rc;
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  rc = fcp_result_wait(out, stream_); // `out` lies in an arena whose lookup kernel may run on a private stream: write after it
  if (rc) return rc;
  size_t total = 0;
  std::vector<size_t> at(n);
  for (int32_t k = 0; k < n; ++k) {
    at[k] = total;
    total += ((size_t)prefix_size * dims[k] * 4 + 15) / 16 * 16;
  }
  // Small payloads (the reference's models E / F: 32 dense features, 64 KB per request): the scatter kernel reads the
  // pinned slot through its device mapping — no copy, no device staging buffer, one runtime call less per request.
  // Large ones keep the H2D copy (the scatter would hold its CUs for the length of the PCIe transfer).
  static const size_t direct_max = (size_t)fcp::diag_ll("concat_host_direct_max", 1 << 20); // tuning aid: bytes (0: always copy)
  const bool direct = total <= direct_max;
  if (!direct && !malloc_temp) return fail(FCP_ERR_INVALID_ARGUMENT, "malloc_temp callback is required for payloads that are copied to the device");
  HostStageRing *ring = host_stage_ring(device);
  // Reserve a slot under the ring lock, then pack (the reference: one memcpy per input into a std::vector,
  // concat_outputs_op_gpu.cu.cc:195-201), wait for the slot's previous reader if it is still running, and enqueue
  // OUTSIDE it: concurrent serve workers (models E / F) only meet on the bookkeeping.
  HostStageSlot *slp = nullptr;
  for (;;) {
    {
      std::lock_guard<std::mutex> lock(ring->mu);
      for (int t = 0; t < 4 && !slp; ++t) {
        HostStageSlot &c = ring->slots[(ring->next + t) % 4];
        if (!c.busy) {
          c.busy = true;
          ring->next = (ring->next + t + 1) % 4;
          slp = &c;
        }
      }
    }
    if (slp) break;
    std::this_thread::yield(); // more than four calls in flight on this device
  }
  HostStageSlot &sl = *slp;
  struct Release {
    HostStageRing *r;
    HostStageSlot *s;
    ~Release() {
      std::lock_guard<std::mutex> lock(r->mu);
      s->busy = false;
    }
  } release{ring, slp};
  if (sl.copied && hipEventQuery(sl.copied) != hipSuccess) HIP_TRY(hipEventSynchronize(sl.copied));
  if (sl.cap < total) {
    if (sl.buf) HIP_TRY(hipHostFree(sl.buf));
    sl.buf = nullptr;
    sl.buf_dev = nullptr;
    sl.cap = 0;
    const size_t cap = std::max<size_t>(total, 1 << 16);
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&sl.buf), cap, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&sl.buf_dev), sl.buf, 0));
    sl.cap = cap;
  }
  if (!sl.copied) HIP_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
  for (int32_t k = 0; k < n; ++k) std::memcpy(sl.buf + at[k], host_inputs[k], (size_t)prefix_size * dims[k] * 4);
  const char *src = sl.buf_dev;
  if (!direct) {
    char *d_stage = static_cast<char *>(malloc_temp(malloc_temp_ctx, total));
    if (!d_stage) return fail(FCP_ERR_ALLOC, "malloc_temp returned NULL");
    HIP_TRY(hipMemcpyAsync(d_stage, sl.buf, total, hipMemcpyHostToDevice, stream));
    src = d_stage;
  } else {
    __atomic_thread_fence(__ATOMIC_SEQ_CST); // the packed bytes are in memory before the launch that reads them is queued
  }
  std::vector<const void *> d_in(n);
  for (int32_t k = 0; k < n; ++k) d_in[k] = src + at[k];
  const int e = fcp_launch_concat_outputs(d_in.data(), dims, col_offsets, nullptr, n, prefix_size, out_width, 0, out, stream);
  if (e) return hip_fail("concat-outputs launch", (hipError_t)e);
  HIP_TRY(hipEventRecord(sl.copied, stream)); // the slot is free once its last reader (copy or scatter) has run
  return FCP_OK;
}

// ---- row-shard finalize -------------------------------------------------------------
int fcp_shard_finalize(fcp_plan_t *p, const fcp_process_args_t *a, int32_t group, const void *partial_slices,
                       int32_t world, int64_t row_begin, int64_t row_count, void *out, void *stream_) {
  if (!p || !a || !partial_slices || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan cannot run");
  if (group < 0 || group >= p->desc.n_groups || world < 1 || row_begin < 0 || row_count < 0)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad group / world / rows");
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  fcp_process_args_t args = *a;
  args.stream = stream_;
  // same locking discipline as fcp_process_feature_columns: the plan mutex covers the slot bookkeeping only
  DynSlot *slot = nullptr;
  bool install = false;
  thread_local std::vector<int32_t> key;
  build_key(p, &args, key);
  const bool capturing = stream_is_capturing(stream); // same rule as the process call: nothing is installed inside a capture
  {
    std::unique_lock<std::mutex> lock(p->mu);
    while ((rc = find_or_reserve(p, key, &slot, &install, capturing)) == kAllSlotsBusy) {
      lock.unlock();
      std::this_thread::yield();
      lock.lock();
    }
    if (rc == kNeedsInstall)
      return fail(FCP_ERR_UNSUPPORTED, "stream capture of fcp_shard_finalize for shapes that are not resident: run the request once on "
                                       "this stream before capturing (descriptors cannot be installed inside a capture)");
    if (rc) return rc;
  }
  SlotUnpin unpin{p, slot, false};
  if (install) {
    rc = install_slot(p, &args, *slot);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(p->mu);
    publish_slot(p, *slot, key, args.stream);
  }
  const DynMeta &m = slot->meta;
  if (row_begin + row_count > m.group_rows[group]) return fail(FCP_ERR_SHAPE_MISMATCH, "row slice out of range");
  FcpLaunch L;
  void *scratch = nullptr;
  bool need_csr = false;
  for (int k : p->seg_cols)
    if (p->cols[k].d.form == FCP_FORM_SEGMENT_REDUCE && p->cols[k].d.combiner == FCP_COMBINER_MEAN) need_csr = true;
  if (need_csr) {
    if (!a->malloc_temp) return fail(FCP_ERR_INVALID_ARGUMENT, "malloc_temp is required for mean columns with segment ids");
    const int64_t bytes = m.arena_bytes - m.csr_arena_off;
    scratch = a->malloc_temp(a->malloc_temp_ctx, (size_t)std::max<int64_t>(bytes, 1));
    if (!scratch) return fail(FCP_ERR_ALLOC, "malloc_temp returned NULL");
  }
  fill_launch(p, *slot, 1, a->concated_inputs, scratch, 0, &L);
  L.csr_arena_off = 0;
  L.csr_reg = nullptr; // (the scratch here is a buffer of its own: no regular-CSR shortcut)
  L.groups[0].csr_reg_stride = 0;
  if (need_csr) {
    FcpSegLaunch S;
    S.seg_cols = p->d_seg_cols;
    S.cols = p->d_cols;
    S.dyn = slot->d_dyn;
    S.blob = L.blob;
    S.arena = static_cast<char *>(scratch);
    S.bad_ids = nullptr; // the partial pass has counted already
    S.xforms = p->d_xforms;
    S.segmaps = p->d_segmaps;
    S.skip_inverse = 1;  // only the row lengths of mean columns are wanted here
    S.csr_arena_off = 0;
    const int e = fcp_launch_segment_offsets(S, (int)p->seg_cols.size(), m.max_seg_nnz, stream);
    if (e) return hip_fail("segment-offsets launch", (hipError_t)e);
  }
  const int e = fcp_launch_shard_finalize(L, group, static_cast<const float *>(partial_slices), world, row_begin,
                                          row_count, static_cast<float *>(out), p->vec, stream);
  if (e) return hip_fail("shard-finalize launch", (hipError_t)e);
  if (install) { // the descriptors were installed by this call
    HIP_TRY(hipEventRecord(slot->done, stream));
    unpin.recorded = true;
  }
  return FCP_OK;
}

} // extern "C"
