// fcp_process.hip — the request path: ProcessFeatureColumns (cuda_emitter.cc:2303-2494) and its kernel caller (:2139-2258):
// table binding, the descriptor-slot cache, launch records, fcp_internal_process, fcp_process_feature_columns.
// Carved out of fcp_api.hip in round 6 (see fcp_host.h).
#include "fcp_host.h"

namespace fcph {

bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess) return st == hipStreamCaptureStatusActive;
  (void)hipGetLastError();
  return false;
}

// Bind the table addresses (FeatureColumnProcess `inputs`).  TF variables keep their address between requests, so this
// uploads once.  `capturing`: the caller's stream is being captured into a HIP graph — binding (or re-binding) tables copies records
// and may synchronise the device, neither of which a capture tolerates: refused, the plan stays as it was.
int bind_tables(fcp_plan *p, const void *const *input_ptrs, bool capturing) {
  const int nt = p->desc.n_device_inputs;
  if (p->tables_bound && std::memcmp(p->bound_tables.data(), input_ptrs, nt * sizeof(void *)) == 0)
    return FCP_OK;
  if (capturing)
    return fail(FCP_ERR_UNSUPPORTED, "stream capture of a request whose tables are not bound to the plan yet (or have moved): run it "
                                     "once on this stream before capturing");
  // validate first and build the new records aside: a failure leaves the plan exactly as it was
  std::vector<FcpColStatic> cols = p->h_cols;
  for (size_t k = 0; k < p->cols.size(); ++k) {
    const int t = p->cols[k].d.table_input;
    const bool lookup = p->cols[k].d.form == FCP_FORM_GATHER || p->cols[k].d.form == FCP_FORM_SEGMENT_REDUCE ||
                        p->cols[k].d.form == FCP_FORM_GATHER_SCATTER;
    if (lookup) {
      // (a row shard of a table with fewer rows than ranks may be empty: no id maps to it, nothing is read)
      const bool empty_shard = p->desc.shard_world > 1 && p->cols[k].d.vocab <= p->desc.shard_rank;
      if (!input_ptrs[t] && !empty_shard) return fail(FCP_ERR_INVALID_ARGUMENT, "null table pointer");
      cols[p->pos_of[k]].table = static_cast<const float *>(input_ptrs[t]);
    }
  }
  if (p->tables_bound) HIP_TRY(hipDeviceSynchronize()); // in-flight requests still read the old tables
  p->tables_bound = false;                               // until the upload below has succeeded
  HIP_TRY(hipMemcpy(p->d_cols, cols.data(), cols.size() * sizeof(FcpColStatic), hipMemcpyHostToDevice));
  p->h_cols.swap(cols);
  for (int t = 0; t < nt; ++t) p->bound_tables[t] = input_ptrs[t];
  p->tables_bound = true;
  return FCP_OK;
}

// `done` only tells the host that a kernel has finished READING the slot; nothing the host or another
// device reads depends on it, so it needs no system-scope fence (a fenced record costs 2.9 us of GPU
// timeline between two kernels, an unfenced one 1.1 us: RAGGED with new shapes 33.5 -> 31.5 us).
int done_event_for(DynSlot &s, void *stream) {
  s.done_gen = 0;
  for (auto &e : s.done_pool)
    if (e.first == stream) {
      s.done = e.second;
      return FCP_OK;
    }
  hipEvent_t ev = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
  s.done_pool.emplace_back(stream, ev);
  s.done = ev;
  return FCP_OK;
}

// Device-resident dynamic descriptors for a request's shapes, in three steps so that the plan
// mutex is only held for the bookkeeping:
//   find_or_reserve (locked)   a slot that already holds these shapes for this stream, or a victim
//                              reserved (pinned, invalid) for this request to fill;
//   install_slot    (unlocked) wait for the victim's old readers, evaluate the shapes, write the
//                              records through the BAR (or enqueue the upload kernel);
//   publish_slot    (locked)   make the slot findable.
void build_key(const fcp_plan *p, const fcp_process_args_t *a, std::vector<int32_t> &key) {
  const int nh = (int)p->ranks.size();
  const int nsym = a->symbols ? p->desc.n_symbols : 0;
  // The stream is part of the key: descriptors are never shared between streams, so the `done` event
  // of a slot (recorded on its stream) or a synchronisation of that stream covers every kernel that read it.
  const size_t key_len = (size_t)nh + p->rank_sum + nsym + 3;
  key.resize(key_len);
  std::memcpy(key.data(), a->concated_offsets, nh * sizeof(int32_t));
  std::memcpy(key.data() + nh, a->concated_shapes, p->rank_sum * sizeof(int32_t));
  if (nsym) std::memcpy(key.data() + nh + p->rank_sum, a->symbols, nsym * sizeof(int32_t));
  key[key_len - 3] = (int32_t)std::min<int64_t>(a->concated_bytes, 0x7fffffff);
  const uint64_t stream_bits = (uint64_t)reinterpret_cast<uintptr_t>(a->stream);
  key[key_len - 2] = (int32_t)(uint32_t)stream_bits;
  key[key_len - 1] = (int32_t)(uint32_t)(stream_bits >> 32);
}

// `capturing`: the request's stream is being captured into a HIP graph.  The launch that gets recorded bakes
// in the slot's device address, geometry and arena pointer and may be replayed at any later time: its slot
// is marked `captured` and never evicted again; shapes that are not resident cannot be installed while
// capturing (the installation synchronises and writes descriptors NOW, a replay would read whatever the
// slot holds THEN): kNeedsInstall.
int find_or_reserve(fcp_plan *p, const std::vector<int32_t> &key, DynSlot **out, bool *install, bool capturing) {
  ++p->tick;
  DynSlot *victim = nullptr;
  int pinned = 0;
  for (auto &s : p->slots) {
    if (s.valid && s.key == key) {
      s.tick = p->tick;
      ++s.users;
      ++s.uses;
      s.done_valid = false; // one more reader that `done` (recorded by the installer) does not cover
      if (capturing) s.captured = true;
      *out = &s;
      *install = false;
      return FCP_OK;
    }
    if (s.captured) {
      ++pinned;
      continue;
    }
    if (s.users > 0) continue; // being filled, or its kernels are being enqueued right now
    // preference: an empty slot; then the least recently used slot whose `done` event covers all of its
    // readers (one event wait); only then slots that were hit again after they were installed
    auto rank = [](const DynSlot &x) { return !x.valid ? 0 : (x.done_valid ? 1 : 2); };
    if (!victim || rank(s) < rank(*victim) || (rank(s) == rank(*victim) && s.tick < victim->tick)) victim = &s;
  }
  if (capturing) return kNeedsInstall;
  if (!victim)
    return pinned == kSlots ? fail(FCP_ERR_UNSUPPORTED, "every descriptor slot belongs to a captured graph: fcp_plan_release_captures")
                            : kAllSlotsBusy; // more concurrent requests than slots: the caller retries
  victim->was_valid = victim->valid;
  victim->valid = false;
  victim->users = 1;
  *out = victim;
  *install = true;
  return FCP_OK;
}

int install_slot(fcp_plan *p, const fcp_process_args_t *a, DynSlot &s) {
  hipStream_t stream = static_cast<hipStream_t>(a->stream);
  if (s.was_valid) {
    // Kernels of earlier requests may still read the slot (the host runs ahead of the GPU): wait
    // for the last one.  `done` covers it unless the slot was hit again after its installation;
    // then everything enqueued on its stream so far has to drain.
    if (!s.done_valid) {
      if (s.stream == a->stream) {
        HIP_TRY(hipStreamSynchronize(stream));
      } else {
        HIP_TRY(hipDeviceSynchronize());
      }
    } else if (s.done_gen != 0 && s.done_gen != g_lane_generation.load(std::memory_order_acquire)) {
      // `done` was a private lane's event and those lanes have been let go of since: their kernels have all run
    } else if (hipEventQuery(s.done) != hipSuccess) {
      HIP_TRY(hipEventSynchronize(s.done)); // back-pressure: at most kSlots requests in flight
    }
    if (!p->host_writes_dyn && hipEventQuery(s.uploaded) != hipSuccess) HIP_TRY(hipEventSynchronize(s.uploaded));
  }
  static const bool stats = fcp::diag_on("install_stats"); // diagnostic: where a descriptor installation spends its host time
  auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const uint64_t t0 = stats ? now_ns() : 0;
  int rc = compute_dyn(p, a->concated_offsets, a->concated_shapes, a->symbols, a->concated_bytes, s.h_dyn, &s.meta);
  if (rc) return rc;
  p->last_work_bytes.store(s.meta.work_bytes, std::memory_order_relaxed);
  const uint64_t t1 = stats ? now_ns() : 0;
  const size_t dyn_bytes = p->cols.size() * sizeof(FcpColDyn);
  if (p->host_writes_dyn) {
    std::memcpy(s.d_dyn, s.h_dyn, dyn_bytes); // CPU stores through the BAR into fine-grained VRAM
    __builtin_ia32_sfence();                  // posted before the launch's doorbell write
    if (stats) {
      static std::atomic<uint64_t> n{0}, ns_dyn{0}, ns_bar{0};
      const uint64_t t2 = now_ns();
      ns_dyn += t1 - t0;
      ns_bar += t2 - t1;
      if ((++n & 1023) == 0)
        std::fprintf(stderr, "fcp install: %llu installs, shapes -> records %.2f us, records -> device (%zu bytes through the BAR) %.2f us\n",
                     (unsigned long long)n.load(), ns_dyn.load() / 1e3 / n.load(), dyn_bytes, ns_bar.load() / 1e3 / n.load());
    }
  } else {
    const int e = fcp_launch_upload(s.h_dyn_dev, s.d_dyn, dyn_bytes, stream);
    if (e) return hip_fail("descriptor upload launch", (hipError_t)e);
    HIP_TRY(hipEventRecord(s.uploaded, stream));
  }
  return done_event_for(s, a->stream);
}

void publish_slot(fcp_plan *p, DynSlot &s, const std::vector<int32_t> &key, void *stream) {
  s.done_valid = false;
  s.uses = 1;
  s.key = key;
  s.stream = stream;
  s.tick = p->tick;
  s.valid = true;
}

// Which cache policy the output stores of a request take (FcpLaunch::store_through bits 0 and 2; st_out in fcp_kernels.hip):
//   * outputs that FIT the L2s (below FCP_STORE_THROUGH_BYTES, 32 MiB: DLRM 3.5 MB, RAGGED 15.7 MB, models E / F 16-20 MB):
//     `nt`, whatever the arena — plain stores leave the whole output dirty in L2 for the kernel boundary to write back
//     (model F 10.5-10.9 us with nt against 12.4-13.4 with plain stores, one arena or six; RAGGED / DLRM within 0.2 us);
//   * larger outputs into the arena this plan's previous request wrote, or the one before it (TF's allocate_output hands a
//     serving loop the block it just freed, feature_column_process_op_gpu.cu.cc:107-111), up to 160 MiB: PLAIN stores — the
//     lines are still in the Infinity Cache / L2 and rewriting a resident line beats streaming it (S2, one arena: 27.2 us
//     against 27.9 nt / 28.0 sc1 nt; two arenas 28.0 / 28.2 / 28.2; batch 1024, 123 MB: 50.2 against 53.2; batch 2048,
//     246 MB: plain LOSES, 109 against 103);
//   * larger outputs into any other arena (a ring of three or more, fresh memory): plain stores lose 3 us of 28 there —
//     write-through `sc1 nt`
//   (profiles/r06_arena_reuse_store_policy.txt).  A performance hint only: read and updated without the plan's mutex.
int store_policy_for(fcp_plan *p, const void *arena, int64_t out_bytes) {
  const int64_t through_bytes = p->env.store_through_bytes;
  // tuning aid: FCP_DIAG=store_plain_reuse=0 never plain stores, 2 always, default 1 = the rule above
  static const int reuse_mode = (int)fcp::diag_ll("store_plain_reuse", 1);
  const uintptr_t ar = reinterpret_cast<uintptr_t>(arena);
  const uintptr_t a0 = p->recent_arena[0].load(std::memory_order_relaxed), a1 = p->recent_arena[1].load(std::memory_order_relaxed);
  const bool reused = ar == a0 || ar == a1;
  if (ar != a0) {
    p->recent_arena[1].store(a0, std::memory_order_relaxed);
    p->recent_arena[0].store(ar, std::memory_order_relaxed);
  }
  constexpr int64_t kPlainMaxBytes = (int64_t)160 << 20;
  if (reuse_mode == 2) return 4;
  if (out_bytes < through_bytes) return 0;
  return (reuse_mode == 1 && reused && out_bytes <= kPlainMaxBytes) ? 4 : 1;
}

void fill_launch(const fcp_plan *p, const DynSlot &s, int kind, const void *blob, void *arena, int store_policy, FcpLaunch *L) {
  L->slot_map = p->d_slot_map;
  L->span_list = p->d_span_list;
  L->cols = p->d_cols;
  L->xforms = p->d_xforms;
  L->zeros = p->d_zeros;
  L->dyn = s.d_dyn;
  L->blob = static_cast<const char *>(blob);
  L->arena = static_cast<char *>(arena);
  L->bad_ids = p->d_bad;
  L->stamps = p->d_stamps;
  L->csr_arena_off = s.meta.csr_arena_off;
  L->shard_rank = p->desc.shard_rank;
  L->shard_world = p->desc.shard_world;
  L->n_groups = p->desc.n_groups;
  L->rows_per_wave = s.meta.geo[kind].rows_per_wave;
  L->seg_search = s.meta.seg_search ? 1 : 0;
  L->store_through = store_policy | (p->wide_rows ? 2 : 0); // (store_policy_for: bit 0 write-through, bit 2 plain stores)
  L->csr_reg = s.meta.csr_reg_mode == 1   ? reinterpret_cast<const int32_t *>(static_cast<const char *>(arena) + s.meta.csr_arena_off)
               : s.meta.csr_reg_mode == 2 ? reinterpret_cast<const int32_t *>(static_cast<const char *>(blob) + s.meta.csr_reg_base)
                                          : nullptr;
  for (int g = 0; g < p->desc.n_groups; ++g) L->groups[g] = s.meta.geo[kind].groups[g];
  L->groups[0].csr_reg_stride = (s.meta.csr_reg_mode && L->csr_reg) ? s.meta.csr_reg_stride : 0;
}

} // namespace fcph

extern "C" {

// ---- ProcessFeatureColumns ------------------------------------------------------
// The request on the stream it names (not part of the ABI: the sharded step calls it — its exchange follows on the same
// stream — and fcp_process_feature_columns below after it has chosen a private lane).
int fcp_internal_process(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  if (!p || !a) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan / args");
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan cannot run");
  if (!a->concated_offsets || !a->concated_shapes) {
    if (!p->ranks.empty()) return fail(FCP_ERR_INVALID_ARGUMENT, "null offsets / shapes");
  }
  if (!a->malloc_buff) return fail(FCP_ERR_INVALID_ARGUMENT, "malloc_buff callback is required");
  if (p->desc.n_device_inputs > 0 && !a->input_ptrs) return fail(FCP_ERR_INVALID_ARGUMENT, "null input_ptrs");
  if (p->desc.n_symbols > 0 && !a->symbols) return fail(FCP_ERR_INVALID_ARGUMENT, "plan needs symbols");
  if (a->input_shapes) { // optional cross-check of the table shapes
    for (const HostColumn &hc : p->cols) {
      const int f = hc.d.form;
      if (f == FCP_FORM_PASSTHROUGH || f == FCP_FORM_BATCH_COL_REDUCTION || f == FCP_FORM_EXTERNAL) continue;
      const int32_t *s = a->input_shapes + 2 * hc.d.table_input;
      const int64_t local_vocab = p->desc.shard_world > 1
                                      ? (hc.d.vocab - p->desc.shard_rank + p->desc.shard_world - 1) / p->desc.shard_world
                                      : hc.d.vocab;
      if (s[0] != local_vocab || s[1] != hc.d.dim)
        return fail(FCP_ERR_SHAPE_MISMATCH, "table shape does not match the plan");
    }
  }
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(a->stream);

  // The plan mutex covers table binding and the descriptor-slot bookkeeping only.  The slot is
  // pinned (`users`) while this request evaluates new shapes, calls the allocator and enqueues its
  // kernels outside the lock, so host threads serving different streams overlap (a launch alone is
  // ~4-5 us of HIP runtime: 3 serve workers 4.4 -> 1.8 us of host time per request, cached shapes).
  DynSlot *slot = nullptr;
  bool install = false;
  thread_local std::vector<int32_t> key;
  build_key(p, a, key);
  const bool capturing = stream_is_capturing(stream);
  {
    std::unique_lock<std::mutex> lock(p->mu);
    if (p->desc.n_device_inputs > 0) {
      rc = bind_tables(p, a->input_ptrs, capturing);
      if (rc) return rc;
    }
    while ((rc = find_or_reserve(p, key, &slot, &install, capturing)) == kAllSlotsBusy) {
      lock.unlock();
      std::this_thread::yield();
      lock.lock();
    }
    if (rc == kNeedsInstall)
      return fail(FCP_ERR_UNSUPPORTED, "stream capture of a request whose shapes are not resident: run it once on this stream "
                                       "before capturing (descriptors cannot be installed inside a capture)");
    if (rc) return rc;
  }
  SlotUnpin unpin{p, slot, false};
  if (install) {
    rc = install_slot(p, a, *slot); // on failure the slot stays invalid and is released by `unpin`
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(p->mu);
    publish_slot(p, *slot, key, a->stream);
  }
  const DynMeta &m = slot->meta;
  tl_work_bytes = m.work_bytes;
  // an empty blob (every input tensor empty: all bags empty) may come with a null pointer, as an empty
  // TF tensor does; nothing dereferences it then
  if (m.arena_bytes > 0 && !a->concated_inputs && !p->ranks.empty() && a->concated_bytes != 0)
    return fail(FCP_ERR_INVALID_ARGUMENT, "null blob");

  void *arena = a->malloc_buff(a->malloc_buff_ctx, (size_t)std::max<int64_t>(m.arena_bytes, 128)); // never a zero-size request
  if (!arena) return fail(FCP_ERR_ALLOC, "malloc_buff returned NULL");
  if (const LaneDep *dep = tl_lane_dep) {
    // a private-stream request: everything queued on the caller's stream UP TO THE ALLOCATION — the blob's producer, and
    // whatever still uses the memory the allocator has just handed out — before the lane's first command on the arena
    tl_lane_dep = nullptr;
    HIP_TRY(hipEventRecord(dep->in, dep->caller));
    HIP_TRY(hipStreamWaitEvent(dep->lane, dep->in, 0));
  }

  FcpLaunch L;
  const int store_policy = store_policy_for(p, arena, m.csr_arena_off);
  fill_launch(p, *slot, 1, a->concated_inputs, arena, store_policy, &L);
  if (!p->seg_cols.empty() && !m.seg_search) {
    FcpSegLaunch S;
    S.seg_cols = p->d_seg_cols;
    S.cols = p->d_cols;
    S.dyn = slot->d_dyn;
    S.blob = L.blob;
    S.arena = L.arena;
    S.bad_ids = p->d_bad;
    S.xforms = p->d_xforms;
    S.segmaps = p->d_segmaps;
    S.skip_inverse = 0;
    S.csr_arena_off = m.csr_arena_off;
    // any-order scatter columns build their inverse map with atomic max: their scratch — the tail of the CSR area, the
    // pooled columns' offsets in front of it are overwritten by the pre-pass anyway — starts from zero
    if (p->has_inverse) {
      const int64_t first = m.csr_arena_off + 4 * (int64_t)slot->h_dyn[p->pos_of[p->seg_cols[p->n_seg_plain]]].csr_base;
      if (m.arena_bytes > first) HIP_TRY(hipMemsetAsync(static_cast<char *>(arena) + first, 0, (size_t)(m.arena_bytes - first), stream));
    }
    // FCP_ORDER_INPUTS_READY: the pre-pass depends on nothing queued before it (blob complete, arena unused) unless this call
    // has queued the inverse-map memset or a descriptor upload in front of it
    const bool prepass_any_order = p->request_order == FCP_ORDER_INPUTS_READY && !capturing && !p->has_inverse &&
                                   !(install && !p->host_writes_dyn);
    const int e = fcp_launch_segment_offsets(S, (int)p->seg_cols.size(), m.max_seg_nnz, stream, prepass_any_order);
    if (e) return hip_fail("segment-offsets launch", (hipError_t)e);
  }
  // Freshly installed descriptors: the slot's `done` event rides on the dispatch packet of the request's last kernel (its
  // stop event) instead of being recorded behind it — one runtime call (~1.9 us of host time) and one marker packet less
  // per request with new shapes.  (A private-lane request has already claimed the stop event for its completion event.)
  static const bool done_on_kernel = fcp::diag_ll("done_stop_event", 1) != 0; // tuning aid: 0 = always record
  const bool attach_done = install && done_on_kernel && !fcp_stop_event_pending() && !capturing;
  if (attach_done) fcp_set_stop_event(slot->done);
  struct ClearStop { // (an early return between here and the launch must not leave the event armed for this thread's next launch)
    bool armed;
    ~ClearStop() {
      if (armed) fcp_set_stop_event(nullptr);
    }
  } clear_stop{attach_done};
  // FCP_ORDER_INPUTS_READY: nothing this kernel reads or writes depends on the commands queued in front of it — unless this
  // very call has queued some (segment-offset pre-pass, inverse-map memset, descriptor upload kernel): then stream order it is
  const bool queued_before = (!p->seg_cols.empty() && !m.seg_search) || (install && !p->host_writes_dyn);
  struct ClearAnyOrder {
    ~ClearAnyOrder() { fcp_set_any_order(false); }
  } clear_any_order;
  if (p->request_order == FCP_ORDER_INPUTS_READY && !queued_before && !capturing) fcp_set_any_order(true);
  // hybrid dispatch: spans with pooled columns -> ragged body, all other spans -> dense body
  if (m.geo[1].grid_blocks > 0 && m.geo[0].grid_blocks > 0) {
    FcpLaunch Ld;
    fill_launch(p, *slot, 0, a->concated_inputs, arena, store_policy, &Ld);
    const int e = fcp_launch_hybrid(Ld, m.geo[0].grid_blocks, L, m.geo[1].grid_blocks, p->vec, stream);
    if (e) return hip_fail("hybrid kernel launch", (hipError_t)e);
  } else if (m.geo[1].grid_blocks > 0) {
    const int e = fcp_launch_fused(L, p->vec, false, m.geo[1].grid_blocks, stream);
    if (e) return hip_fail("ragged kernel launch", (hipError_t)e);
  } else if (m.geo[0].grid_blocks > 0) {
    fill_launch(p, *slot, 0, a->concated_inputs, arena, store_policy, &L);
    const int e = fcp_launch_fused(L, p->vec, true, m.geo[0].grid_blocks, stream);
    if (e) return hip_fail("dense kernel launch", (hipError_t)e);
  }
  if (install) { // first kernel on freshly installed descriptors: lets a later install reuse the slot precisely
    const bool taken = attach_done && !fcp_stop_event_pending(); // the launcher took it: the kernel carries the event
    clear_stop.armed = false;
    if (attach_done && !taken) fcp_set_stop_event(nullptr);      // nothing was launched (an empty request)
    if (tl_lane_done) {
      // a private-stream request: its completion event — on the last kernel's dispatch packet, or recorded by the caller of
      // this function right behind it — IS "the readers of this slot have finished": no event of the slot's own (one runtime
      // call and one marker packet less per request with new shapes)
      slot->done = tl_lane_done;
      slot->done_gen = tl_lane_done_gen;
    } else if (!taken) {
      HIP_TRY(hipEventRecord(slot->done, stream));
    }
    unpin.recorded = true;
  }

  if (r) {
    const int nc = (int)p->cols.size();
    for (int k = 0; k < nc; ++k) {
      const FcpColDyn &d = slot->h_dyn[p->pos_of[k]];
      if (r->output_ptrs) r->output_ptrs[k] = static_cast<char *>(arena) + d.out_base;
      if (r->output_shapes) {
        r->output_shapes[2 * k] = d.rows;
        r->output_shapes[2 * k + 1] = p->cols[k].d.dim;
      }
      if (r->output_row_strides) r->output_row_strides[k] = d.out_stride;
    }
    for (int g = 0; g < p->desc.n_groups; ++g) {
      if (r->group_ptrs)
        r->group_ptrs[g] = p->desc.layout == FCP_LAYOUT_CONCAT ? static_cast<char *>(arena) + m.group_base[g] : nullptr;
      if (r->group_shapes) {
        r->group_shapes[2 * g] = m.group_rows[g];
        r->group_shapes[2 * g + 1] = p->group_width[g];
      }
    }
    r->buffer = arena;
    r->buffer_bytes = m.arena_bytes;
  }
  return FCP_OK;
}

int fcp_process_feature_columns(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  if (!p || !a) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan / args");
  if (!p->pool || p->lane_count == 0) return fcp_internal_process(p, a, r);
  return process_on_private_streams(p, a, r); // (experimental: fcp_lanes.hip)
}

} // extern "C"
