// fcp_lanes.hip — EXPERIMENTAL: plan-owned private streams (fcp_plan_set_private_streams, fcp_result_wait): the device's
// lanes, their verification and run-time supervision, the registry of results a consumer has to wait for, and the request
// path of a plan that has lanes (process_on_private_streams).  Opt-in; the default request path (fcp_process.hip) never comes
// here.  Carved out of fcp_api.hip in round 6 (see fcp_host.h).
#include "fcp_host.h"

namespace fcph {

std::atomic<uint64_t> g_lane_generation{1};
// the completion event of the private-stream request this thread is issuing (fcp_process_feature_columns -> fcp_internal_process)
thread_local hipEvent_t tl_lane_done = nullptr;
thread_local uint64_t tl_lane_done_gen = 0;

thread_local const LaneDep *tl_lane_dep = nullptr;
// gathered + written bytes of the request this thread processed last (DynMeta::work_bytes; the supervisor's unit of work)
thread_local int64_t tl_work_bytes = 0;

std::mutex g_lane_pools_mu;
std::map<int, LanePool *> g_lane_pools;
LanePool *lane_pool_for(int device) {
  std::lock_guard<std::mutex> lock(g_lane_pools_mu);
  LanePool *&lp = g_lane_pools[device];
  if (!lp) {
    lp = new LanePool();
    lp->lanes.reserve(kMaxPoolLanes);
  }
  return lp;
}

std::mutex g_pending_mu;
using PendingMap = std::map<uintptr_t, PendingResult>;
PendingMap g_pending;
// The same by the address range of the request's INPUT blob: the lane kernel is its last reader, and nothing on the caller's
// stream says when it has run.  TensorFlow's allocator cannot hand the blob out again before Addons>ConcatOutputs (which
// holds it as a `tensor_buffers` input and waits for the result) has been enqueued; the request stager recycles its ring by
// itself and asks here (stager_input_wait / stager_input_synchronize) before it overwrites a slot.
PendingMap g_pending_inputs;

void pending_put(PendingMap &m, const void *owner, const void *base, int64_t bytes, hipEvent_t done) { // (g_pending_mu held)
  const uintptr_t b = reinterpret_cast<uintptr_t>(base), e = b + (uintptr_t)std::max<int64_t>(bytes, 1);
  auto it = m.lower_bound(b);
  if (it != m.begin() && std::prev(it)->second.end > b) --it;
  while (it != m.end() && it->first < e) it = m.erase(it); // every older entry that overlaps this range
  m[b] = PendingResult{e, done, owner};
  if (m.size() > 256) // ranges that are never handed out again: drop what has completed
    for (auto j = m.begin(); j != m.end();)
      j = (j->first != b && hipEventQuery(j->second.done) == hipSuccess) ? m.erase(j) : std::next(j);
}

void pending_register(const void *owner, void *arena, int64_t bytes, const void *blob, int64_t blob_bytes, hipEvent_t done) {
  std::lock_guard<std::mutex> lock(g_pending_mu);
  pending_put(g_pending, owner, arena, bytes, done);
  if (blob && blob_bytes > 0) pending_put(g_pending_inputs, owner, blob, blob_bytes, done);
}

// the event of the last private-stream request that read [base, base + bytes), or nullptr (g_pending_mu held)
hipEvent_t pending_input_event(const void *base, int64_t bytes) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(base), e = b + (uintptr_t)std::max<int64_t>(bytes, 1);
  auto it = g_pending_inputs.lower_bound(b);
  if (it != g_pending_inputs.begin() && std::prev(it)->second.end > b) --it;
  return (it != g_pending_inputs.end() && it->first < e) ? it->second.done : nullptr;
}

// A request that stays on the caller's stream although its plan has private streams (below the work threshold, or the
// stream is being captured) writes its arena in stream order: an older entry for that memory would make the reader wait
// for an event that has nothing to do with it — harmless outside a capture, an isolation error inside one.
void pending_clear_range(void *arena, int64_t bytes) {
  const uintptr_t b = reinterpret_cast<uintptr_t>(arena), e = b + (uintptr_t)std::max<int64_t>(bytes, 1);
  std::lock_guard<std::mutex> lock(g_pending_mu);
  if (g_pending.empty()) return;
  auto it = g_pending.lower_bound(b);
  if (it != g_pending.begin() && std::prev(it)->second.end > b) --it;
  while (it != g_pending.end() && it->first < e) it = g_pending.erase(it);
}

void pending_forget(const void *owner) {
  std::lock_guard<std::mutex> lock(g_pending_mu);
  for (PendingMap *m : {&g_pending, &g_pending_inputs})
    for (auto j = m->begin(); j != m->end();) j = j->second.owner == owner ? m->erase(j) : std::next(j);
}

// The request stager is about to overwrite a device blob / a pinned buffer the kernels read directly: `stream` (the copy
// stream) or the host waits for the private-stream request that read it last.  (The stager's own `consumed` events are
// recorded on the caller's stream, which does not wait for private-stream kernels.)
bool stager_reader_wait_off() { // test aid: FCP_DIAG=stager_no_reader_wait reproduces the hazard the two functions below close
  static const bool off = fcp::diag_on("stager_no_reader_wait");
  return off;
}
int stager_input_wait(const void *base, int64_t bytes, hipStream_t stream) {
  if (stager_reader_wait_off()) return FCP_OK;
  std::lock_guard<std::mutex> lock(g_pending_mu); // (held over the runtime call: the owning plan may not go away meanwhile)
  if (hipEvent_t ev = pending_input_event(base, bytes)) HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
  return FCP_OK;
}
int stager_input_synchronize(const void *base, int64_t bytes) {
  if (stager_reader_wait_off()) return FCP_OK;
  std::lock_guard<std::mutex> lock(g_pending_mu);
  if (hipEvent_t ev = pending_input_event(base, bytes))
    if (hipEventQuery(ev) != hipSuccess) HIP_TRY(hipEventSynchronize(ev));
  return FCP_OK;
}

// (callers hold pool->cal_mu)
// Only the STREAMS go: the PrivateLane objects and their events live as long as the process, so a request that raced
// with a re-creation (a plan using the lanes unverified, a descriptor slot querying a lane's completion event it
// remembered) finds a lane without a stream — and stays on its caller's stream — or a valid, long-completed event,
// never freed memory.
void destroy_lanes(LanePool *pool) {
  g_lane_generation.fetch_add(1, std::memory_order_acq_rel);
  for (auto &l : pool->lanes) {
    std::lock_guard<std::mutex> lane_lock(l->mu);
    if (l->stream) (void)hipStreamDestroy(l->stream);
    l->stream = nullptr;
  }
}

// makes the pool's first n lanes usable (objects appended, streams created where a lane has none); prio: 0 = the
// caller's (normal), 1 = lowest, 2 = highest
int create_lanes(LanePool *pool, int n, int prio) {
  int least = 0, greatest = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  if (n > kMaxPoolLanes) n = kMaxPoolLanes; // (the vector never reallocates: requests index it without the pool's mutex)
  while ((int)pool->lanes.size() < n) {
    std::unique_ptr<PrivateLane> l(new PrivateLane());
    for (int i = 0; i < kLaneEvents; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&l->in[i], hipEventDisableTiming | hipEventDisableSystemFence));
      HIP_TRY(hipEventCreateWithFlags(&l->out[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    pool->lanes.push_back(std::move(l));
  }
  for (int i = 0; i < n; ++i) {
    PrivateLane &L = *pool->lanes[i];
    std::lock_guard<std::mutex> lane_lock(L.mu);
    if (L.stream) continue;
    if (prio != 0 && least != greatest) {
      HIP_TRY(hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, prio == 1 ? least : greatest));
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    }
  }
  return FCP_OK;
}
} // namespace fcph

extern "C" {

// Plan-owned private streams.  TensorFlow gives a GPU op ONE compute stream (feature_column_process_op_gpu.cu.cc:65-131
// takes it from the op context; the reference harness' serve workers share one Session, recom_examples.patch:193-216), so
// behind the op surface consecutive requests serialise: every launch pays its own kernel boundary, its dependent front
// and its drain (S2: ~28.5 us isolated against ~23 us when neighbours cover them).  With n lanes the request
//   1. records an event on the CALLER's stream (everything enqueued there so far: the blob's producer, the previous user
//      of the arena memory the allocator hands out — TF's allocator reuses memory in compute-stream order),
//   2. runs on lane k = round robin, which waits for that event,
//   3. records the lane's `out` event and files it under the arena's address range;
// the consumer (Addons>ConcatOutputs, or any reader of the arena) calls fcp_result_wait(buffer, its stream) before it
// enqueues work that reads the result.  Lifetime: blob, tables and arena are `tensor_buffers` inputs of ConcatOutputs in
// the rewritten graph (cuda_emitter.cc:2632-2643), i.e. alive until the consumer has been enqueued behind that wait.
int fcp_plan_set_private_streams(fcp_plan_t *p, int32_t n_streams, uint32_t flags) {
  if (!p || n_streams < 0 || n_streams > 16) return fail(FCP_ERR_INVALID_ARGUMENT, "private streams: 0..16");
  if (flags & ~(uint32_t)(FCP_PRIVATE_NO_CALLER_WAIT | FCP_PRIVATE_ALWAYS | FCP_PRIVATE_NO_VERIFY)) return fail(FCP_ERR_INVALID_ARGUMENT, "unknown private-stream flags");
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan");
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(p->mu);
  LanePool *pool = p->pool ? p->pool : lane_pool_for(p->desc.device);
  std::lock_guard<std::mutex> cal_lock(pool->cal_mu);
  if (p->lane_count > 0)
    for (auto &l : pool->lanes)
      if (l->stream) HIP_TRY(hipStreamSynchronize(l->stream)); // this plan's results are complete from here on
  pending_forget(p);
  p->lane_verdicts.clear();
  p->lane_good_caller.store(nullptr, std::memory_order_release);
  if (p->lane_relies.exchange(false)) pool->n_relying.fetch_sub(1, std::memory_order_acq_rel);
  {
    std::lock_guard<std::mutex> sup_lock(p->sup.mu);
    sup_reset(p->sup, p->env);
  }
  p->lane_demoted.store(false, std::memory_order_release);
  {
    // test aid (FCP_DIAG=lane_fault_us=N), see LanePool::fault_us
    pool->fault_us.store((int)std::max<long long>(fcp::diag_ll("lane_fault_us", 0), 0), std::memory_order_relaxed);
    pool->last_out.store(nullptr, std::memory_order_relaxed);
  }
  p->lane_flags = flags;
  {
    const int64_t e = p->env.private_min_work_bytes; // FCP_PRIVATE_MIN_WORK_BYTES
    p->lane_min_work = (flags & FCP_PRIVATE_ALWAYS) ? 0 : (e >= 0 ? e : (int64_t)48 << 20);
  }
  // More than three lanes are not used: with four or more event-linked queues in flight every request took 35-100 us
  // (one stream: 28.5) under every queue count, priority and mapping tried (profiles/r04_private_streams_queue_mapping.txt);
  // independent streams do not show it (2..8 serve workers: 23-25 us).  The request stays accepted — the round robin
  // simply runs over three.
  constexpr int kMaxLanes = 3;
  if (n_streams > kMaxLanes && !fcp::diag_on("private_lanes_uncapped")) n_streams = kMaxLanes;
  p->lane_count = n_streams;
  p->pool = n_streams > 0 ? pool : nullptr;
  if (n_streams == 0) return FCP_OK;
  // The device's pool holds three lanes (more only for experiments); a plan that asks for fewer uses the first ones.
  // FCP_DIAG=lane_priority=normal|low|high: the priority lanes are CREATED with (normal: a mapping that does not overlap then
  // costs 29-44 us per S2 request; with another priority 74-87 us).  Verification (below) moves on to the others.
  int prio = 0;
  if (const char *e = fcp::diag("lane_priority")) prio = !std::strcmp(e, "low") ? 1 : !std::strcmp(e, "high") ? 2 : 0;
  return create_lanes(pool, std::max(n_streams, kMaxLanes), prio);
}

// ---- diagnostic: do the plan's private streams overlap behind THIS caller stream, in THIS process? ----------------------
// Whether event-linked streams overlap depends on which hardware queues the runtime gave them (creation order of every
// stream of the process, GPU_MAX_HW_QUEUES, priorities): the same three lanes measured 24.5 us per S2 request or 40-85 us
// (one stream: 28.6) with nothing changed but the number of streams the process had created before
// (profiles/r04_private_streams_queue_mapping.txt).  The probe replays the request pattern with kernels that only wait:
// `requests` kernels of `spin_us` microseconds (grid_blocks x 256 threads), each followed — lanes - 1 requests later — by
// its consumer (fcp_result_wait's stream wait + a one-thread kernel) on the caller's stream; once back to back on the
// caller's stream (serial_us), once through the lanes (lanes_us); host clock around each, ending with a synchronisation
// of `stream`.  lanes_us well below serial_us: the lanes overlap; at or above it: they do not, or worse.
} // extern "C"

namespace fcph {
__global__ void fcp_spin_kernel(unsigned long long ticks) { // s_memrealtime: 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
__global__ void fcp_probe_consumer_kernel() {}

// (the plan's device is current; pool->cal_mu is held: the lanes are not re-created meanwhile)
int run_lane_probe(fcp_plan *p, hipStream_t caller, int requests, int spin_us, int grid_blocks, double *serial_us, double *lanes_us) {
  const unsigned long long ticks = 100ull * (unsigned long long)spin_us;
  const dim3 grid(grid_blocks), block(FCP_BLOCK_THREADS);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  HIP_TRY(hipStreamSynchronize(caller));
  auto t0 = now();
  for (int i = 0; i < requests; ++i) {
    hipLaunchKernelGGL(fcp_spin_kernel, grid, block, 0, caller, ticks);
    hipLaunchKernelGGL(fcp_probe_consumer_kernel, dim3(1), dim3(1), 0, caller);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(caller));
  if (serial_us) *serial_us = us(t0, now());
  if (!p->pool || p->lane_count == 0) {
    if (lanes_us) *lanes_us = 0.0;
    return FCP_OK;
  }
  const int nl = p->lane_count;
  std::vector<hipEvent_t> done((size_t)requests, nullptr);
  t0 = now();
  for (int i = 0; i < requests; ++i) {
    PrivateLane &L = *p->pool->lanes[i % nl];
    {
      std::lock_guard<std::mutex> lane_lock(L.mu);
      const uint32_t e = L.next++ % kLaneEvents;
      if (!(p->lane_flags & FCP_PRIVATE_NO_CALLER_WAIT)) {
        HIP_TRY(hipEventRecord(L.in[e], caller));
        HIP_TRY(hipStreamWaitEvent(L.stream, L.in[e], 0));
      }
      hipExtLaunchKernelGGL(fcp_spin_kernel, grid, block, 0, L.stream, nullptr, L.out[e], 0, ticks);
      HIP_TRY(hipGetLastError());
      done[i] = L.out[e];
    }
    const int k = i - (nl - 1);
    if (k >= 0) {
      HIP_TRY(hipStreamWaitEvent(caller, done[k], 0));
      hipLaunchKernelGGL(fcp_probe_consumer_kernel, dim3(1), dim3(1), 0, caller);
    }
  }
  for (int k = std::max(requests - (nl - 1), 0); k < requests; ++k) {
    HIP_TRY(hipStreamWaitEvent(caller, done[k], 0));
    hipLaunchKernelGGL(fcp_probe_consumer_kernel, dim3(1), dim3(1), 0, caller);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(caller));
  if (lanes_us) *lanes_us = us(t0, now());
  return FCP_OK;
}

// The first lane-eligible request of a caller stream (or fcp_plan_verify_private_streams, at warm-up): do the lanes overlap
// behind it?  While no caller has been found good, other mappings are tried: the lanes re-created with the next priority,
// then behind one more spacer stream (a stream that has run one empty kernel holds a hardware queue and shifts everything
// created after it), up to kMaxSpacers — and, whatever is left to try, until `budget_ms` of wall time are spent (a mapping
// costs ~8 ms: the search never holds a request for more than the budget plus one probe).  The probe: 24 one-block kernels
// of 80 us with their consumers, three times (the first pass brings the queues up, the better of the other two counts);
// serial / lanes >= 2.15 (three lanes) counts as overlap (scripts/probes/lane_probe_vs_real.py, lanes_cold_start.py,
// profiles/r04_private_streams_queue_mapping.txt).  *ok = false: this caller's requests stay on its own stream.
// `again`: forget an earlier verdict of this caller and verify afresh (the warm-up entry point after a cheap first look).
int verify_lanes(fcp_plan *p, hipStream_t caller, int budget_ms, bool again, bool *ok) {
  LanePool *pool = p->pool;
  *ok = false;
  if (!pool) return FCP_OK;
  std::lock_guard<std::mutex> cal_lock(pool->cal_mu);
  for (size_t i = 0; i < p->lane_verdicts.size(); ++i)
    if (p->lane_verdicts[i].first == caller) {
      if (!again || p->lane_verdicts[i].second) {
        *ok = p->lane_verdicts[i].second;
        return FCP_OK;
      }
      p->lane_verdicts.erase(p->lane_verdicts.begin() + (long)i);
      break;
    }
  if (p->lane_count == 0 || pool->lanes.empty()) return FCP_OK;
  const bool verbose = fcp::diag_on("private_verify_verbose"); // (looked up per verification: rare)
  constexpr int kMaxSpacers = 6, kProbeSpinUs = 80;
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(std::max(budget_ms, 0));
  auto in_budget = [&] { return std::chrono::steady_clock::now() < deadline; };
  // With 40-us kernels three lanes gave 1.9-2.1 on mappings that overlap (the lanes' side is then bound by the host's five
  // runtime calls per request) and up to 1.48 on mappings that do not (S2 at 29-44 us per request) — too close: one process in
  // a dozen accepted a bad one.  80-us kernels: 2.31-2.42 where three lanes overlap, 1.8-1.97 where only two do (two of the
  // streams on one hardware queue), <= 1.5 otherwise.  Two lanes: ~1.7 against <= 1.4.  One lane never gains (its consumer
  // waits for it at once: S2 45 us per request against 28.7 on the caller's stream, ratio 0.7-1.0).
  const double kProbeGood = p->lane_count >= 3 ? 2.15 : p->lane_count == 2 ? 1.55 : 1.3;
  auto overlap = [&](double *ratio) -> int {
    double best = 0;
    for (int pass = 0; pass < 3; ++pass) { // the first pass brings the queues up; the better of the other two counts
      double a = 0, b = 0;
      const int rc = run_lane_probe(p, caller, 24, kProbeSpinUs, 1, &a, &b);
      if (rc) return rc;
      if (pass > 0 && b > 0) best = std::max(best, a / b);
    }
    *ratio = best;
    return FCP_OK;
  };
  const int n = (int)pool->lanes.size();
  int first_prio = 0;
  if (const char *e = fcp::diag("lane_priority")) first_prio = !std::strcmp(e, "low") ? 1 : !std::strcmp(e, "high") ? 2 : 0;
  double ratio = 0;
  int rc = overlap(&ratio);
  if (rc) return rc;
  bool good = ratio >= kProbeGood;
  int tried = 1;
  if (verbose) std::fprintf(stderr, "fcp private streams: caller %p, lanes as created: serial / lanes = %.2f\n", (void *)caller, ratio);
  // other mappings only while nobody — no live plan of this device — relies on the present one
  if (!good && pool->n_relying.load(std::memory_order_acquire) == 0 && pool->spacers.size() > 12) { // earlier searches' spacers
    for (hipStream_t sp : pool->spacers) (void)hipStreamDestroy(sp);
    pool->spacers.clear();
  }
  for (int spacers = 0; !good && pool->n_relying.load(std::memory_order_acquire) == 0 && spacers <= kMaxSpacers && in_budget(); ++spacers) {
    if (spacers > 0) {
      hipStream_t sp = nullptr;
      HIP_TRY(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
      pool->spacers.push_back(sp);
      hipLaunchKernelGGL(fcp_probe_consumer_kernel, dim3(1), dim3(1), 0, sp);
      HIP_TRY(hipStreamSynchronize(sp));
    }
    for (int k = (spacers == 0 ? 1 : 0); !good && k < 3 && in_budget(); ++k) { // (spacers == 0, first priority: probed above)
      const int prio = (first_prio + k) % 3;
      for (auto &l : pool->lanes)
        if (l->stream) HIP_TRY(hipStreamSynchronize(l->stream));
      destroy_lanes(pool);
      rc = create_lanes(pool, n, prio);
      if (rc) return rc;
      rc = overlap(&ratio);
      if (rc) return rc;
      good = ratio >= kProbeGood;
      ++tried;
      if (verbose)
        std::fprintf(stderr, "fcp private streams: caller %p, %d spacer(s), priority %s: serial / lanes = %.2f\n", (void *)caller,
                     (int)pool->spacers.size(), prio == 0 ? "normal" : prio == 1 ? "low" : "high", ratio);
    }
  }
  if (verbose && !good)
    std::fprintf(stderr, "fcp private streams: caller %p keeps its requests: %d mapping(s) tried within %d ms, none overlaps\n", (void *)caller, tried, budget_ms);
  p->lane_verdicts.emplace_back((void *)caller, good);
  if (good) {
    if (!p->lane_relies.exchange(true)) pool->n_relying.fetch_add(1, std::memory_order_acq_rel);
    p->lane_good_caller.store((void *)caller, std::memory_order_release);
  }
  *ok = good;
  return FCP_OK;
}

// wall-time budget of a verification that a REQUEST triggers (FCP_PRIVATE_VERIFY_BUDGET_MS; the warm-up entry point
// fcp_plan_verify_private_streams names its own)
int default_verify_budget_ms(const fcp_plan *p) {
  return p->env.private_verify_budget_ms >= 0 ? p->env.private_verify_budget_ms : 120;
}

// ---- the supervisor (struct LaneSupervisor above) -----------------------------------------------------------------------
void sup_reset(LaneSupervisor &S, const fcp::Env &env) { // (S.mu held, or no request in flight); events are kept
  S.caller = nullptr;
  S.use_lanes = true;
  S.seq = 0;
  S.next_eval = 1;
  S.phase = 0;
  S.w_lane = nullptr;
  S.strikes = 0;
  S.evaluations = S.lane_requests = 0;
  S.so_ns_per_byte = S.last_ratio = S.worst_ratio = 0;
  S.demoted = false;
  S.on = true;
  S.period = 8192;
  S.first_gap = 256;
  S.keep_ratio = 0.97;
  if (env.lane_supervise >= 0) S.on = env.lane_supervise != 0;                                                  // FCP_LANE_SUPERVISE
  if (env.lane_supervise_period >= 0) S.period = (uint32_t)std::max(env.lane_supervise_period, 4 * kSupWindow); // FCP_LANE_SUPERVISE_PERIOD
  if (env.lane_keep_ratio >= 0) S.keep_ratio = std::max(env.lane_keep_ratio, 0.1);                              // FCP_LANE_KEEP_RATIO
  S.first_gap = std::min<uint32_t>(S.first_gap, S.period);
  S.gap = S.first_gap;
}

int sup_events(LaneSupervisor &S) { // timing events, created on first use (S.mu held)
  if (S.w0) return FCP_OK;
  HIP_TRY(hipEventCreate(&S.w0));
  HIP_TRY(hipEventCreate(&S.w1));
  HIP_TRY(hipEventCreate(&S.b0));
  HIP_TRY(hipEventCreate(&S.b1));
  return FCP_OK;
}

// An evaluation whose four events have completed -> ratio -> strikes.  Returns +1: demote the caller now, -1: re-admit it
// now, 0: nothing changes.  (S.mu held)
int sup_decide(LaneSupervisor &S) {
  if (S.phase != 4 || hipEventQuery(S.b1) != hipSuccess || hipEventQuery(S.w1) != hipSuccess) return 0;
  S.phase = 0;
  S.next_eval = S.seq + S.gap;
  S.gap = std::min<uint64_t>(S.gap * 2, S.period);
  float so_ms = 0, w_ms = 0;
  if (hipEventElapsedTime(&so_ms, S.b0, S.b1) != hipSuccess || hipEventElapsedTime(&w_ms, S.w0, S.w1) != hipSuccess || so_ms <= 0 ||
      w_ms <= 0 || S.so_bytes <= 0 || S.w_bytes <= 0)
    return 0;
  ++S.evaluations;
  S.so_ns_per_byte = (double)so_ms * 1e6 / (double)S.so_bytes;
  S.last_ratio = ((double)w_ms * 1e6 / (double)S.w_bytes) / S.so_ns_per_byte;
  S.worst_ratio = std::max(S.worst_ratio, S.last_ratio);
  const bool lanes_win = S.last_ratio <= S.keep_ratio;
  S.strikes = lanes_win != S.use_lanes ? S.strikes + 1 : 0;
  if (S.strikes < 2) return 0;
  S.strikes = 0;
  S.gap = S.first_gap; // after a switch: look again soon — the NEXT evaluation too (next_eval had been computed
  S.next_eval = S.seq + S.first_gap; // from the old, possibly 8192-request gap: re-admission could take > 8 k requests)
  return lanes_win ? -1 : +1;
}

// (no lane mutex held: cal_mu is taken, and verify_lanes takes the lanes' mutexes under cal_mu)
// to_lanes false: the supervisor demotes `caller` — verdict 0, its requests stay on its stream; true: it re-admits it.
void switch_caller(fcp_plan *p, void *caller, bool to_lanes) {
  LanePool *pool = p->pool;
  if (!pool) return;
  double ratio = 0, keep = 0;
  uint64_t evals = 0;
  {
    std::lock_guard<std::mutex> sup_lock(p->sup.mu);
    if (p->sup.use_lanes == to_lanes) return;
    p->sup.use_lanes = to_lanes;
    p->sup.demoted = !to_lanes;
    ratio = p->sup.last_ratio;
    keep = p->sup.keep_ratio;
    evals = p->sup.evaluations;
  }
  std::lock_guard<std::mutex> cal_lock(pool->cal_mu);
  bool found = false;
  for (auto &v : p->lane_verdicts)
    if (v.first == caller) v.second = to_lanes, found = true;
  if (!found) p->lane_verdicts.emplace_back(caller, to_lanes);
  if (to_lanes) {
    p->lane_good_caller.store(caller, std::memory_order_release);
    if (!p->lane_relies.exchange(true)) pool->n_relying.fetch_add(1, std::memory_order_acq_rel);
  } else {
    void *expect = caller;
    p->lane_good_caller.compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel);
    p->lane_demoted.store(true, std::memory_order_release);
    bool any_good = false;
    for (auto &v : p->lane_verdicts) any_good = any_good || v.second;
    // nobody of this plan relies on the mapping any more: a later verification may search another one
    if (!any_good && p->lane_relies.exchange(false)) pool->n_relying.fetch_sub(1, std::memory_order_acq_rel);
  }
  // A library does not write to stderr on its own: the switch is queryable — fcp_plan_private_streams_stats
  // (`demoted`, `last_ratio`, `evaluations`) and fcp_plan_private_streams_verdict — and logged only under FCP_DIAG=lane_log.
  if (!fcp::diag_on("lane_log")) return;
  std::fprintf(stderr, "fcp private streams: caller stream %p %s: its requests ran at %.2fx the stream-order time per byte on the private "
                       "streams in two consecutive evaluations (kept below %.2f; %llu evaluation(s) so far)%s\n",
               caller, to_lanes ? "RE-ADMITTED to the private streams" : "DEMOTED to its own stream", ratio, keep, (unsigned long long)evals,
               to_lanes ? "" : " — readers right behind their requests, sparse traffic, or a hardware-queue mapping that no longer overlaps "
                               "(fcp_plan_verify_private_streams searches a new one)");
}
} // namespace fcph

extern "C" {

// What the verification decided for `stream`: 1 = its requests take the private streams, 0 = they stay on `stream` (nothing
// overlapped behind it, or the supervisor demoted it), -1 = no request of that stream has been verified yet, or the mode is off.
int fcp_plan_private_streams_verdict(fcp_plan_t *p, void *stream, int32_t *verdict) {
  if (!p || !verdict) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  *verdict = -1;
  LanePool *pool = p->pool;
  if (!pool || p->lane_count == 0) return FCP_OK;
  std::lock_guard<std::mutex> cal_lock(pool->cal_mu);
  for (auto &v : p->lane_verdicts)
    if (v.first == stream) *verdict = v.second ? 1 : 0;
  return FCP_OK;
}

// The verification at a time of the caller's choosing — the warm-up request every deployment of the reference runs anyway
// (docs/build_from_source.md:42) — so that no serving request pays for it.
int fcp_plan_verify_private_streams(fcp_plan_t *p, void *stream, int32_t budget_ms, int32_t *verdict) {
  if (!p) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan");
  if (verdict) *verdict = -1;
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan");
  if (!p->pool || p->lane_count == 0) return FCP_OK; // mode off: nothing to verify
  // a plan whose requests (the shapes it has seen last) are below the work threshold keeps them on the caller's stream:
  // nothing to verify, nothing to pay (a later, heavier request verifies itself within the request budget)
  const int64_t seen = p->last_work_bytes.load(std::memory_order_relaxed);
  if (seen > 0 && seen < p->lane_min_work) return FCP_OK;
  hipStream_t caller = static_cast<hipStream_t>(stream);
  if (stream_is_capturing(caller)) return fail(FCP_ERR_INVALID_ARGUMENT, "verify: the stream is being captured");
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  bool ok = false;
  rc = verify_lanes(p, caller, budget_ms > 0 ? budget_ms : 400, /*again=*/true, &ok);
  if (rc) return rc;
  if (ok) { // a new mapping, or a new look at the old one: the supervisor starts over (first evaluation at the next request)
    std::lock_guard<std::mutex> sup_lock(p->sup.mu);
    if (p->sup.demoted || p->sup.caller != stream) sup_reset(p->sup, p->env);
  }
  if (ok) { // the plan-wide "somebody is demoted" flag goes only when no caller stream is left with a negative verdict
    std::lock_guard<std::mutex> cal_lock(p->pool->cal_mu);
    bool any_demoted = false;
    for (auto &v : p->lane_verdicts) any_demoted = any_demoted || !v.second;
    if (!any_demoted) p->lane_demoted.store(false, std::memory_order_release);
  }
  if (verdict) *verdict = ok ? 1 : 0;
  return FCP_OK;
}

int fcp_plan_private_streams_stats(fcp_plan_t *p, fcp_private_streams_stats_t *out) {
  if (!p || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  std::memset(out, 0, sizeof(*out));
  std::lock_guard<std::mutex> sup_lock(p->sup.mu);
  const LaneSupervisor &S = p->sup;
  out->supervised_stream = S.caller;
  out->requests = (int64_t)S.seq;
  out->lane_requests = (int64_t)S.lane_requests;
  out->evaluations = (int64_t)S.evaluations;
  out->stream_order_us_per_mib = S.so_ns_per_byte * 1048576.0 / 1e3;
  out->last_ratio = S.last_ratio;
  out->worst_ratio = S.worst_ratio;
  out->keep_ratio = S.keep_ratio;
  out->demoted = S.demoted ? 1 : 0;
  out->evaluation_in_progress = S.phase != 0 ? 1 : 0;
  return FCP_OK;
}

int fcp_plan_probe_private_streams(fcp_plan_t *p, void *stream, int32_t requests, int32_t spin_us, int32_t grid_blocks,
                                   double *serial_us, double *lanes_us) {
  if (!p || requests < 1 || requests > 4096 || spin_us < 1 || spin_us > 10000 || grid_blocks < 1 || grid_blocks > (1 << 20))
    return fail(FCP_ERR_INVALID_ARGUMENT, "probe: bad arguments");
  if (p->host_only) return fail(FCP_ERR_NO_DEVICE, "host-only plan");
  hipStream_t caller = static_cast<hipStream_t>(stream);
  if (stream_is_capturing(caller)) return fail(FCP_ERR_INVALID_ARGUMENT, "probe: the stream is being captured");
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  LanePool *pool = p->pool ? p->pool : lane_pool_for(p->desc.device);
  std::lock_guard<std::mutex> cal_lock(pool->cal_mu);
  return run_lane_probe(p, caller, requests, spin_us, grid_blocks, serial_us, lanes_us);
}

int fcp_plan_set_request_order(fcp_plan_t *p, int32_t order) {
  if (!p || (order != FCP_ORDER_STREAM && order != FCP_ORDER_INPUTS_READY)) return fail(FCP_ERR_INVALID_ARGUMENT, "bad request order");
  p->request_order = order;
  return FCP_OK;
}

} // extern "C"

namespace fcph {
// the request on the caller's own stream although the plan has private streams (small, captured, unverified, demoted)
int process_on_caller(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  fcp_process_result_t local{};
  if (!r) r = &local;
  const int rc = fcp_internal_process(p, a, r);
  if (rc == FCP_OK && r->buffer) pending_clear_range(r->buffer, r->buffer_bytes);
  return rc;
}

// An evaluation whose window saw something it does not account for — a request that failed inside it, or one that went
// around the supervisor (below the work threshold, a stream capture) and so ran between the timing events with its time
// counted and its bytes not — is thrown away and rescheduled: no verdict from a skewed ratio.
void sup_abort(fcp_plan *p) {
  LaneSupervisor &S = p->sup;
  std::lock_guard<std::mutex> sup_lock(S.mu);
  if (S.phase >= 1 && S.phase <= 3) {
    S.phase = 0;
    S.next_eval = S.seq + S.first_gap;
  }
  p->sup_window.store(false, std::memory_order_relaxed);
}

int process_on_private_streams(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  hipStream_t caller = static_cast<hipStream_t>(a->stream);
  // small requests stay on the caller's stream; so does a capture, which records the caller's stream only (cross-stream
  // events would fork it)
  if (p->last_work_bytes.load(std::memory_order_relaxed) < p->lane_min_work || stream_is_capturing(caller)) {
    if (p->sup_window.load(std::memory_order_relaxed)) sup_abort(p); // it runs inside an open timing window, unaccounted
    return process_on_caller(p, a, r);
  }
  DeviceGuard guard;
  int rc = guard.enter(p->desc.device);
  if (rc) return rc;
  LaneSupervisor &S = p->sup;
  bool supervised = false; // the supervisor routes this caller's requests (it may have demoted it: the verdict is then its own)
  if (S.on) {
    std::lock_guard<std::mutex> sup_lock(S.mu);
    supervised = S.caller == (void *)caller;
  }
  if (!supervised) {
    if (!(p->lane_flags & FCP_PRIVATE_NO_VERIFY)) {
      if (p->lane_good_caller.load(std::memory_order_acquire) != (void *)caller) {
        bool ok = false;
        // first request of this caller: probes (and may re-create) the lanes within the budget; later: a lookup
        rc = verify_lanes(p, caller, default_verify_budget_ms(p), /*again=*/false, &ok);
        if (rc) return rc;
        if (!ok) return process_on_caller(p, a, r);
      }
    } else if (p->lane_demoted.load(std::memory_order_acquire)) { // a caller of this plan has been demoted: this one?
      bool mine = false;
      {
        std::lock_guard<std::mutex> cal_lock(p->pool->cal_mu);
        for (auto &v : p->lane_verdicts) mine = mine || (v.first == (void *)caller && !v.second);
      }
      if (mine) return process_on_caller(p, a, r);
    }
  }
  if ((p->lane_flags & FCP_PRIVATE_NO_VERIFY) && !p->lane_relies.load(std::memory_order_acquire)) {
    // first unverified use: counted under the pool's mutex, where verify_lanes of another plan reads the count before it
    // re-creates the lanes
    std::lock_guard<std::mutex> cal_lock(p->pool->cal_mu);
    if (!p->lane_relies.exchange(true)) p->pool->n_relying.fetch_add(1, std::memory_order_acq_rel);
  }
  static const bool lane_stats = fcp::diag_on("lane_stats"); // diagnostic: host time of a private-stream request by part
  auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // -- supervisor: which way does this request go? --------------------------------------------------------------------------
  bool to_lane = true, so_window = false, lane_window = false;
  int decision = 0; // +1: demote the caller after this request, -1: re-admit it
  if (S.on) {
    std::lock_guard<std::mutex> sup_lock(S.mu);
    if (!S.caller) S.caller = (void *)caller; // the first caller that got this far: verified good, or taken unverified
    if (S.caller == (void *)caller) {
      ++S.seq;
      decision = sup_decide(S);
      if (S.phase == 0 && S.seq >= S.next_eval) { // an evaluation starts with this request: stream order first
        if ((rc = sup_events(S))) return rc;
        S.phase = 1;
        S.so_count = 0;
        S.so_bytes = 0;
        p->sup_window.store(true, std::memory_order_relaxed);
        HIP_TRY(hipEventRecord(S.b0, caller));
      }
      to_lane = S.use_lanes != (decision != 0); // (a decision taken just now counts from this request on)
      if (S.phase == 1) {
        to_lane = false;
        so_window = true;
      } else if (S.phase == 2 || S.phase == 3) {
        to_lane = true;
        lane_window = true;
      }
      if (to_lane) ++S.lane_requests;
    }
  }
  if (!to_lane) {
    rc = process_on_caller(p, a, r);
    if (rc) {
      if (so_window) sup_abort(p);
      return rc;
    }
    if (so_window) {
      std::lock_guard<std::mutex> sup_lock(S.mu);
      if (S.phase == 1) {
        S.so_bytes += tl_work_bytes;
        if (++S.so_count >= kSupWindow) { // every counted request has been enqueued: the window ends behind this one
          HIP_TRY(hipEventRecord(S.b1, caller));
          S.phase = 2;
          p->sup_window.store(false, std::memory_order_relaxed);
        }
      }
    }
    if (decision) switch_caller(p, (void *)caller, decision < 0);
    return FCP_OK;
  }
  {
    PrivateLane &L = *p->pool->lanes[p->pool->rr.fetch_add(1, std::memory_order_relaxed) % (uint32_t)p->lane_count];
    const uint64_t s0 = lane_stats ? now_ns() : 0;
    std::unique_lock<std::mutex> lane_lock(L.mu);
    if (!L.stream) { // the lanes are being re-created by another plan's verification (this plan uses them unverified)
      lane_lock.unlock();
      return process_on_caller(p, a, r);
    }
    const uint32_t e = L.next++ % kLaneEvents;
    // -- supervisor: the lane window of an evaluation -----------------------------------------------------------------------
    int mark = 0; // 1: this request opens the window (w0 behind it), 2: it closes it (w1 behind it)
    bool in_window = false;
    if (lane_window) {
      std::lock_guard<std::mutex> sup_lock(S.mu);
      if (S.phase == 2) {
        S.phase = 3; // (w0 is recorded behind this request; a failure before that aborts the evaluation: sup_abort below)
        p->sup_window.store(true, std::memory_order_relaxed);
        S.w_lane = &L;
        S.w_count = 0;
        S.w_bytes = 0;
        mark = 1;
      } else if (S.phase == 3) {
        in_window = true;
        if (++S.w_count >= kSupWindow && S.w_lane == &L) mark = 2;
      }
    }
    LaneDep dep{L.in[e], caller, L.stream};
    fcp_process_args_t b = *a;
    b.stream = L.stream;
    fcp_process_result_t local{};
    if (!r) r = &local;
    const int fault = p->pool->fault_us.load(std::memory_order_relaxed);
    if (fault) {
      if (hipEvent_t prev = p->pool->last_out.load(std::memory_order_acquire)) HIP_TRY(hipStreamWaitEvent(L.stream, prev, 0));
      hipLaunchKernelGGL(fcp_spin_kernel, dim3(1), dim3(64), 0, L.stream, 100ull * (unsigned long long)fault);
    }
    // the completion event rides on the dispatch packet of the request's last kernel (no marker packet of its own);
    // a request without a kernel (nothing to compute) records it the plain way
    static const bool attach = fcp::diag_ll("lane_stop_event", 1) != 0; // tuning aid: 0 = always record a marker
    if (attach) fcp_set_stop_event(L.out[e]);
    static const bool alias_done = fcp::diag_ll("lane_done_alias", 1) != 0; // tuning aid: 0 = the slot records an event of its own
    if (alias_done) {
      tl_lane_done = L.out[e];
      tl_lane_done_gen = g_lane_generation.load(std::memory_order_acquire);
    }
    // the caller's stream is recorded, and the lane made to wait for it, inside the call: right behind malloc_buff
    if (!(p->lane_flags & FCP_PRIVATE_NO_CALLER_WAIT)) tl_lane_dep = &dep;
    rc = fcp_internal_process(p, &b, r);
    const bool dep_left = tl_lane_dep != nullptr; // (an error before the allocation)
    tl_lane_dep = nullptr;
    tl_lane_done = nullptr;
    const bool pending = fcp_stop_event_pending();
    fcp_set_stop_event(nullptr);
    if (rc || dep_left) {
      if (lane_window) sup_abort(p);
      return rc ? rc : fail(FCP_ERR_HIP, "private streams: the request never reached its allocation");
    }
    const uint64_t s3 = lane_stats ? now_ns() : 0;
    if (!attach || pending) HIP_TRY(hipEventRecord(L.out[e], L.stream));
    pending_register(p, r->buffer, r->buffer_bytes, a->concated_inputs, a->concated_bytes, L.out[e]);
    if (fault) p->pool->last_out.store(L.out[e], std::memory_order_release);
    if (mark || in_window) {
      std::lock_guard<std::mutex> sup_lock(S.mu);
      if (in_window && S.phase == 3) S.w_bytes += tl_work_bytes;
      if (mark == 1 && S.phase == 3) HIP_TRY(hipEventRecord(S.w0, L.stream));
      if (mark == 2 && S.phase == 3) {
        HIP_TRY(hipEventRecord(S.w1, L.stream));
        S.phase = 4;
        p->sup_window.store(false, std::memory_order_relaxed);
      }
    }
    if (lane_stats) {
      static std::atomic<uint64_t> n{0}, a_proc{0}, a_reg{0};
      const uint64_t s4 = now_ns();
      a_proc += s3 - s0;
      a_reg += s4 - s3;
      if ((++n & 1023) == 0)
        std::fprintf(stderr, "fcp private-stream request, host us: lane lock + request itself (incl. the record on the caller's stream and the lane's wait) %.2f, completion event + registry %.2f\n",
                     a_proc.load() / 1e3 / n.load(), a_reg.load() / 1e3 / n.load());
    }
  }
  if (decision) switch_caller(p, (void *)caller, decision < 0);
  return FCP_OK;
}

// the pending result whose address range contains x, or nullptr (g_pending_mu held)
const PendingResult *pending_find(uintptr_t x, uintptr_t *base) {
  auto it = g_pending.upper_bound(x);
  if (it == g_pending.begin()) return nullptr;
  --it;
  if (x >= it->second.end) return nullptr;
  if (base) *base = it->first;
  return &it->second;
}
} // namespace fcph

extern "C" {

// The consumer's half: `stream` waits (on the device; the host does not block) for the request whose arena contains
// `buffer`.  Nothing pending for that address — no private streams, or the request has long completed — is FCP_OK.
int fcp_result_wait(const void *buffer, void *stream) {
  if (!buffer) return fail(FCP_ERR_INVALID_ARGUMENT, "null buffer");
  std::lock_guard<std::mutex> lock(g_pending_mu); // (held over the runtime call: the entry may not be replaced meanwhile)
  if (const PendingResult *pr = pending_find(reinterpret_cast<uintptr_t>(buffer), nullptr))
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), pr->done, 0));
  return FCP_OK;
}

// The same for a HOST reader: returns when the request whose arena contains `buffer` has completed.  The registry's mutex
// is NOT held while the host waits (every other thread's request, wait or stager call would block for a kernel's duration,
// ADVICE r04): lane events live as long as the process (destroy_lanes), so the handle stays valid outside the lock; an event
// re-recorded meanwhile belongs to a LATER request of the same lane, whose completion implies this one's.
int fcp_result_synchronize(const void *buffer) {
  if (!buffer) return fail(FCP_ERR_INVALID_ARGUMENT, "null buffer");
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_pending_mu);
    const PendingResult *pr = pending_find(reinterpret_cast<uintptr_t>(buffer), nullptr);
    if (!pr || hipEventQuery(pr->done) == hipSuccess) return FCP_OK;
    ev = pr->done;
  }
  HIP_TRY(hipEventSynchronize(ev));
  return FCP_OK;
}

} // extern "C"

namespace fcph {
// ConcatOutputs reads columns of FeatureColumnProcess arenas — normally ONE arena (output_ptrs of one op), possibly several
// (per-column inputs that come from two ops; callers of the scatter variants): `stream` waits for every distinct pending
// result that contains an input.  One map lookup per arena, a range comparison per input; nothing pending: one lock.
int wait_for_inputs(const void *const *inputs, int32_t n, void *stream) {
  std::lock_guard<std::mutex> lock(g_pending_mu);
  if (g_pending.empty()) return FCP_OK;
  uintptr_t lo = 1, hi = 0; // the range found (or known to hold nothing) last
  for (int32_t k = 0; k < n; ++k) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(inputs[k]);
    if (!x || (x >= lo && x < hi)) continue;
    uintptr_t base = 0;
    if (const PendingResult *pr = pending_find(x, &base)) {
      HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), pr->done, 0));
      lo = base;
      hi = pr->end;
    }
  }
  return FCP_OK;
}
} // namespace fcph

