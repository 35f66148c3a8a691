// fcp_api.hip — host side of libfcp_hip.so: the C ABI declared in
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
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fcp_hip.h"
#include "fcp_internal.h"

namespace {

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

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(#expr, e_);   \
  } while (0)

inline int64_t align128(int64_t x) { return (x + 127) / 128 * 128; } // alignmem, cuda_emitter.cc:967-969

constexpr int kSlots = 32; // descriptor slots per plan: distinct (shapes, stream) pairs resident at once (8 until round 4: 16 rotating shapes reinstalled on every request)
constexpr uint32_t kFlagHostOnly = FCP_FLAG_HOST_ONLY; // plan without device resources (layout queries)

struct HostColumn {
  fcp_column_desc_t d;
  fcp_column_ext_t ext = {}; // extensions (segment-id map); all zeros = none
  std::vector<float> boundaries;
  std::vector<int64_t> xf_lo, xf_hi; // id transform intervals (closed)
  int64_t xf_const_off = -1;         // byte offset in the const buffer of intervals 1.. as (lo, hi) pairs
  int32_t out_off = 0;
  int64_t const_off = -1; // byte offset of the boundaries in the const buffer
};

const int64_t kSegSearchMaxPairs = [] { // tuning aid: FCP_SEG_SEARCH_MAX_PAIRS
  const char *e = std::getenv("FCP_SEG_SEARCH_MAX_PAIRS");
  return e ? std::atoll(e) : (int64_t)32768;
}();
constexpr int kAllSlotsBusy = -2; // find_or_reserve: every slot is pinned by a concurrent request
constexpr int kNeedsInstall = -3; // find_or_reserve during stream capture: these shapes are not resident

struct DynMeta {
  std::vector<int32_t> group_rows;
  std::vector<int64_t> group_base;
  int64_t arena_bytes = 0;
  int64_t csr_arena_off = 0;
  int32_t max_seg_nnz = 0;
  int64_t seg_pairs = 0;   // sum of rows over the segment-id columns: cost of the in-block search
  bool seg_search = false; // this request: blocks search the segment ids (no pre-pass launch)
  int64_t work_bytes = 0;  // table rows gathered + output written: what decides whether a private lane pays (fcp_plan_set_private_streams)
  // regular CSR (FcpLaunch::csr_reg): 0 = none, 1 = the arena scratch laid out by column position, 2 = CSR inputs that lie
  // one stride apart in the blob; byte offset of position 0's array in the arena / the blob; stride in int32 elements
  int32_t csr_reg_mode = 0, csr_reg_stride = 0;
  int64_t csr_reg_base = 0;
  // launch geometry per kernel kind: [0] dense kernel (spans whose columns all have exactly
  // one source row per output row), [1] ragged kernel (spans with pooled / scatter / reduction columns)
  struct Geo {
    int32_t grid_blocks = 0;
    int32_t rows_per_wave = 1;
    FcpGroupLaunch groups[FCP_MAX_GROUPS];
  } geo[2];
};

struct DynSlot {
  bool valid = false;
  std::vector<int32_t> key;
  FcpColDyn *h_dyn = nullptr; // pinned, mapped (host copy; source of the upload-kernel path)
  void *h_dyn_dev = nullptr;  // device-side address of h_dyn
  FcpColDyn *d_dyn = nullptr; // what the kernels read
  hipEvent_t uploaded = nullptr; // upload-kernel path: recorded after the upload
  uint64_t done_gen = 0;         // != 0: `done` is a private lane's completion event of that lane generation (see g_lane_generation)
  hipEvent_t done = nullptr;     // recorded after the first kernel that used this content; one event per
                                 // (slot, stream) — re-recording an event last used on ANOTHER stream costs
                                 // ~20 us under concurrency (HIP serialises it), on its own stream < 1 us
  std::vector<std::pair<void *, hipEvent_t>> done_pool;
  bool done_valid = false;       // `done` covers every kernel that has used this content so far
  void *stream = nullptr;        // stream of the request that installed this content
  uint64_t tick = 0;
  int users = 0;                 // requests between "slot chosen" and "kernels enqueued": not evictable
  int uses = 0;                  // requests that have used this content since it was installed
  bool captured = false;         // a stream capture recorded a launch that reads this slot: a graph replay will read
                                 // it at any later time, so it is never evicted (fcp_plan_release_captures)
  bool was_valid = false;        // reserved for installation: the previous content had readers to wait for
  DynMeta meta;
};

// One private stream of a plan (fcp_plan_set_private_streams).  A request that takes the lane holds `mu` while it
// enqueues (record on the caller's stream, wait on the lane, kernels, record on the lane), so an event pair is never
// interleaved between two host threads.  Events rotate: re-recording an event that an older consumer has not waited
// for yet makes that consumer wait for LATER work of the same lane, which covers the older request (same stream).
constexpr int kLaneEvents = 8;
constexpr int kMaxPoolLanes = 16; // (more than three only for experiments: FCP_PRIVATE_LANES_UNCAPPED)
struct PrivateLane {
  hipStream_t stream = nullptr;
  std::mutex mu;
  hipEvent_t in[kLaneEvents] = {};  // recorded on the caller's stream: everything the request may depend on
  hipEvent_t out[kLaneEvents] = {}; // recorded on the lane behind the request's last kernel
  uint32_t next = 0;
};

// The private lanes of ONE DEVICE, shared by every plan on it (created on first use, kept for the life of the process).
// The command processor overlaps at most four event-linked queues — a caller's stream and three lanes; with a fifth every
// request costs a multiple (profiles/r04_private_streams_queue_mapping.txt).  Lanes per plan would multiply them: two
// models served by one process, three lanes each, fall off that cliff as soon as both are busy.
// Lanes are destroyed and re-created only while no live plan relies on them (verify_lanes), i.e. after every plan that used
// them has synchronised them and let go.  A descriptor slot whose `done` event is a lane's completion event remembers the
// generation it was taken from: a different generation now means those lanes — and every kernel that ran on them — are gone.
std::atomic<uint64_t> g_lane_generation{1};
// the completion event of the private-stream request this thread is issuing (fcp_process_feature_columns -> fcp_internal_process)
thread_local hipEvent_t tl_lane_done = nullptr;
thread_local uint64_t tl_lane_done_gen = 0;

// What a private-stream request waits for on the caller's stream is known only once the allocator has answered: TF's
// allocator hands out memory in compute-stream order, and between this call's entry and its malloc_buff another
// Session::Run thread may have queued a kernel that still reads the very memory the arena gets.  So
// fcp_process_feature_columns leaves the dependency here and fcp_internal_process enqueues it — event record on the caller's
// stream, wait on the lane — right behind malloc_buff, in front of the first command that touches the arena.
struct LaneDep {
  hipEvent_t in;
  hipStream_t caller, lane;
};
thread_local const LaneDep *tl_lane_dep = nullptr;
// gathered + written bytes of the request this thread processed last (DynMeta::work_bytes; the supervisor's unit of work)
thread_local int64_t tl_work_bytes = 0;

// Supervision of one plan's lane traffic behind ONE caller stream (the first that took the lanes; TensorFlow has one): an
// online A/B of the two ways a request can run, under the caller's REAL traffic — its consumers, its host threads, its pace.
//   evaluation: kSupWindow consecutive requests run on the CALLER's stream between two timing events there (time per byte of
//               work in stream order), the next kSupWindow on the private streams between two timing events on one of them
//               (the second behind the first request >= kSupWindow later that lands on the same lane); ratio = lanes / stream
//               order.  No idle / busy heuristics: a caller that issues sparsely measures ~1.0 in both (nothing to overlap:
//               the lanes buy nothing), readers right behind their requests measure > 1 (the events are pure cost), lanes
//               that overlap measure 0.64-0.92, a hardware-queue mapping that stopped overlapping 1.1 and more
//               (profiles/r05_caller_threads_grid.txt).
//   decision:   two consecutive evaluations with ratio > keep_ratio (0.97): the caller is DEMOTED — verdict 0, one line on
//               stderr, its requests stay on its stream; two consecutive ones below it re-admit a demoted caller (traffic
//               changes: a trickle at start-up, load later).  One evaluation alone never switches.
//   schedule:   the first at the caller's first eligible request, the next after 256 requests, then the gap doubles up to
//               `period` (8192): one evaluation costs kSupWindow requests in the mode that loses, < 1 % of the traffic.
// Cost outside evaluations: one mutex and a counter per request.
constexpr int kSupWindow = 48;
struct LaneSupervisor {
  std::mutex mu;
  bool on = true;
  uint32_t period = 8192, first_gap = 256;
  double keep_ratio = 0.97;
  void *caller = nullptr;
  bool use_lanes = true;        // the mode requests run in between evaluations
  uint64_t seq = 0;             // eligible requests of the supervised caller so far
  uint64_t next_eval = 1, gap = 256;
  // phase of the running evaluation: 0 none, 1 stream-order window open, 2 lane window to open, 3 lane window open,
  // 4 both closed (events pending)
  int phase = 0;
  hipEvent_t b0 = nullptr, b1 = nullptr; // on the caller's stream
  hipEvent_t w0 = nullptr, w1 = nullptr; // on one lane
  PrivateLane *w_lane = nullptr;
  int so_count = 0, w_count = 0;
  int64_t so_bytes = 0, w_bytes = 0;
  int strikes = 0;              // consecutive evaluations that contradict the present mode
  uint64_t evaluations = 0, lane_requests = 0;
  double so_ns_per_byte = 0, last_ratio = 0, worst_ratio = 0;
  bool demoted = false;         // == !use_lanes after at least one decision
};

struct LanePool {
  std::mutex cal_mu;                                  // lanes / spacers are created, probed and re-created under it
  std::vector<std::unique_ptr<PrivateLane>> lanes;
  std::vector<hipStream_t> spacers;                   // streams that only hold hardware queues (verify_lanes)
  std::atomic<int> n_relying{0};                      // live plans that found the present mapping good, or use it unverified
  std::atomic<uint32_t> rr{0};
  // Test aid (FCP_LANE_FAULT_US=N, read by fcp_plan_set_private_streams): every lane request first waits for the device's
  // previous lane request and then spins for N us on its lane — lanes that serialise and stall, the signature of a
  // hardware-queue mapping that does not overlap (profiles/r04_private_streams_queue_mapping.txt: 29-86 us per S2
  // request), made deterministic for the supervisor's test.
  std::atomic<int> fault_us{0};
  std::atomic<hipEvent_t> last_out{nullptr};
};
void sup_reset(LaneSupervisor &S);
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

// Results whose kernels run on a private lane, by arena address range: what fcp_result_wait looks up.  An entry is
// replaced when its address range is handed out again and dropped once its event has completed (nothing to wait for).
struct PendingResult {
  uintptr_t end;
  hipEvent_t done;
  const void *owner; // the plan whose lane owns `done`
};
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
bool stager_reader_wait_off() { // test aid: FCP_STAGER_NO_READER_WAIT=1 reproduces the hazard the two functions below close
  static const bool off = std::getenv("FCP_STAGER_NO_READER_WAIT") != nullptr;
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

} // namespace

// failure reporting for the library's other translation units (fcp_shard.hip)
int fcp_internal_fail(int code, const std::string &msg) { return fail(code, msg); }

struct fcp_plan {
  fcp_plan_desc_t desc;
  std::vector<HostColumn> cols;
  std::vector<int32_t> ranks, elem_sizes, shape_off;
  std::vector<int32_t> group_width, group_nslots, group_map_off;
  std::vector<int32_t> seg_cols;
  int32_t n_seg_plain = 0;  // seg_cols[0 .. n_seg_plain): pooled columns; the rest: any-order ScatterNd columns (inverse maps)
  bool seg_search = false;  // blocks search the segment ids themselves; no segment-offset pre-pass
  bool has_inverse = false; // some ScatterNd column brings its row ids as delivered (any order): inverse map in the pre-pass
  // (r6) the CSR scratch of segment-id columns is laid out by column POSITION (one row of round32(rows + 1) entries per
  // column of the plan, pooled or not) instead of packed: the ragged body then knows where a range is before it has the
  // column's record (FcpLaunch::csr_reg).  One concat group, no any-order ScatterNd column (their inverse maps are the
  // scratch's tail, cleared per request), and pooled columns at least half of the plan (the unused rows cost arena bytes,
  // never traffic: RAGGED 512 of 512 columns; the reference's models E / F, 10-20 of ~1000, keep the packed scratch).
  bool csr_by_pos = false;
  // device arrays are kept in concat order (group-major, ascending concat offset)
  // so that the columns of one output span are contiguous: order[pos] = column,
  // pos_of[column] = pos.
  std::vector<int32_t> order, pos_of;
  // compact per-column facts in concat (pos) order + one representative column per group:
  // what compute_dyn_fast walks on the request path
  struct FastCol {
    int32_t ids_input, seg_input, rows_arg, dim, seg_stride;
    int64_t out_off_bytes;
    uint8_t form, rows_source, seg_kind, group;
  };
  std::vector<FastCol> fast_cols;
  std::vector<int32_t> group_rep;
  int vec = 1;
  bool dense_only = true;   // no span needs the ragged kernel
  // hybrid dispatch: per group, the spans served by the dense kernel and by the ragged kernel
  std::vector<uint32_t> span_list;                 // host copy of d_span_list
  std::vector<int32_t> list_off[2], list_n[2];     // [kind][group]
  uint32_t *d_span_list = nullptr;
  bool host_only = false;
  int32_t rank_sum = 0;

  uint32_t *d_slot_map = nullptr;
  FcpColStatic *d_cols = nullptr;
  FcpXform *d_xforms = nullptr; // per column, only for plans with id transforms
  FcpSegMap *d_segmaps = nullptr; // per column, only for plans with segment-id maps
  bool has_seg_map = false;
  bool wide_rows = false; // some table shard has >= 2^32 - 3 slots: FcpLaunch::store_through bit 1
  std::vector<FcpColStatic> h_cols;
  char *d_const = nullptr;
  int32_t *d_seg_cols = nullptr;
  unsigned long long *d_bad = nullptr;
  float *d_zeros = nullptr;     // 256 zero bytes: the row a skipped id of a bag reads (ld_slot_or_zero)
  unsigned long long *d_stamps = nullptr; // diagnostic builds only (-DFCP_STAMPS)
  std::vector<const void *> bound_tables;
  bool tables_bound = false;

  // How new shape-dependent descriptors reach the device (dynamic shapes: every
  // request).  With a large PCIe BAR the descriptor ring lives in fine-grained
  // device memory and the HOST writes it directly (posted writes, ~0.3 us for 24 KiB,
  // nothing on the GPU's critical path); otherwise a small kernel on the request's
  // stream copies it from pinned host memory.  FCP_DYN_UPLOAD=kernel forces the latter.
  bool host_writes_dyn = false;
  std::mutex mu;
  DynSlot slots[kSlots];
  uint64_t tick = 0;

  // Plan-owned private streams (fcp_plan_set_private_streams): requests of ONE caller stream run on a rotating set of
  // lanes so that consecutive requests overlap on the GPU (the front of one launch under the memory phase of another).
  LanePool *pool = nullptr;                             // the device's lanes; the plan uses the first lane_count of them
  uint32_t lane_flags = 0;
  // Verification (verify_lanes): whether event-linked streams overlap depends on the hardware queues the runtime mapped
  // them to, which no API shows.  The first request of every caller stream runs a synthetic probe of the request pattern;
  // while no caller has been found good, other mappings are tried (lanes re-created with another priority, behind
  // `spacers` — streams that only hold hardware queues); a caller behind which no mapping overlaps keeps its requests.
  std::vector<std::pair<void *, bool>> lane_verdicts;   // caller stream -> its requests may take the lanes (under pool->cal_mu)
  std::atomic<void *> lane_good_caller{nullptr};        // the last caller found good: the request path's shortcut
  int32_t lane_count = 0;                               // lanes asked for (<= kMaxLanes); 0: private streams off
  std::atomic<bool> lane_relies{false};                 // counted in pool->n_relying
  // The cross-stream events of a lane cost the host ~8 us per request and the GPU's command processor a few packets:
  // a request pays for them only when its kernel is long enough to have something to overlap.  The plan remembers
  // the work of the shapes it installed last (gathered rows + output bytes); lighter requests stay on the caller's stream.
  std::atomic<int64_t> last_work_bytes{0};
  int64_t lane_min_work = 0;
  int32_t request_order = FCP_ORDER_STREAM; // fcp_plan_set_request_order
  std::atomic<uintptr_t> recent_arena[2] = {}; // the arenas of the last two requests (store_policy_for)
  // Run-time supervision of the lanes (LaneSupervisor below): a verdict is learnt once, a mapping can go bad later (another
  // library of the process creates streams; the runtime re-maps queues): sampled windows of lane requests are timed
  // against the stream-order rate of the same requests and the caller is demoted to its own stream when they lose.
  std::atomic<bool> lane_demoted{false};                // some caller of this plan has been demoted (NO_VERIFY plans look it up)
  LaneSupervisor sup;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  int enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) return hip_fail("hipGetDevice", e);
    if (prev != dev) {
      e = hipSetDevice(dev);
      if (e != hipSuccess) return hip_fail("hipSetDevice", e);
      changed = true;
    }
    return FCP_OK;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

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
  m->seg_search = p->seg_search && m->seg_pairs <= kSegSearchMaxPairs;
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
    if (p->seg_search && m->seg_pairs <= kSegSearchMaxPairs) return; // (m->seg_search is decided from the same two facts)
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
      static const int forced = [] { // tuning aid: FCP_ROWS_PER_WAVE=1|2|4
        const char *e = std::getenv("FCP_ROWS_PER_WAVE");
        const int v = e ? std::atoi(e) : 0;
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
      static const bool no_xcd_map = std::getenv("FCP_NO_XCD_MAP") != nullptr; // tuning aid: plain span order
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
  m->seg_search = p->seg_search && seg_pairs <= kSegSearchMaxPairs;
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
  const bool slow_only = std::getenv("FCP_DYN_GENERAL") != nullptr; // test aid (read per call: only on a descriptor miss)
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
    const char *mode = std::getenv("FCP_DYN_UPLOAD");
    p->host_writes_dyn = large_bar != 0 && !(mode && std::string(mode) == "kernel");
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

// Bind the table addresses (FeatureColumnProcess `inputs`).  TF variables keep
// their address between requests, so this uploads once.
bool stream_is_capturing(hipStream_t stream) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess) return st == hipStreamCaptureStatusActive;
  (void)hipGetLastError();
  return false;
}

// `capturing`: the caller's stream is being captured into a HIP graph — binding (or re-binding) tables copies records
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
  static const bool stats = std::getenv("FCP_INSTALL_STATS") != nullptr; // diagnostic: where a descriptor installation spends its host time
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

struct SlotUnpin { // every exit path of a request: publish what `done` covers and give the slot back
  fcp_plan *p;
  DynSlot *s;
  bool recorded;
  ~SlotUnpin() {
    std::lock_guard<std::mutex> lock(p->mu);
    if (recorded) s->done_valid = s->uses == 1; // hits that joined meanwhile are not covered by `done`
    --s->users;
  }
};

// Which cache policy the output stores of a request take (FcpLaunch::store_through bits 0 and 2; st_out in fcp_kernels.hip):
//   * the arena is the one this plan's previous request wrote, or the one before it (TF's allocate_output hands a serving
//     loop the block it just freed, feature_column_process_op_gpu.cu.cc:107-111): PLAIN stores — the lines are still in the
//     Infinity Cache / L2 and rewriting a resident line beats streaming it (S2, one arena: 27.2 us against 27.9 nt / 28.0
//     sc1 nt; two arenas 28.0 / 28.2 / 28.2; RAGGED 27.25 / 27.4 / 27.55);
//   * any other arena (a ring of three or more, fresh memory): plain stores LOSE there (S2 31.1 us against 28.1) —
//     write-through `sc1 nt` once the outputs exceed what the eight 4-MiB L2s hold, `nt` below
//   (profiles/r06_arena_reuse_store_policy.txt).  A performance hint only: read and updated without the plan's mutex.
int store_policy_for(fcp_plan *p, const void *arena, int64_t out_bytes) {
  static const int64_t through_bytes = [] {
    const char *e = std::getenv("FCP_STORE_THROUGH_BYTES"); // tuning aid
    return e ? std::atoll(e) : (int64_t)32 << 20;
  }();
  static const int reuse_mode = [] { // tuning aid: 0 = never plain stores, 2 = always, default 1 = for a reused arena
    const char *e = std::getenv("FCP_STORE_PLAIN_REUSE");
    return e ? std::atoi(e) : 1;
  }();
  const uintptr_t ar = reinterpret_cast<uintptr_t>(arena);
  const uintptr_t a0 = p->recent_arena[0].load(std::memory_order_relaxed), a1 = p->recent_arena[1].load(std::memory_order_relaxed);
  const bool reused = ar == a0 || ar == a1;
  if (ar != a0) {
    p->recent_arena[1].store(a0, std::memory_order_relaxed);
    p->recent_arena[0].store(ar, std::memory_order_relaxed);
  }
  // ... while the output is of a size the caches can still hold on to: S2 at batch 512 / 640 / 768 / 1024 (61-123 MB of
  // output) gains 1.4-3.0 us per request from plain stores into its one arena, batch 2048 (246 MB) LOSES 6 of 103 us
  constexpr int64_t kPlainMaxBytes = (int64_t)160 << 20;
  if (reuse_mode == 2 || (reuse_mode == 1 && reused && out_bytes <= kPlainMaxBytes)) return 4;
  return out_bytes >= through_bytes ? 1 : 0;
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

} // namespace

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

// ---- Addons>ConcatInputs, concat_inputs_ops.cc:42-77 -------------------------
int fcp_concat_inputs_sizes(const fcp_host_tensor_t *inputs, int32_t n, int64_t *blob_bytes,
                            int32_t *rank_sum) {
  if (n < 0 || (n > 0 && !inputs)) return fail(FCP_ERR_INVALID_ARGUMENT, "null inputs");
  int64_t bytes = 0;
  int32_t ranks = 0;
  for (int32_t i = 0; i < n; ++i) {
    const fcp_host_tensor_t &t = inputs[i];
    if (t.rank < 0 || t.elem_size <= 0 || (t.rank > 0 && !t.dims))
      return fail(FCP_ERR_INVALID_ARGUMENT, "bad host tensor");
    int64_t ne = 1;
    for (int32_t j = 0; j < t.rank; ++j) {
      if (t.dims[j] < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "negative dimension");
      ne *= t.dims[j];
    }
    bytes += ne * t.elem_size;
    ranks += t.rank;
  }
  if (blob_bytes) *blob_bytes = bytes;
  if (rank_sum) *rank_sum = ranks;
  return FCP_OK;
}

int fcp_concat_inputs(const fcp_host_tensor_t *inputs, int32_t n, void *blob, int64_t blob_capacity,
                      int32_t *offsets, int32_t *shapes) {
  int64_t need = 0;
  int rc = fcp_concat_inputs_sizes(inputs, n, &need, nullptr);
  if (rc) return rc;
  if (n > 0 && (!offsets || !shapes)) return fail(FCP_ERR_INVALID_ARGUMENT, "null outputs");
  if (need > blob_capacity || (need > 0 && !blob)) return fail(FCP_ERR_INVALID_ARGUMENT, "blob too small");
  // The reference keeps offsets in int32 (:52-60); refuse what it would overflow.
  if (need > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "blob larger than 2^31 bytes (int32 offsets)");
  char *itr = static_cast<char *>(blob);
  int64_t size = 0;
  int32_t *shape_itr = shapes;
  for (int32_t i = 0; i < n; ++i) {
    const fcp_host_tensor_t &t = inputs[i];
    int64_t ne = 1;
    for (int32_t j = 0; j < t.rank; ++j) {
      ne *= t.dims[j];
      *(shape_itr++) = (int32_t)t.dims[j];
    }
    const int64_t nb = ne * t.elem_size;
    offsets[i] = (int32_t)size;
    if (nb) {
      if (!t.data) return fail(FCP_ERR_INVALID_ARGUMENT, "null tensor data");
      std::memcpy(itr, t.data, (size_t)nb);
    }
    itr += nb;
    size += nb;
  }
  return FCP_OK;
}

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
  p->seg_search = desc->shard_world <= 1 && std::getenv("FCP_SEG_PREPASS") == nullptr;
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
  if (const char *e = std::getenv("FCP_CSR_BY_POS")) // tuning aid: 0 = packed scratch (the round-5 layout)
    if (std::atoi(e) == 0) p->csr_by_pos = false;
  // The kernels park a table row as one 32-bit number (the three largest values are their sentinels): a table — or
  // one shard of it — may hold up to 2^32 - 3 ROWS, of any width: the byte offset is formed in 64 bits where the row
  // is read.  The dense body keeps round 2's pre-scaled 32-bit slot offsets (row * dim / vec) while every table of
  // the plan stays below 2^32 - 3 slots (64 GB at vec 4), which saves it a 64-bit multiply per read.  (The
  // reference's int arithmetic stops at 2^31 elements = 8 GB, cuda_emitter.cc:270-271.)
  for (int k = 0; k < desc->n_columns; ++k) {
    const fcp_column_desc_t &c = p->cols[k].d;
    if (c.form == FCP_FORM_PASSTHROUGH || c.form == FCP_FORM_BATCH_COL_REDUCTION || c.form == FCP_FORM_EXTERNAL) continue;
    const int64_t local_vocab = (c.vocab - desc->shard_rank + desc->shard_world - 1) / desc->shard_world;
    if (local_vocab >= 0xFFFFFFFDLL) {
      delete p;
      return fail(FCP_ERR_UNSUPPORTED, "column " + std::to_string(k) + ": table shard exceeds 2^32 - 3 rows");
    }
    if (local_vocab * (c.dim / p->vec) >= 0xFFFFFFFDLL || std::getenv("FCP_WIDE_ROWS") != nullptr) p->wide_rows = true; // (env: test aid)
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

namespace {
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
} // namespace

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
  static const bool done_on_kernel = [] {
    const char *v = std::getenv("FCP_DONE_STOP_EVENT"); // tuning aid: 0 = always record
    return !v || std::atoi(v) != 0;
  }();
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
    sup_reset(p->sup);
  }
  p->lane_demoted.store(false, std::memory_order_release);
  {
    const char *e = std::getenv("FCP_LANE_FAULT_US"); // test aid, see LanePool::fault_us
    pool->fault_us.store(e ? std::max(std::atoi(e), 0) : 0, std::memory_order_relaxed);
    pool->last_out.store(nullptr, std::memory_order_relaxed);
  }
  p->lane_flags = flags;
  {
    const char *e = std::getenv("FCP_PRIVATE_MIN_WORK_BYTES"); // tuning aid
    p->lane_min_work = (flags & FCP_PRIVATE_ALWAYS) ? 0 : (e ? std::atoll(e) : (int64_t)48 << 20);
  }
  // More than three lanes are not used: with four or more event-linked queues in flight every request took 35-100 us
  // (one stream: 28.5) under every queue count, priority and mapping tried (profiles/r04_private_streams_queue_mapping.txt);
  // independent streams do not show it (2..8 serve workers: 23-25 us).  The request stays accepted — the round robin
  // simply runs over three.
  constexpr int kMaxLanes = 3;
  if (n_streams > kMaxLanes && !std::getenv("FCP_PRIVATE_LANES_UNCAPPED")) n_streams = kMaxLanes;
  p->lane_count = n_streams;
  p->pool = n_streams > 0 ? pool : nullptr;
  if (n_streams == 0) return FCP_OK;
  // The device's pool holds three lanes (more only for experiments); a plan that asks for fewer uses the first ones.
  // FCP_LANE_PRIORITY=normal|low|high: the priority lanes are CREATED with (normal: a mapping that does not overlap then
  // costs 29-44 us per S2 request; with another priority 74-87 us).  Verification (below) moves on to the others.
  int prio = 0;
  if (const char *e = std::getenv("FCP_LANE_PRIORITY")) prio = !std::strcmp(e, "low") ? 1 : !std::strcmp(e, "high") ? 2 : 0;
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
namespace {
__global__ void fcp_spin_kernel(unsigned long long ticks) { // s_memrealtime: 100 MHz
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
__global__ void fcp_probe_consumer_kernel() {}
} // namespace

namespace {
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
  const bool verbose = std::getenv("FCP_PRIVATE_VERIFY_VERBOSE") != nullptr; // (read per verification: rare)
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
  if (const char *e = std::getenv("FCP_LANE_PRIORITY")) first_prio = !std::strcmp(e, "low") ? 1 : !std::strcmp(e, "high") ? 2 : 0;
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
int default_verify_budget_ms() {
  static const int ms = [] {
    const char *e = std::getenv("FCP_PRIVATE_VERIFY_BUDGET_MS");
    return e ? std::max(std::atoi(e), 0) : 120;
  }();
  return ms;
}

// ---- the supervisor (struct LaneSupervisor above) -----------------------------------------------------------------------
void sup_reset(LaneSupervisor &S) { // (S.mu held, or no request in flight); events are kept
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
  if (const char *e = std::getenv("FCP_LANE_SUPERVISE")) S.on = std::atoi(e) != 0;
  if (const char *e = std::getenv("FCP_LANE_SUPERVISE_PERIOD")) S.period = (uint32_t)std::max(std::atoi(e), 4 * kSupWindow);
  if (const char *e = std::getenv("FCP_LANE_KEEP_RATIO")) S.keep_ratio = std::max(std::atof(e), 0.1);
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
  S.gap = S.first_gap; // after a switch: look again soon
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
  std::fprintf(stderr, "fcp private streams: caller stream %p %s: its requests ran at %.2fx the stream-order time per byte on the private "
                       "streams in two consecutive evaluations (kept below %.2f; %llu evaluation(s) so far)%s\n",
               caller, to_lanes ? "RE-ADMITTED to the private streams" : "DEMOTED to its own stream", ratio, keep, (unsigned long long)evals,
               to_lanes ? "" : " — readers right behind their requests, sparse traffic, or a hardware-queue mapping that no longer overlaps "
                               "(fcp_plan_verify_private_streams searches a new one)");
}
} // namespace

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
    if (p->sup.demoted || p->sup.caller != stream) sup_reset(p->sup);
    p->lane_demoted.store(false, std::memory_order_release);
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

namespace {
// the request on the caller's own stream although the plan has private streams (small, captured, unverified, demoted)
int process_on_caller(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  fcp_process_result_t local{};
  if (!r) r = &local;
  const int rc = fcp_internal_process(p, a, r);
  if (rc == FCP_OK && r->buffer) pending_clear_range(r->buffer, r->buffer_bytes);
  return rc;
}
} // namespace

int fcp_process_feature_columns(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r) {
  if (!p || !a) return fail(FCP_ERR_INVALID_ARGUMENT, "null plan / args");
  if (!p->pool || p->lane_count == 0) return fcp_internal_process(p, a, r);
  hipStream_t caller = static_cast<hipStream_t>(a->stream);
  // small requests stay on the caller's stream; so does a capture, which records the caller's stream only (cross-stream
  // events would fork it)
  if (p->last_work_bytes.load(std::memory_order_relaxed) < p->lane_min_work || stream_is_capturing(caller)) return process_on_caller(p, a, r);
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
        rc = verify_lanes(p, caller, default_verify_budget_ms(), /*again=*/false, &ok);
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
  static const bool lane_stats = std::getenv("FCP_LANE_STATS") != nullptr; // diagnostic: host time of a private-stream request by part
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
    if (rc) return rc;
    if (so_window) {
      std::lock_guard<std::mutex> sup_lock(S.mu);
      if (S.phase == 1) {
        S.so_bytes += tl_work_bytes;
        if (++S.so_count >= kSupWindow) { // every counted request has been enqueued: the window ends behind this one
          HIP_TRY(hipEventRecord(S.b1, caller));
          S.phase = 2;
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
        S.phase = 3;
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
    static const bool attach = [] {
      const char *v = std::getenv("FCP_LANE_STOP_EVENT"); // tuning aid: 0 = always record a marker
      return !v || std::atoi(v) != 0;
    }();
    if (attach) fcp_set_stop_event(L.out[e]);
    static const bool alias_done = [] {
      const char *v = std::getenv("FCP_LANE_DONE_ALIAS"); // tuning aid: 0 = the slot records an event of its own
      return !v || std::atoi(v) != 0;
    }();
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
    if (rc) return rc;
    if (dep_left) return fail(FCP_ERR_HIP, "private streams: the request never reached its allocation");
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

namespace {
// the pending result whose address range contains x, or nullptr (g_pending_mu held)
const PendingResult *pending_find(uintptr_t x, uintptr_t *base) {
  auto it = g_pending.upper_bound(x);
  if (it == g_pending.begin()) return nullptr;
  --it;
  if (x >= it->second.end) return nullptr;
  if (base) *base = it->first;
  return &it->second;
}
} // namespace

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

namespace {
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
} // namespace

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
  static const size_t direct_max = [] { // tuning aid: FCP_CONCAT_HOST_DIRECT_MAX=<bytes> (0: always copy)
    const char *e = std::getenv("FCP_CONCAT_HOST_DIRECT_MAX");
    return e ? (size_t)std::atoll(e) : (size_t)1 << 20;
  }();
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

// ============================ request staging ===============================
// ConcatInputs + H2D as one step.  See include/fcp_hip.h.
#include "numa_util.h"
#include "pack_pool.h"

namespace {

struct StageSlot {
  char *h_blob = nullptr; // pinned
  char *h_blob_dev = nullptr; // the device's mapping of h_blob (what a zero-copy slot hands out)
  bool direct = false;    // the slot's current contents are read from the pinned buffer (zero copy, or the fallback below)
  char *d_blob = nullptr;
  int32_t *offsets = nullptr, *shapes = nullptr;
  hipEvent_t copied = nullptr;   // H2D of this slot done (copy stream)
  hipEvent_t consumed = nullptr; // consumer work of this slot enqueued before this point (caller's stream)
  bool consumed_valid = false;
};

} // namespace

struct fcp_stager {
  int device = 0;
  int64_t capacity = 0;
  int32_t max_inputs = 0, max_rank_sum = 0;
  std::vector<StageSlot> slots;
  size_t next = 0;
  int last = -1;                 // slot handed out by the previous call
  hipStream_t copy_stream = nullptr;
  fcp::PackPool *pool = nullptr;
  int n_threads = 1;
  std::mutex mu;
  std::vector<int64_t> byte_off, in_off; // scratch
  bool zero_copy = false;        // the kernels read the pinned ring over PCIe themselves (no H2D copy)
  // Fallback of the copying mode: on a busy host hipMemcpyAsync sometimes BLOCKS the caller for 25-70 us per call, request
  // after request (profiles/r04_pcie_staging_memcpy_anomaly.txt: one run in two with unpinned pack workers, rarely with
  // pinned ones), which doubles the pipelined time.  Eight such calls among the last 32 and the next 256 requests are served zero
  // copy (steady 65 us on S2); then the copy engine gets another chance.  FCP_STAGER_NO_FALLBACK=1 disables it.
  uint32_t blocked_hist = 0;
  int direct_left = 0;
  uint64_t n_fallbacks = 0, n_blocked = 0;
  // (r5) a request is packed in `groups` groups of inputs and every group is shipped as soon as it is packed — by the calling
  // thread, which watches the workers instead of packing — so the H2D copy of the first groups runs under the pack of the later ones: a lone request
  // costs pack + copy / groups + kernel instead of pack + copy + kernel.  copy_kernel: the copies are kernels on the copy
  // stream that read the pinned ring through its device mapping (no SDMA engine, no runtime copy path: the eliminating
  // experiment for the ~14 ms hipMemcpyAsync stalls, profiles/r05_pcie_staging_stalls.txt).
  int groups = 4;
  bool copy_kernel = true;
  // Groups cost throughput (the caller watches instead of packing, four copy launches instead of one: S2 58 -> 66 us per
  // request pipelined) and buy latency (lone request 142 -> 130 us): they are used when the caller is NOT issuing back to
  // back — more than kLatencyGapNs since the previous staging call returned — i.e. when nothing is there to overlap with
  // but the request itself.  FCP_STAGER_GROUPS_ALWAYS=1: every request.
  uint64_t t_last_return_ns = 0;
  bool groups_always = false;
  std::atomic<uint64_t> max_copy_call_ns{0}, n_copy_calls{0}, n_copy_over_1ms{0};
  uint64_t n_total_calls = 0;
  // FCP_STAGER_STATS=1: where a call spends its host time (ns per phase, printed when the stager is destroyed)
  bool stats = false;
  uint64_t n_calls = 0, ns_wait = 0, ns_layout = 0, ns_pack = 0, ns_enqueue = 0, ns_api[4] = {0, 0, 0, 0};
};

extern "C" {

int fcp_stager_create(int32_t device, int64_t capacity_bytes, int32_t max_inputs, int32_t max_rank_sum,
                      int32_t depth, int32_t n_threads, fcp_stager_t **out) {
  static const bool zero_copy_env = std::getenv("FCP_STAGER_ZERO_COPY") != nullptr; // tuning aid
  return fcp_stager_create_ex(device, capacity_bytes, max_inputs, max_rank_sum, depth, n_threads,
                              zero_copy_env ? FCP_STAGER_ZERO_COPY : FCP_STAGER_DEFAULT, out);
}

int fcp_stager_create_ex(int32_t device, int64_t capacity_bytes, int32_t max_inputs, int32_t max_rank_sum,
                         int32_t depth, int32_t n_threads, uint32_t flags, fcp_stager_t **out) {
  if (flags & ~(uint32_t)(FCP_STAGER_ZERO_COPY | FCP_STAGER_COPY_KERNEL | FCP_STAGER_COPY_SDMA)) return fail(FCP_ERR_INVALID_ARGUMENT, "unknown stager flags");
  if ((flags & FCP_STAGER_COPY_KERNEL) && (flags & FCP_STAGER_COPY_SDMA)) return fail(FCP_ERR_INVALID_ARGUMENT, "copy kernel and SDMA at once");
  if (!out || capacity_bytes <= 0 || capacity_bytes > 0x7fffffff || max_inputs <= 0 || max_rank_sum < 0 ||
      depth < 1 || n_threads < 1)
    return fail(FCP_ERR_INVALID_ARGUMENT, "bad stager parameters (capacity is limited to 2^31 bytes: int32 offsets)");
  *out = nullptr;
  DeviceGuard guard;
  int rc = guard.enter(device);
  if (rc) return rc;
  fcp_stager *s = new (std::nothrow) fcp_stager();
  if (!s) return fail(FCP_ERR_ALLOC, "out of host memory");
  s->device = device;
  s->capacity = capacity_bytes;
  s->max_inputs = max_inputs;
  s->max_rank_sum = max_rank_sum;
  s->n_threads = n_threads;
  s->zero_copy = (flags & FCP_STAGER_ZERO_COPY) != 0;
  // Copies are KERNELS by default since round 5: hipMemcpyAsync's SDMA submission (hsa_amd_memory_async_copy_on_engine) blocks
  // its caller for 6-14 ms a few times per thousand calls on this pool's boxes; 28 fresh processes with kernel copies (or
  // HSA_ENABLE_SDMA=0) showed none (profiles/r05_pcie_staging_stalls.txt).  FCP_STAGER_COPY_SDMA / FCP_STAGER_COPY=sdma: the engine.
  s->copy_kernel = (flags & FCP_STAGER_COPY_SDMA) == 0;
  if (const char *e = std::getenv("FCP_STAGER_COPY")) s->copy_kernel = std::strcmp(e, "sdma") != 0; // tuning aid: kernel | sdma
  s->groups_always = std::getenv("FCP_STAGER_GROUPS_ALWAYS") != nullptr;
  if (const char *e = std::getenv("FCP_STAGER_GROUPS")) s->groups = std::max(1, std::min(std::atoi(e), 16)); // tuning aid; 1 = one copy per request
  s->stats = std::getenv("FCP_STAGER_STATS") != nullptr;
  s->slots.resize(depth);
  for (auto &sl : s->slots) {
    const size_t alloc_bytes = (size_t)capacity_bytes + 64; // (the copy kernel rounds a group's range up to 16 bytes)
    if (hipHostMalloc(reinterpret_cast<void **>(&sl.h_blob), alloc_bytes, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&sl.h_blob_dev), sl.h_blob, 0) != hipSuccess ||
        (s->zero_copy ? ((sl.d_blob = sl.h_blob_dev), hipSuccess) : hipMalloc(reinterpret_cast<void **>(&sl.d_blob), alloc_bytes)) != hipSuccess ||
        hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming) != hipSuccess) {
      fcp_stager_destroy(s);
      return hip_fail("stager allocation", hipGetLastError());
    }
    sl.offsets = new int32_t[max_inputs];
    sl.shapes = new int32_t[max_rank_sum > 0 ? max_rank_sum : 1];
  }
  if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    fcp_stager_destroy(s);
    return hip_fail("stager copy stream", hipGetLastError());
  }
  if (!s->zero_copy && !s->slots.empty()) {
    // the copy kernel's first launch loads its code object (~3.5 ms, once per process): here, not inside a request
    std::memset(s->slots[0].h_blob, 0, 64);
    if (fcp_launch_h2d_copy(s->slots[0].h_blob_dev, s->slots[0].d_blob, 64, s->copy_stream) != 0 ||
        hipStreamSynchronize(s->copy_stream) != hipSuccess) {
      fcp_stager_destroy(s);
      return hip_fail("stager copy kernel warm-up", hipGetLastError());
    }
  }
  {
    // pack next to the GPU: the H2D copy reads the pinned ring from that socket's memory
    cpu_set_t near;
    static const bool no_pin = std::getenv("FCP_STAGER_NO_PIN") != nullptr; // tuning aid
    const bool pin = !no_pin && fcp::cpus_near_device(device, &near);
    s->pool = new fcp::PackPool(n_threads, pin ? &near : nullptr);
  }
  s->byte_off.resize(max_inputs + 1);
  *out = s;
  return FCP_OK;
}

int fcp_stager_stage_narrow(fcp_stager_t *s, const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *narrow,
                            void *stream, const void **device_blob, int64_t *blob_bytes, const int32_t **offsets,
                            const int32_t **shapes) {
  // the flags are booleans: anything non-zero means "narrow" (FCP_STAGE_NARROW_I64), never another mode
  std::vector<uint8_t> modes;
  if (narrow && n > 0) {
    modes.resize(n);
    for (int32_t i = 0; i < n; ++i) modes[i] = narrow[i] ? FCP_STAGE_NARROW_I64 : FCP_STAGE_COPY;
  }
  return fcp_stager_stage_ex(s, inputs, n, modes.empty() ? nullptr : modes.data(), nullptr, stream, device_blob, blob_bytes, offsets,
                             shapes);
}

namespace {
// the two host loops of the staged pack, built per instruction set (fcp_pack.cc)
extern "C" void fcp_pack_narrow_i64(const int64_t *src, int32_t *dst, int64_t n);
extern "C" int fcp_pack_seg_to_csr(const void *seg, int elem_size, int64_t stride, int64_t nnz, int64_t rows, int32_t *out);

// Layout of the staged blob: byte offsets (byte_off[0..n]), the op's `offsets` and `shapes` outputs — exactly
// ConcatInputsOp::Compute (concat_inputs_ops.cc:52-66), except that a narrowed int64 input occupies 4 bytes per element
// and a converted row-id input is int32[rows + 1] (one dim).  max_rank_sum < 0: no limit.
int stage_layout(const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes, const int64_t *mode_args, int64_t capacity,
                 int32_t max_rank_sum, int64_t *byte_off, int32_t *offsets, int32_t *shapes, int32_t *rank_sum_out) {
  int64_t size = 0;
  int32_t rank_sum = 0;
  for (int32_t i = 0; i < n; ++i) {
    const fcp_host_tensor_t &t = inputs[i];
    if (t.rank < 0 || t.elem_size <= 0 || (t.rank > 0 && !t.dims)) return fail(FCP_ERR_INVALID_ARGUMENT, "bad host tensor");
    const int mode = modes ? modes[i] : FCP_STAGE_COPY;
    if (mode > FCP_STAGE_SEG_TO_CSR) return fail(FCP_ERR_INVALID_ARGUMENT, "unknown staging mode");
    if (mode == FCP_STAGE_NARROW_I64 && t.elem_size != 8) return fail(FCP_ERR_INVALID_ARGUMENT, "only 8-byte inputs can be narrowed");
    if (max_rank_sum >= 0 && rank_sum + t.rank > max_rank_sum) return fail(FCP_ERR_INVALID_ARGUMENT, "more dims than the stager was created for");
    int64_t ne = 1;
    for (int32_t j = 0; j < t.rank; ++j) {
      if (t.dims[j] < 0) return fail(FCP_ERR_INVALID_ARGUMENT, "negative dimension");
      ne *= t.dims[j];
    }
    byte_off[i] = size;
    if (offsets) offsets[i] = (int32_t)size;
    if (mode == FCP_STAGE_SEG_TO_CSR) {
      // sorted row ids [nnz] or SparseTensor indices [nnz, k] -> int32 offsets[rows + 1]: one dim in the shapes
      if ((t.elem_size != 4 && t.elem_size != 8) || t.rank < 1 || t.rank > 2 || !mode_args || mode_args[i] < 0 ||
          mode_args[i] >= 0x7fffffff || (t.rank == 2 && t.dims[1] < 1))
        return fail(FCP_ERR_INVALID_ARGUMENT, "segment-id input to convert: int32 / int64 [nnz] or [nnz, k], with its number of rows");
      if (shapes) shapes[rank_sum] = (int32_t)(mode_args[i] + 1);
      rank_sum += 1;
      size += (mode_args[i] + 1) * 4;
    } else {
      for (int32_t j = 0; j < t.rank; ++j)
        if (shapes) shapes[rank_sum + j] = (int32_t)t.dims[j];
      rank_sum += t.rank;
      size += ne * (mode == FCP_STAGE_NARROW_I64 ? 4 : t.elem_size);
    }
    if (capacity >= 0 && size > capacity) return fail(FCP_ERR_INVALID_ARGUMENT, "request larger than the blob / stager capacity");
    // The reference keeps offsets in int32 (:52-60); refuse what it would overflow.
    if (size > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "blob larger than 2^31 bytes (int32 offsets)");
    if (ne && !t.data) return fail(FCP_ERR_INVALID_ARGUMENT, "null tensor data");
  }
  byte_off[n] = size;
  if (rank_sum_out) *rank_sum_out = rank_sum;
  return FCP_OK;
}

// A request is a thousand or two SMALL tensors (RAGGED: 11 KB of ids and 22 KB of indices per column), each somewhere else
// in memory: the hardware prefetcher starts over at every one of them.  While tensor i is packed the head of tensor
// i + 1 is requested (FCP_PACK_PREFETCH_BYTES, default 2 KiB: tuning aid; 0 = off).
inline void prefetch_head(const fcp_host_tensor_t &t) {
  static const int64_t bytes = [] {
    const char *e = std::getenv("FCP_PACK_PREFETCH_BYTES");
    return e ? std::atoll(e) : (int64_t)2048;
  }();
  if (!t.data) return;
  int64_t n = t.elem_size;
  for (int32_t j = 0; j < t.rank; ++j) n *= t.dims[j];
  if (n > bytes) n = bytes;
  static const int hint = [] { // FCP_PACK_PREFETCH_HINT (tuning aid): 0 = non-temporal, 1 = every cache level, 2 = L2 and up
    const char *e = std::getenv("FCP_PACK_PREFETCH_HINT");
    return e ? std::atoi(e) : 0;
  }();
  const char *p = static_cast<const char *>(t.data);
  if (hint == 2) {
    for (int64_t o = 0; o < n; o += 64) __builtin_prefetch(p + o, 0, 2);
  } else if (hint == 1) {
    for (int64_t o = 0; o < n; o += 64) __builtin_prefetch(p + o, 0, 3);
  } else {
    for (int64_t o = 0; o < n; o += 64) __builtin_prefetch(p + o, 0, 0);
  }
}

// One input into its place in the staged blob (`nbytes` = its bytes there).
// returns false for an input that cannot be converted: row ids that are not sorted (TF's SparseSegment* ops refuse them too)
bool stage_pack_one(const fcp_host_tensor_t &t, int mode, int64_t mode_arg, char *dst, int64_t nbytes) {
  if (nbytes <= 0) return true;
  if (mode == FCP_STAGE_SEG_TO_CSR) {
    return fcp_pack_seg_to_csr(t.data, t.elem_size, t.rank == 2 ? t.dims[1] : 1, t.dims[0], mode_arg, reinterpret_cast<int32_t *>(dst)) == 0;
  } else if (mode == FCP_STAGE_NARROW_I64) {
    fcp_pack_narrow_i64(static_cast<const int64_t *>(t.data), reinterpret_cast<int32_t *>(dst), nbytes / 4);
  } else {
    std::memcpy(dst, t.data, (size_t)nbytes);
  }
  return true;
}
const char *const kUnsortedRows = "row ids of a converted input are not sorted (segment ids must be non-decreasing)";

// The pack of one request on a pool: contiguous ranges of inputs of about equal INPUT bytes per chunk (a converted index
// matrix is 16 bytes per id in and 4 bytes per ROW out: output bytes say little about the work), the head of the next
// input requested while the current one is packed.  `in_off`: scratch, n + 1 entries.  false: some row ids were not sorted.
// `groups` (optional): the chunks are dealt into groups->n consecutive groups of about equal input bytes; the CALLING thread
// does not pack (the pool's workers do) but watches the groups complete and calls groups->done(ctx, group, first byte, end byte
// of the group in the blob) for each, in order — the stager ships a group over PCIe while the later groups are still being
// packed.  (Runtime calls from the pack workers themselves: hipMemcpyAsync issued from many threads stalled for ~8.5 ms a
// dozen times per 700 requests, profiles/r05_pcie_staging_stalls.txt.)
struct PackGroups {
  int n;
  void (*done)(void *ctx, int group, int64_t byte_begin, int64_t byte_end);
  void *ctx;
};
bool pack_on_pool(fcp::PackPool &pool, int n_threads, const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes,
                  const int64_t *mode_args, char *dst, const int64_t *bo, int64_t *in_off, const PackGroups *groups = nullptr) {
  in_off[0] = 0;
  for (int32_t i = 0; i < n; ++i) {
    int64_t ne = 1;
    for (int32_t j = 0; j < inputs[i].rank; ++j) ne *= inputs[i].dims[j];
    in_off[i + 1] = in_off[i] + ne * inputs[i].elem_size;
  }
  const int64_t total = in_off[n];
  static const int per_thread = [] { // tuning aid: chunks per pack thread (every chunk costs two contended atomics)
    const char *e = std::getenv("FCP_PACK_CHUNKS_PER_THREAD");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 4;
  }();
  const int chunks = (int)std::min<int64_t>(std::max<int64_t>(1, total / (64 << 10)), (int64_t)per_thread * n_threads);
  std::atomic<int> refused{0};
  // group bookkeeping: chunk c belongs to group c * ng / chunks; first input of every chunk up front (the groups' byte ranges)
  constexpr int kMaxGroups = 16;
  const int ng = groups ? std::max(1, std::min(std::min(groups->n, chunks), kMaxGroups)) : 0;
  std::atomic<int> group_left[kMaxGroups];
  std::vector<int> chunk_lo;
  if (ng > 0) {
    chunk_lo.resize((size_t)chunks + 1);
    for (int c = 0; c < chunks; ++c) chunk_lo[(size_t)c] = (int)(std::lower_bound(in_off, in_off + n, total * c / chunks) - in_off);
    chunk_lo[(size_t)chunks] = n;
    for (int g = 0; g < ng; ++g) group_left[g].store(0, std::memory_order_relaxed);
    for (int c = 0; c < chunks; ++c) group_left[c * ng / chunks].fetch_add(1, std::memory_order_relaxed);
  }
  // FCP_PACK_TRACE=1 (diagnostic): when did every chunk of a call start and end, and on which thread — printed for every 128th call
  static const bool trace = std::getenv("FCP_PACK_TRACE") != nullptr;
  static std::atomic<uint64_t> n_calls{0};
  struct ChunkTrace { uint64_t t0, t1; unsigned long tid; };
  std::vector<ChunkTrace> tr;
  const bool tracing = trace && (n_calls.fetch_add(1) & 127) == 100;
  auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  if (tracing) tr.resize((size_t)chunks);
  const uint64_t t_pub = tracing ? now_ns() : 0;
  auto chunk_fn = [&](int c) {
    struct Stamp { // (scope guard: the end stamp on every exit path of the chunk)
      ChunkTrace *e;
      uint64_t (*now)();
      ~Stamp() { if (e) e->t1 = now(); }
    };
    if (tracing) {
      tr[(size_t)c].t0 = now_ns();
      tr[(size_t)c].tid = (unsigned long)pthread_self();
    }
    Stamp stamp{tracing ? &tr[(size_t)c] : nullptr, +[] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }};
    const int64_t b0 = total * c / chunks, b1 = total * (c + 1) / chunks;
    int lo = (int)(std::lower_bound(in_off, in_off + n, b0) - in_off);
    const int hi = c + 1 == chunks ? n : (int)(std::lower_bound(in_off, in_off + n, b1) - in_off); // (the last chunk also takes trailing empty inputs)
    for (; lo < hi; ++lo) {
      if (lo + 1 < hi) prefetch_head(inputs[lo + 1]);
      if (!stage_pack_one(inputs[lo], modes ? modes[lo] : FCP_STAGE_COPY, mode_args ? mode_args[lo] : 0, dst + bo[lo], bo[lo + 1] - bo[lo]))
        refused.store(1, std::memory_order_relaxed);
    }
    if (ng > 0) group_left[c * ng / chunks].fetch_sub(1, std::memory_order_release);
  };
  auto ship = [&](int g) { // (all chunks of group g are packed)
    int c0 = 0;
    while (c0 < chunks && c0 * ng / chunks < g) ++c0;
    int c1 = c0;
    while (c1 < chunks && c1 * ng / chunks == g) ++c1;
    groups->done(groups->ctx, g, bo[chunk_lo[(size_t)c0]], bo[chunk_lo[(size_t)c1]]);
  };
  if (ng > 0 && pool.start(chunks, chunk_fn)) {
    for (int g = 0; g < ng; ++g) {
      // the last group: help with what is left instead of watching
      if (g == ng - 1) pool.finish();
      while (group_left[g].load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
      ship(g);
    }
    pool.finish();
  } else {
    pool.run(chunks, chunk_fn);
    for (int g = 0; g < ng; ++g) ship(g);
  }
  if (tracing) {
    const uint64_t t_end = now_ns();
    std::fprintf(stderr, "fcp pack trace: %d chunks, %d threads, call %.1f us; chunk: start-after-publish us, duration us, thread\n", chunks,
                 n_threads, (t_end - t_pub) / 1e3);
    for (int c = 0; c < chunks; ++c)
      std::fprintf(stderr, "  %2d: %6.1f %6.1f %lx\n", c, (tr[(size_t)c].t0 - t_pub) / 1e3, (tr[(size_t)c].t1 - tr[(size_t)c].t0) / 1e3, tr[(size_t)c].tid & 0xffffff);
  }
  return refused.load() == 0;
}
} // namespace

int fcp_stager_stage_ex(fcp_stager_t *s, const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes,
                        const int64_t *mode_args, void *stream, const void **device_blob, int64_t *blob_bytes,
                        const int32_t **offsets, const int32_t **shapes) {
  if (!s || n < 0 || (n > 0 && !inputs)) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  if (n > s->max_inputs) return fail(FCP_ERR_INVALID_ARGUMENT, "more inputs than the stager was created for");
  DeviceGuard guard;
  int rc = guard.enter(s->device);
  if (rc) return rc;
  hipStream_t user = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(s->mu);
  auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const uint64_t t_begin = s->stats ? now_ns() : 0;
  static const bool early_wake = std::getenv("FCP_STAGER_NO_EARLY_WAKE") == nullptr; // tuning aid
  if (early_wake) s->pool->expect(); // the pack workers wake up while this thread waits for the slot and lays the blob out
  // whatever consumes the previous slot has been enqueued on the caller's stream by now
  if (s->last >= 0) {
    StageSlot &prev = s->slots[s->last];
    HIP_TRY(hipEventRecord(prev.consumed, user));
    prev.consumed_valid = true;
  }
  const int slot_idx = (int)s->next;
  StageSlot &sl = s->slots[s->next];
  s->next = (s->next + 1) % s->slots.size();
  // the slot's previous copy must have left the pinned buffer (zero copy: the kernels that read it must have run)
  if (s->zero_copy || sl.direct) {
    if (sl.consumed_valid && hipEventQuery(sl.consumed) != hipSuccess) HIP_TRY(hipEventSynchronize(sl.consumed));
    // a plan with private streams runs the reader elsewhere; `consumed` (caller's stream) does not cover it
    if (int rc3 = stager_input_synchronize(sl.h_blob_dev, s->capacity)) return rc3;
  } else if (hipEventQuery(sl.copied) != hipSuccess) {
    HIP_TRY(hipEventSynchronize(sl.copied));
  }
  static const bool no_fallback = std::getenv("FCP_STAGER_NO_FALLBACK") != nullptr;
  const bool direct = s->zero_copy || s->direct_left > 0;
  if (s->direct_left > 0) --s->direct_left;
  const uint64_t t_waited = s->stats ? now_ns() : 0;
  // sizes / offsets / shapes (stage_layout), then the pack: contiguous ranges of inputs per chunk, ~equal bytes
  int rc2 = stage_layout(inputs, n, modes, mode_args, s->capacity, s->max_rank_sum, s->byte_off.data(), sl.offsets, sl.shapes, nullptr);
  if (rc2) return rc2;
  const int64_t size = s->byte_off[n];
  const uint64_t t_layout = s->stats ? now_ns() : 0;
  s->in_off.resize((size_t)n + 1);
  // Copying mode: the slot's device twin is free once the work that read its previous contents has run — queued on the copy
  // stream BEFORE the pack, because the groups of this request are shipped from inside it
  struct Ship {
    fcp_stager *s;
    StageSlot *sl;
    std::atomic<int> err{0};
    std::atomic<uint64_t> max_ns{0};
  } ship{s, &sl};
  constexpr uint64_t kLatencyGapNs = 40000;
  const bool lone = s->groups_always || s->t_last_return_ns == 0 || now_ns() - s->t_last_return_ns > kLatencyGapNs;
  PackGroups pg{lone ? s->groups : 1, nullptr, &ship};
  pg.done = [](void *ctx, int, int64_t b0, int64_t b1) {
    Ship &x = *static_cast<Ship *>(ctx);
    if (b1 <= b0) return;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e;
    if (x.s->copy_kernel) {
      // the kernel moves 16-byte words: the range is widened to 16-byte boundaries (tensors of 1- or 2-byte elements put group
      // boundaries anywhere).  The bytes it picks up from a neighbouring group are either final already (the group before: groups
      // are shipped in order) or rewritten by that group's own copy, which follows on the same stream; the buffers have slack.
      const int64_t a0 = b0 & ~(int64_t)15, a1 = (b1 + 15) & ~(int64_t)15;
      e = (hipError_t)fcp_launch_h2d_copy(x.sl->h_blob_dev + a0, x.sl->d_blob + a0, (size_t)(a1 - a0), x.s->copy_stream);
    } else
      e = hipMemcpyAsync(x.sl->d_blob + b0, x.sl->h_blob + b0, (size_t)(b1 - b0), hipMemcpyHostToDevice, x.s->copy_stream);
    if (e != hipSuccess) x.err.store((int)e, std::memory_order_relaxed);
    const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    uint64_t m = x.max_ns.load(std::memory_order_relaxed);
    while (ns > m && !x.max_ns.compare_exchange_weak(m, ns, std::memory_order_relaxed)) {
    }
    x.s->n_copy_calls.fetch_add(1, std::memory_order_relaxed);
    if (ns > 1000000) x.s->n_copy_over_1ms.fetch_add(1, std::memory_order_relaxed);
    m = x.s->max_copy_call_ns.load(std::memory_order_relaxed);
    while (ns > m && !x.s->max_copy_call_ns.compare_exchange_weak(m, ns, std::memory_order_relaxed)) {
    }
  };
  uint64_t a0 = now_ns(), a1;
  if (!direct) {
    if (sl.consumed_valid) HIP_TRY(hipStreamWaitEvent(s->copy_stream, sl.consumed, 0));
    if (int rc3 = stager_input_wait(sl.d_blob, s->capacity, s->copy_stream)) return rc3; // (private-stream readers, see there)
    if (s->stats) { a1 = now_ns(); s->ns_api[0] += a1 - a0; }
  }
  const uint64_t t_pack0 = s->stats ? now_ns() : 0;
  const bool sorted = pack_on_pool(*s->pool, s->n_threads, inputs, n, modes, mode_args, sl.h_blob, s->byte_off.data(), s->in_off.data(),
                                   direct || size == 0 ? nullptr : &pg);
  const uint64_t t_packed = s->stats ? now_ns() : 0;
  (void)t_pack0;
  sl.direct = direct;
  if (!direct) {
    // (whatever was shipped before a refusal is ordinary traffic on the copy stream: the slot is simply reused)
    if (ship.err.load()) return hip_fail("H2D copy of a request group", (hipError_t)ship.err.load());
    const uint64_t worst = ship.max_ns.load();
    if (!no_fallback && sorted) {
      s->blocked_hist = (s->blocked_hist << 1) | (worst > 20000 ? 1u : 0u); // the last 32 requests: whose copy calls blocked > 20 us
      s->n_blocked += worst > 20000;
      if (__builtin_popcount(s->blocked_hist) >= 8) { // the copy call holds the host up: the kernels read the pinned ring for a while
        s->blocked_hist = 0;
        s->direct_left = 256;
        ++s->n_fallbacks;
      }
    }
    if (s->stats) s->ns_api[1] += worst;
  }
  if (!sorted) return fail(FCP_ERR_INVALID_ARGUMENT, kUnsortedRows);
  if (!direct) {
    a0 = now_ns();
    HIP_TRY(hipEventRecord(sl.copied, s->copy_stream));
    if (s->stats) { a1 = now_ns(); s->ns_api[2] += a1 - a0; a0 = a1; }
    HIP_TRY(hipStreamWaitEvent(user, sl.copied, 0));
    if (s->stats) { a1 = now_ns(); s->ns_api[3] += a1 - a0; }
  } else {
    __atomic_thread_fence(__ATOMIC_SEQ_CST); // the packed bytes are in memory before the launch that reads them is queued
  }
  if (s->stats) {
    const uint64_t t_end = now_ns();
    ++s->n_calls;
    s->ns_wait += t_waited - t_begin;
    s->ns_layout += t_layout - t_waited;
    s->ns_pack += t_packed - t_layout;
    s->ns_enqueue += t_end - t_packed;
  }
  s->last = slot_idx;
  ++s->n_total_calls;
  s->t_last_return_ns = now_ns();
  if (device_blob) *device_blob = direct ? sl.h_blob_dev : sl.d_blob;
  if (blob_bytes) *blob_bytes = size;
  if (offsets) *offsets = sl.offsets;
  if (shapes) *shapes = sl.shapes;
  return FCP_OK;
}

// ---- Addons>ConcatInputs, staged form (host only; no stager, no device) --------------------------------
int fcp_concat_inputs_ex_sizes(const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes, const int64_t *mode_args,
                               int64_t *blob_bytes, int32_t *rank_sum) {
  if (n < 0 || (n > 0 && !inputs)) return fail(FCP_ERR_INVALID_ARGUMENT, "null inputs");
  std::vector<int64_t> bo((size_t)n + 1);
  int32_t ranks = 0;
  const int rc = stage_layout(inputs, n, modes, mode_args, -1, -1, bo.data(), nullptr, nullptr, &ranks);
  if (rc) return rc;
  if (blob_bytes) *blob_bytes = bo[n];
  if (rank_sum) *rank_sum = ranks;
  return FCP_OK;
}

int fcp_concat_inputs_ex(const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes, const int64_t *mode_args, void *blob,
                         int64_t blob_capacity, int32_t *offsets, int32_t *shapes) {
  if (n < 0 || (n > 0 && (!inputs || !offsets || !shapes))) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<int64_t> bo((size_t)n + 1);
  const int rc = stage_layout(inputs, n, modes, mode_args, blob_capacity, -1, bo.data(), offsets, shapes, nullptr);
  if (rc) return rc;
  if (bo[n] > 0 && !blob) return fail(FCP_ERR_INVALID_ARGUMENT, "blob too small");
  bool ok = true;
  for (int32_t i = 0; i < n; ++i)
    ok = stage_pack_one(inputs[i], modes ? modes[i] : FCP_STAGE_COPY, mode_args ? mode_args[i] : 0, static_cast<char *>(blob) + bo[i],
                        bo[i + 1] - bo[i]) && ok;
  return ok ? FCP_OK : fail(FCP_ERR_INVALID_ARGUMENT, kUnsortedRows);
}

struct fcp_pack_pool {
  fcp::PackPool *pool = nullptr;
  int n_threads = 1;
  std::mutex busy; // one call at a time splits its work over the pool; others pack on their own thread
};

int fcp_pack_pool_create(int32_t n_threads, fcp_pack_pool_t **out) {
  if (!out || n_threads < 1 || n_threads > 1024) return fail(FCP_ERR_INVALID_ARGUMENT, "bad pack pool arguments");
  fcp_pack_pool *p = new (std::nothrow) fcp_pack_pool();
  if (!p) return fail(FCP_ERR_ALLOC, "out of host memory");
  p->n_threads = n_threads;
  p->pool = new fcp::PackPool(n_threads);
  *out = p;
  return FCP_OK;
}

int fcp_pack_pool_destroy(fcp_pack_pool_t *p) {
  if (!p) return FCP_OK;
  {
    std::lock_guard<std::mutex> lock(p->busy); // a call in flight finishes first
  }
  delete p->pool;
  delete p;
  return FCP_OK;
}

int fcp_concat_inputs_ex_pool(fcp_pack_pool_t *pool, const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes,
                              const int64_t *mode_args, void *blob, int64_t blob_capacity, int32_t *offsets, int32_t *shapes) {
  if (!pool || pool->n_threads <= 1 || n < 2) return fcp_concat_inputs_ex(inputs, n, modes, mode_args, blob, blob_capacity, offsets, shapes);
  std::unique_lock<std::mutex> mine(pool->busy, std::try_to_lock);
  if (!mine.owns_lock()) return fcp_concat_inputs_ex(inputs, n, modes, mode_args, blob, blob_capacity, offsets, shapes);
  pool->pool->expect(); // the workers wake up while this thread lays the blob out
  if (!inputs || !offsets || !shapes) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<int64_t> bo((size_t)n + 1), in_off((size_t)n + 1);
  const int rc = stage_layout(inputs, n, modes, mode_args, blob_capacity, -1, bo.data(), offsets, shapes, nullptr);
  if (rc) return rc;
  if (bo[n] > 0 && !blob) return fail(FCP_ERR_INVALID_ARGUMENT, "blob too small");
  return pack_on_pool(*pool->pool, pool->n_threads, inputs, n, modes, mode_args, static_cast<char *>(blob), bo.data(), in_off.data())
             ? FCP_OK
             : fail(FCP_ERR_INVALID_ARGUMENT, kUnsortedRows);
}

int fcp_stager_stage(fcp_stager_t *s, const fcp_host_tensor_t *inputs, int32_t n, void *stream,
                     const void **device_blob, int64_t *blob_bytes, const int32_t **offsets,
                     const int32_t **shapes) {
  return fcp_stager_stage_narrow(s, inputs, n, nullptr, stream, device_blob, blob_bytes, offsets, shapes);
}

int fcp_stager_stats(fcp_stager_t *s, fcp_stager_stats_t *out) {
  if (!s || !out) return fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lock(s->mu);
  out->calls = (int64_t)s->n_total_calls;
  out->copy_calls = (int64_t)s->n_copy_calls.load();
  out->copy_calls_over_1ms = (int64_t)s->n_copy_over_1ms.load();
  out->fallback_switches = (int64_t)s->n_fallbacks;
  out->requests_with_blocked_copy = (int64_t)s->n_blocked;
  out->max_copy_call_us = s->max_copy_call_ns.load() / 1e3;
  return FCP_OK;
}

int fcp_stager_destroy(fcp_stager_t *s) {
  if (!s) return FCP_OK;
  if (s->stats && s->n_calls)
    std::fprintf(stderr, "fcp_stager: %llu calls, host us per call: wait for the slot %.2f, layout %.2f, pack %.2f (%d threads), enqueue %.2f "
                         "(wait-event on the copy stream %.2f, hipMemcpyAsync %.2f, event record %.2f, wait-event on the request's stream %.2f); "
                         "requests whose copy calls blocked > 20 us: %llu, zero-copy fallbacks %llu; %llu copy calls (%s, %d groups), slowest %.1f us, "
                         "%llu over 1 ms\n",
                 (unsigned long long)s->n_calls, s->ns_wait / 1e3 / s->n_calls, s->ns_layout / 1e3 / s->n_calls, s->ns_pack / 1e3 / s->n_calls,
                 s->n_threads, s->ns_enqueue / 1e3 / s->n_calls, s->ns_api[0] / 1e3 / s->n_calls, s->ns_api[1] / 1e3 / s->n_calls,
                 s->ns_api[2] / 1e3 / s->n_calls, s->ns_api[3] / 1e3 / s->n_calls, (unsigned long long)s->n_blocked, (unsigned long long)s->n_fallbacks,
                 (unsigned long long)s->n_copy_calls.load(), s->copy_kernel ? "copy kernel" : "hipMemcpyAsync", s->groups,
                 s->max_copy_call_ns.load() / 1e3, (unsigned long long)s->n_copy_over_1ms.load());
  DeviceGuard guard;
  (void)guard.enter(s->device);
  delete s->pool;
  (void)hipDeviceSynchronize(); // kernels may still be reading the pinned ring (zero copy, or a slot served by the fallback)
  if (s->copy_stream) {
    (void)hipStreamSynchronize(s->copy_stream);
    (void)hipStreamDestroy(s->copy_stream);
  }
  for (auto &sl : s->slots) {
    if (sl.copied) {
      (void)hipEventSynchronize(sl.copied);
      (void)hipEventDestroy(sl.copied);
    }
    if (sl.consumed) (void)hipEventDestroy(sl.consumed);
    if (sl.d_blob && !s->zero_copy) (void)hipFree(sl.d_blob);
    if (sl.h_blob) (void)hipHostFree(sl.h_blob);
    delete[] sl.offsets;
    delete[] sl.shapes;
  }
  delete s;
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
