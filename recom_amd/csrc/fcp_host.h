// fcp_host.h — what the host-side translation units of libfcp_hip.so share: the plan object, the per-request descriptor
// cache, the private-stream (lane) state and the helpers that cross files.  Internal: nothing here is part of the ABI
// (include/fcp_hip.h).  Round 6 carved the single 3 700-line fcp_api.hip into
//   fcp_plan.hip     plan creation / validation / plan files / geometry / const buffers / placement / accessors
//   fcp_process.hip  the request path: descriptor slots, launch records, fcp_internal_process, fcp_process_feature_columns
//   fcp_lanes.hip    plan-owned private streams (EXPERIMENTAL): lanes, verification, supervisor, result registry
//   fcp_concat.hip   Addons>ConcatOutputs entry points and fcp_shard_finalize
//   fcp_stager.hip   Addons>ConcatInputs packers, the request stager, the pack pool
#pragma once
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
#include "fcp_env.h"

namespace fcph {

extern thread_local std::string g_last_error;
int hip_fail(const char *what, hipError_t e);
int fail(int code, const std::string &msg);

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
extern std::atomic<uint64_t> g_lane_generation;
// the completion event of the private-stream request this thread is issuing (fcp_process_feature_columns -> fcp_internal_process)
extern thread_local hipEvent_t tl_lane_done;
extern thread_local uint64_t tl_lane_done_gen;

// What a private-stream request waits for on the caller's stream is known only once the allocator has answered: TF's
// allocator hands out memory in compute-stream order, and between this call's entry and its malloc_buff another
// Session::Run thread may have queued a kernel that still reads the very memory the arena gets.  So
// fcp_process_feature_columns leaves the dependency here and fcp_internal_process enqueues it — event record on the caller's
// stream, wait on the lane — right behind malloc_buff, in front of the first command that touches the arena.
struct LaneDep {
  hipEvent_t in;
  hipStream_t caller, lane;
};
extern thread_local const LaneDep *tl_lane_dep;
// gathered + written bytes of the request this thread processed last (DynMeta::work_bytes; the supervisor's unit of work)
extern thread_local int64_t tl_work_bytes;

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
void sup_reset(LaneSupervisor &S, const fcp::Env &env);
LanePool *lane_pool_for(int device);

// Results whose kernels run on a private lane, by arena address range: what fcp_result_wait looks up.  An entry is
// replaced when its address range is handed out again and dropped once its event has completed (nothing to wait for).
struct PendingResult {
  uintptr_t end;
  hipEvent_t done;
  const void *owner; // the plan whose lane owns `done`
};
// (fcp_lanes.hip) the registry of results / inputs whose kernels run on a private lane
void pending_register(const void *owner, void *arena, int64_t bytes, const void *blob, int64_t blob_bytes, hipEvent_t done);
void pending_clear_range(void *arena, int64_t bytes);
void pending_forget(const void *owner);
int stager_input_wait(const void *base, int64_t bytes, hipStream_t stream);
int stager_input_synchronize(const void *base, int64_t bytes);
int wait_for_inputs(const void *const *inputs, int32_t n, void *stream);

} // namespace fcph
using namespace fcph; // (an internal header: every includer is one of the five files above)

struct fcp_plan {
  fcp::Env env; // the shipping environment switches as they were when the plan was created (fcp_env.h)
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
  std::atomic<bool> sup_window{false};                  // a timing window of an evaluation is open (sup.phase 1 or 3): what a
                                                        // request that bypasses the supervisor looks at (sup_abort)
};

namespace fcph {

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

// ---- functions that cross translation units -----------------------------------------------------------------------
// fcp_plan.hip
int compute_dyn(const fcp_plan *p, const int32_t *offsets, const int32_t *shapes, const int32_t *symbols, int64_t blob_bytes,
                FcpColDyn *dyn, DynMeta *m);
// fcp_process.hip
bool stream_is_capturing(hipStream_t stream);
int bind_tables(fcp_plan *p, const void *const *input_ptrs, bool capturing);
void build_key(const fcp_plan *p, const fcp_process_args_t *a, std::vector<int32_t> &key);
int find_or_reserve(fcp_plan *p, const std::vector<int32_t> &key, DynSlot **out, bool *install, bool capturing);
int install_slot(fcp_plan *p, const fcp_process_args_t *a, DynSlot &s);
void publish_slot(fcp_plan *p, DynSlot &s, const std::vector<int32_t> &key, void *stream);
void fill_launch(const fcp_plan *p, const DynSlot &s, int kind, const void *blob, void *arena, int store_policy, FcpLaunch *L);
// fcp_lanes.hip
void destroy_lanes(LanePool *pool);
int create_lanes(LanePool *pool, int n, int prio);
int process_on_private_streams(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r);

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

} // namespace fcph

// the request on the stream it names (fcp_process.hip; also called by fcp_shard.hip)
extern "C" int fcp_internal_process(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r);
