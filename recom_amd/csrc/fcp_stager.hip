// fcp_stager.hip — the host packers of Addons>ConcatInputs (custom_ops/concat_inputs/concat_inputs_ops.cc:42-77; staged form),
// the request stager (pinned ring, pack pool, H2D) and the pack pool.  Carved out of fcp_api.hip in round 6 (see fcp_host.h).
#include "fcp_host.h"

extern "C" {

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
  return fcp_stager_create_ex(device, capacity_bytes, max_inputs, max_rank_sum, depth, n_threads,
                              fcp::read_env().stager_zero_copy ? FCP_STAGER_ZERO_COPY : FCP_STAGER_DEFAULT, out); // FCP_STAGER_ZERO_COPY
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
  const fcp::Env env = fcp::read_env(); // the stager's shipping switches, read once: here
  if (env.stager_copy >= 0) s->copy_kernel = env.stager_copy != 0;                          // FCP_STAGER_COPY=kernel|sdma
  if (env.stager_groups >= 0) s->groups = std::max(1, std::min(env.stager_groups, 16));     // FCP_STAGER_GROUPS; 1 = one copy per request
  s->groups_always = fcp::diag_on("stager_groups_always"); // test aid: groups for every request, not only a lone one
  s->stats = fcp::diag_on("stager_stats");                 // diagnostic: host-time breakdown on stderr
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
    static const bool no_pin = fcp::diag_on("stager_no_pin"); // tuning aid
    const bool pin = !no_pin && fcp::cpus_near_device(device, &near);
    s->pool = new fcp::PackPool(n_threads, pin ? &near : nullptr);
  }
  s->byte_off.resize(2 * (size_t)max_inputs + 2); // offsets [0, n], lengths [n + 1, 2n + 1] (stage_layout)
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

// Layout of the staged blob: where every input goes (byte_off[0..n): its offset, byte_off[n]: the blob's size,
// byte_off[n + 1 .. 2n + 1): its LENGTH there — callers hand in 2n + 2 entries), the op's `offsets` and `shapes` outputs.
// Inputs that are copied or narrowed lie back to back in input order from offset 0 — exactly ConcatInputsOp::Compute
// (concat_inputs_ops.cc:52-66), except that a narrowed int64 input occupies 4 bytes per element.  (r6) Inputs CONVERTED to row
// offsets (int32[rows + 1], one dim) follow behind ALL of them, in input order: columns of one concat group have the same
// row count, so their CSR arrays form one [columns, rows + 1] matrix in the blob and the ragged kernel's front finds a column's
// row ranges from its position alone — requested together with the column records instead of behind them
// (FcpLaunch::csr_reg, recognised per descriptor install; RAGGED staged 27.5 -> 26.7 us).  Where an input lies is the op's
// own `offsets` output: nothing downstream assumes input order.  max_rank_sum < 0: no limit.
int stage_layout(const fcp_host_tensor_t *inputs, int32_t n, const uint8_t *modes, const int64_t *mode_args, int64_t capacity,
                 int32_t max_rank_sum, int64_t *byte_off, int32_t *offsets, int32_t *shapes, int32_t *rank_sum_out) {
  int64_t size = 0;
  int32_t rank_sum = 0;
  int64_t *byte_len = byte_off + n + 1;
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
    if (mode == FCP_STAGE_SEG_TO_CSR) {
      // sorted row ids [nnz] or SparseTensor indices [nnz, k] -> int32 offsets[rows + 1]: one dim in the shapes; placed by
      // the second pass below, behind every input of the first region
      if ((t.elem_size != 4 && t.elem_size != 8) || t.rank < 1 || t.rank > 2 || !mode_args || mode_args[i] < 0 ||
          mode_args[i] >= 0x7fffffff || (t.rank == 2 && t.dims[1] < 1))
        return fail(FCP_ERR_INVALID_ARGUMENT, "segment-id input to convert: int32 / int64 [nnz] or [nnz, k], with its number of rows");
      if (shapes) shapes[rank_sum] = (int32_t)(mode_args[i] + 1);
      rank_sum += 1;
      byte_off[i] = -1;
      byte_len[i] = (mode_args[i] + 1) * 4;
    } else {
      for (int32_t j = 0; j < t.rank; ++j)
        if (shapes) shapes[rank_sum + j] = (int32_t)t.dims[j];
      rank_sum += t.rank;
      byte_off[i] = size;
      byte_len[i] = ne * (mode == FCP_STAGE_NARROW_I64 ? 4 : t.elem_size);
      size += byte_len[i];
      if (offsets) offsets[i] = (int32_t)byte_off[i];
    }
    if (capacity >= 0 && size > capacity) return fail(FCP_ERR_INVALID_ARGUMENT, "request larger than the blob / stager capacity");
    // The reference keeps offsets in int32 (:52-60); refuse what it would overflow.
    if (size > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "blob larger than 2^31 bytes (int32 offsets)");
    if (ne && !t.data) return fail(FCP_ERR_INVALID_ARGUMENT, "null tensor data");
  }
  bool padded = false;
  for (int32_t i = 0; i < n; ++i) {  // second region: the converted inputs, in input order
    if (byte_off[i] >= 0) continue;
    if (!padded) size = (size + 3) & ~(int64_t)3; // (the matrix of row offsets starts on a 4-byte boundary whatever was copied before it)
    padded = true;
    byte_off[i] = size;
    if (offsets) offsets[i] = (int32_t)size;
    size += byte_len[i];
    if (capacity >= 0 && size > capacity) return fail(FCP_ERR_INVALID_ARGUMENT, "request larger than the blob / stager capacity");
    if (size > 0x7fffffff) return fail(FCP_ERR_UNSUPPORTED, "blob larger than 2^31 bytes (int32 offsets)");
  }
  byte_off[n] = size;
  if (rank_sum_out) *rank_sum_out = rank_sum;
  return FCP_OK;
}

// A request is a thousand or two SMALL tensors (RAGGED: 11 KB of ids and 22 KB of indices per column), each somewhere else
// in memory: the hardware prefetcher starts over at every one of them.  While tensor i is packed the head of tensor
// i + 1 is requested (FCP_DIAG=pack_prefetch_bytes=N, default 2 KiB: tuning aid; 0 = off).
inline void prefetch_head(const fcp_host_tensor_t &t) {
  static const int64_t bytes = fcp::diag_ll("pack_prefetch_bytes", 2048);
  if (!t.data) return;
  int64_t n = t.elem_size;
  for (int32_t j = 0; j < t.rank; ++j) n *= t.dims[j];
  if (n > bytes) n = bytes;
  static const int hint = (int)fcp::diag_ll("pack_prefetch_hint", 0); // tuning aid: 0 = non-temporal, 1 = every cache level, 2 = L2 and up
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
    const int v = (int)fcp::diag_ll("pack_chunks_per_thread", 0);
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
  // FCP_DIAG=pack_trace (diagnostic): when did every chunk of a call start and end, and on which thread — printed for every 128th call
  static const bool trace = fcp::diag_on("pack_trace");
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
      if (!stage_pack_one(inputs[lo], modes ? modes[lo] : FCP_STAGE_COPY, mode_args ? mode_args[lo] : 0, dst + bo[lo], bo[n + 1 + lo]))
        refused.store(1, std::memory_order_relaxed);
    }
    if (ng > 0) group_left[c * ng / chunks].fetch_sub(1, std::memory_order_release);
  };
  auto ship = [&](int g) { // (all chunks of group g are packed)
    int c0 = 0;
    while (c0 < chunks && c0 * ng / chunks < g) ++c0;
    int c1 = c0;
    while (c1 < chunks && c1 * ng / chunks == g) ++c1;
    // the group's inputs [i0, i1) occupy one range of the copied / narrowed region and one of the row-offset region behind it
    // (stage_layout): each is contiguous in input order
    const int i0 = chunk_lo[(size_t)c0], i1 = chunk_lo[(size_t)c1];
    int64_t lo_a = -1, hi_a = -1, lo_b = -1, hi_b = -1;
    for (int i = i0; i < i1; ++i) {
      const bool conv = modes && modes[i] == FCP_STAGE_SEG_TO_CSR;
      int64_t &lo = conv ? lo_b : lo_a, &hi = conv ? hi_b : hi_a;
      if (lo < 0) lo = bo[i];
      hi = bo[i] + bo[n + 1 + i];
    }
    if (hi_a > lo_a) groups->done(groups->ctx, g, lo_a, hi_a);
    if (hi_b > lo_b) groups->done(groups->ctx, g, lo_b, hi_b);
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
  static const bool early_wake = !fcp::diag_on("stager_no_early_wake"); // tuning aid
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
  static const bool no_fallback = fcp::diag_on("stager_no_fallback");
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
  // In group mode the calling thread ships instead of packing: with a small pool (n_threads 2 / 3 = one / two workers) that
  // takes a third to a half of the packers away and a lone request gets SLOWER, the opposite of what groups are for
  // (ADVICE r05): groups only from five threads up (four workers + the shipping caller).
  const bool pool_can_spare_the_caller = s->n_threads >= 5 || s->groups_always;
  PackGroups pg{lone && pool_can_spare_the_caller ? s->groups : 1, nullptr, &ship};
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
  std::vector<int64_t> bo(2 * (size_t)n + 2);
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
  std::vector<int64_t> bo(2 * (size_t)n + 2);
  const int rc = stage_layout(inputs, n, modes, mode_args, blob_capacity, -1, bo.data(), offsets, shapes, nullptr);
  if (rc) return rc;
  if (bo[n] > 0 && !blob) return fail(FCP_ERR_INVALID_ARGUMENT, "blob too small");
  bool ok = true;
  for (int32_t i = 0; i < n; ++i)
    ok = stage_pack_one(inputs[i], modes ? modes[i] : FCP_STAGE_COPY, mode_args ? mode_args[i] : 0, static_cast<char *>(blob) + bo[i],
                        bo[n + 1 + i]) && ok;
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
  std::vector<int64_t> bo(2 * (size_t)n + 2), in_off((size_t)n + 1);
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

