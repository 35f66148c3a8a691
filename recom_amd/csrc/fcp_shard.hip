// fcp_shard.hip — the exchange step of the sharded path under the C ABI (include/fcp_hip.h, "multi-GPU").
//
// No reference counterpart: RECom is single-GPU.  BASELINE.json's north star
// shards the tables over the 8 GPUs of a node only when they exceed one GPU's 288 GB, with ONE
// exchange per request — an all-to-all over xGMI, partitioned along the batch:
//   row sharding     rank g holds rows {r : r % world == g} of every table and computes partial sums
//                    P_g[rows, width] of the whole batch; rank h receives P_g[rows_h, :] from every g
//                    (fcp_shard_exchange), adds the `world` slices in rank order (fcp_shard_finalize);
//   column sharding  rank g holds whole columns and produces the final block of its columns; rank h
//                    receives the rows rows_h of every rank's block (fcp_shard_exchange_columns) and
//                    puts them side by side (fcp_concat_outputs).
// xGMI is point to point (7 links per GPU): the exchange is a grouped ncclSend / ncclRecv to every
// peer at once, so all links carry traffic concurrently — not a ring, not a reduction tree.
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy the process already holds — torch
// bundles one — or ROCm's), so libfcp_hip.so loads on boxes and in processes that never shard.
// fcp_shard_step_* strings partial kernel -> exchange -> finalize / concat together on one stream
// from buffers the object owns: one native call per request, no host code between the three.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fcp_hip.h"
#include "fcp_env.h"

// failure reporting shared with fcp_plan.hip
int fcp_internal_fail(int code, const std::string &msg);
extern "C" int fcp_internal_process(fcp_plan_t *p, const fcp_process_args_t *a, fcp_process_result_t *r); // fcp_process.hip

namespace {

// ---- the few RCCL entry points, by their public C signatures (rccl.h) ---------------------------------
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == FCP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
enum { kNcclSuccess = 0, kNcclFloat32 = 7 };

struct Rccl {
  void *lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string why;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    static const std::string path = fcp::read_env().rccl_path; // FCP_RCCL_PATH, read once: here
    const char *names[] = {path.empty() ? nullptr : path.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      if (!n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
      r.why = dlerror();
    }
    if (!r.lib) return;
    auto sym = [&](const char *name) {
      void *p = dlsym(r.lib, name);
      if (!p) r.why = std::string("librccl has no ") + name;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return &r;
}

int need_rccl(Rccl **out) {
  Rccl *r = rccl();
  if (!r->lib) return fcp_internal_fail(FCP_ERR_UNSUPPORTED, "RCCL is not available (librccl.so.1): " + r->why);
  *out = r;
  return FCP_OK;
}

int nccl_fail(Rccl *r, const char *what, int code) {
  return fcp_internal_fail(FCP_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(code) : "RCCL error"));
}

struct DeviceScope {
  int prev = -1;
  bool changed = false;
  int enter(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) return fcp_internal_fail(FCP_ERR_NO_DEVICE, "hipGetDevice failed");
    if (prev != dev) {
      if (hipSetDevice(dev) != hipSuccess) return fcp_internal_fail(FCP_ERR_NO_DEVICE, "hipSetDevice failed");
      changed = true;
    }
    return FCP_OK;
  }
  ~DeviceScope() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// contiguous split of the batch: the first rows % world ranks get one extra row (recom_amd/shard.py::batch_slices)
void batch_slice(int64_t rows, int world, int rank, int64_t *begin, int64_t *count) {
  const int64_t base = rows / world, extra = rows % world;
  *count = base + (rank < extra ? 1 : 0);
  *begin = rank * base + (rank < extra ? rank : extra);
}

} // namespace

struct fcp_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

extern "C" {

int fcp_comm_unique_id(uint8_t *id) {
  if (!id) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null id");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc) return rc;
  ncclUniqueId u;
  const int e = r->GetUniqueId(&u);
  if (e != kNcclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
  std::memcpy(id, u.internal, FCP_COMM_ID_BYTES);
  return FCP_OK;
}

int fcp_comm_create(const uint8_t *id, int32_t rank, int32_t world, int32_t device, fcp_comm_t **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "bad communicator arguments");
  *out = nullptr;
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc) return rc;
  DeviceScope scope;
  rc = scope.enter(device);
  if (rc) return rc;
  ncclUniqueId u;
  std::memcpy(u.internal, id, FCP_COMM_ID_BYTES);
  fcp_comm *c = new fcp_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int e = r->CommInitRank(&c->comm, world, u, rank);
  if (e != kNcclSuccess) {
    delete c;
    return nccl_fail(r, "ncclCommInitRank", e);
  }
  *out = c;
  return FCP_OK;
}

int fcp_comm_destroy(fcp_comm_t *c) {
  if (!c) return FCP_OK;
  Rccl *r = rccl();
  if (r->lib && c->comm) {
    DeviceScope scope;
    (void)scope.enter(c->device);
    (void)r->CommDestroy(c->comm);
  }
  delete c;
  return FCP_OK;
}

int fcp_comm_rank(const fcp_comm_t *c, int32_t *rank, int32_t *world) {
  if (!c) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null communicator");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return FCP_OK;
}

int fcp_shard_batch_slice(int64_t rows, int32_t world, int32_t rank, int64_t *begin, int64_t *count) {
  if (rows < 0 || world < 1 || rank < 0 || rank >= world || !begin || !count)
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "bad batch slice arguments");
  batch_slice(rows, world, rank, begin, count);
  return FCP_OK;
}

// Row sharding: every rank holds partial [rows, width]; rank h ends up with slices [world, count_h, width].
int fcp_shard_exchange(fcp_comm_t *c, const void *partial, int64_t rows, int64_t width, void *slices, int64_t *row_begin,
                       int64_t *row_count, void *stream_) {
  if (!c || rows < 0 || width < 0 || (rows * width > 0 && (!partial || !slices)))
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "bad exchange arguments");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc) return rc;
  DeviceScope scope;
  rc = scope.enter(c->device);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int64_t my_begin, my_count;
  batch_slice(rows, c->world, c->rank, &my_begin, &my_count);
  if (row_begin) *row_begin = my_begin;
  if (row_count) *row_count = my_count;
  const float *src = static_cast<const float *>(partial);
  float *dst = static_cast<float *>(slices);
  int e = r->GroupStart();
  if (e != kNcclSuccess) return nccl_fail(r, "ncclGroupStart", e);
  for (int peer = 0; peer < c->world && e == kNcclSuccess; ++peer) {
    int64_t pb, pc;
    batch_slice(rows, c->world, peer, &pb, &pc);
    // my rows of the peer's slice go to the peer; the peer's rows of my slice come here, in rank order
    if (pc * width > 0) e = r->Send(src + pb * width, (size_t)(pc * width), kNcclFloat32, peer, c->comm, stream);
    if (e == kNcclSuccess && my_count * width > 0)
      e = r->Recv(dst + (int64_t)peer * my_count * width, (size_t)(my_count * width), kNcclFloat32, peer, c->comm, stream);
  }
  const int e2 = r->GroupEnd();
  if (e != kNcclSuccess) return nccl_fail(r, "ncclSend / ncclRecv", e);
  if (e2 != kNcclSuccess) return nccl_fail(r, "ncclGroupEnd", e2);
  return FCP_OK;
}

// Column sharding: rank g holds block [rows, widths[g]]; rank h ends up with, back to back, the blocks
// [count_h, widths[g]] of g = 0..world-1 (fcp_concat_outputs then puts them side by side).
int fcp_shard_exchange_columns(fcp_comm_t *c, const void *block, int64_t rows, const int32_t *widths, void *recv,
                               int64_t *row_begin, int64_t *row_count, void *stream_) {
  if (!c || rows < 0 || !widths) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "bad exchange arguments");
  Rccl *r = nullptr;
  int rc = need_rccl(&r);
  if (rc) return rc;
  DeviceScope scope;
  rc = scope.enter(c->device);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int64_t my_begin, my_count;
  batch_slice(rows, c->world, c->rank, &my_begin, &my_count);
  if (row_begin) *row_begin = my_begin;
  if (row_count) *row_count = my_count;
  const int64_t my_width = widths[c->rank];
  const float *src = static_cast<const float *>(block);
  float *dst = static_cast<float *>(recv);
  if (rows * my_width > 0 && !src) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null block");
  // every argument is checked BEFORE the group opens: a rank that left between ncclGroupStart and ncclGroupEnd with
  // some sends queued would leave its peers waiting for the rest
  int64_t total_recv = 0;
  for (int peer = 0; peer < c->world; ++peer) {
    if (widths[peer] < 0) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "negative block width");
    total_recv += my_count * widths[peer];
  }
  if (total_recv > 0 && !dst) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null receive buffer");
  int e = r->GroupStart();
  if (e != kNcclSuccess) return nccl_fail(r, "ncclGroupStart", e);
  int64_t at = 0;
  for (int peer = 0; peer < c->world && e == kNcclSuccess; ++peer) {
    int64_t pb, pc;
    batch_slice(rows, c->world, peer, &pb, &pc);
    if (pc * my_width > 0) e = r->Send(src + pb * my_width, (size_t)(pc * my_width), kNcclFloat32, peer, c->comm, stream);
    const int64_t n = my_count * widths[peer];
    if (e == kNcclSuccess && n > 0) e = r->Recv(dst + at, (size_t)n, kNcclFloat32, peer, c->comm, stream);
    at += n;
  }
  const int e2 = r->GroupEnd();
  if (e != kNcclSuccess) return nccl_fail(r, "ncclSend / ncclRecv", e);
  if (e2 != kNcclSuccess) return nccl_fail(r, "ncclGroupEnd", e2);
  return FCP_OK;
}

} // extern "C"

// ---- one native call per sharded request -----------------------------------------------------------------
struct fcp_shard_step {
  fcp_plan_t *plan = nullptr;
  fcp_comm_t *comm = nullptr;
  int mode = FCP_PLACE_ROW_SHARD, group = 0, device = 0;
  int64_t max_rows = 0, width = 0;     // row mode: the group's full width; column mode: sum of widths
  std::vector<int32_t> widths;         // column mode: width of every rank's block
  std::vector<int32_t> col_offsets;    // column mode: where every rank's block starts in the output row
  // device buffers, a ring of `depth` each so that the host may run ahead of the GPU
  int depth = 3;
  size_t next = 0;
  int64_t arena_bytes = 0;
  std::vector<void *> arenas, recvs, outs, temps;
  int64_t temp_bytes = 0;
  // a ring entry is reused `depth` calls later: on the same stream that is ordered by itself, on another stream the
  // new call waits for the entry's previous use (event recorded at the end of every run)
  std::vector<hipEvent_t> used;
  std::vector<void *> used_on;
  std::mutex mu; // held for a whole fcp_shard_step_run: ring cursor, ring entry state, order of the collectives
};

namespace {
struct OneShot {
  void *p;
  size_t cap;
};
void *oneshot_alloc(void *ctx, size_t bytes) {
  OneShot *o = static_cast<OneShot *>(ctx);
  return bytes <= o->cap ? o->p : nullptr;
}
} // namespace

extern "C" {

int fcp_shard_step_create(fcp_plan_t *plan, fcp_comm_t *comm, int32_t mode, int32_t group, int64_t max_rows,
                          int64_t max_arena_bytes, const int32_t *col_widths, fcp_shard_step_t **out) {
  if (!plan || !comm || !out || max_rows <= 0 || max_arena_bytes <= 0 || (mode != FCP_PLACE_ROW_SHARD && mode != FCP_PLACE_COLUMN_SHARD))
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "bad shard step arguments");
  if (mode == FCP_PLACE_COLUMN_SHARD && !col_widths) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "column sharding needs the block widths");
  *out = nullptr;
  int32_t plan_width = 0;
  int rc = fcp_plan_group_width(plan, group, &plan_width);
  if (rc) return rc;
  fcp_shard_step *s = new fcp_shard_step();
  s->plan = plan;
  s->comm = comm;
  s->mode = mode;
  s->group = group;
  s->device = comm->device;
  s->max_rows = max_rows;
  s->arena_bytes = max_arena_bytes;
  if (mode == FCP_PLACE_COLUMN_SHARD) {
    int32_t off = 0;
    for (int g = 0; g < comm->world; ++g) {
      s->widths.push_back(col_widths[g]);
      s->col_offsets.push_back(off);
      off += col_widths[g];
    }
    s->width = off;
    if (col_widths[comm->rank] != plan_width) {
      delete s;
      return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "this rank's block width is not the plan's group width");
    }
  } else {
    s->width = plan_width;
  }
  DeviceScope scope;
  rc = scope.enter(s->device);
  if (rc) {
    delete s;
    return rc;
  }
  int64_t b, my_max;
  batch_slice(max_rows, comm->world, 0, &b, &my_max); // rank 0 holds the largest slice
  const size_t recv_bytes = (size_t)(mode == FCP_PLACE_ROW_SHARD ? comm->world * my_max * s->width : my_max * s->width) * 4;
  const size_t out_bytes = (size_t)(my_max * s->width) * 4;
  s->temp_bytes = max_arena_bytes; // finalize's CSR scratch (mean columns with segment ids) is part of what an arena holds
  for (int i = 0; i < s->depth; ++i) {
    void *a = nullptr, *rv = nullptr, *o = nullptr, *t = nullptr;
    if (hipMalloc(&a, (size_t)max_arena_bytes) != hipSuccess || hipMalloc(&rv, recv_bytes ? recv_bytes : 16) != hipSuccess ||
        hipMalloc(&o, out_bytes ? out_bytes : 16) != hipSuccess || hipMalloc(&t, (size_t)s->temp_bytes) != hipSuccess) {
      if (a) (void)hipFree(a);
      if (rv) (void)hipFree(rv);
      if (o) (void)hipFree(o);
      if (t) (void)hipFree(t);
      fcp_shard_step_destroy(s);
      return fcp_internal_fail(FCP_ERR_ALLOC, "shard step buffers");
    }
    s->arenas.push_back(a);
    s->recvs.push_back(rv);
    s->outs.push_back(o);
    s->temps.push_back(t);
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      fcp_shard_step_destroy(s);
      return fcp_internal_fail(FCP_ERR_HIP, "shard step event");
    }
    s->used.push_back(ev);
    s->used_on.push_back(nullptr);
  }
  *out = s;
  return FCP_OK;
}

int fcp_shard_step_destroy(fcp_shard_step_t *s) {
  if (!s) return FCP_OK;
  DeviceScope scope;
  (void)scope.enter(s->device);
  (void)hipDeviceSynchronize();
  for (auto *v : {&s->arenas, &s->recvs, &s->outs, &s->temps})
    for (void *p : *v) (void)hipFree(p);
  for (hipEvent_t ev : s->used) (void)hipEventDestroy(ev);
  delete s;
  return FCP_OK;
}

// partial kernel -> exchange -> finalize (row mode) / concat (column mode), all enqueued on args->stream.
// *out: device [row_count, width] of this rank's batch slice, valid until `depth` further calls.
int fcp_shard_step_run(fcp_shard_step_t *s, const fcp_process_args_t *args, void **out, int64_t *row_begin, int64_t *row_count) {
  if (!s || !args || !out) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  // One call at a time per step object: the ring entry's bookkeeping (used / used_on) belongs to the call that holds it,
  // and the collective inside must be issued in the same order on every rank anyway (RCCL: one operation at a time per
  // communicator) — host threads that share a step serialise here, their streams still overlap on the GPU.
  std::lock_guard<std::mutex> lock(s->mu);
  const size_t k = s->next;
  s->next = (s->next + 1) % (size_t)s->depth;
  DeviceScope scope;
  int rc = scope.enter(s->device);
  if (rc) return rc;
  void *const stream_key = args->stream ? args->stream : reinterpret_cast<void *>(1); // the null stream is a stream too
  if (s->used_on[k] && s->used_on[k] != stream_key &&
      hipStreamWaitEvent(static_cast<hipStream_t>(args->stream), s->used[k], 0) != hipSuccess)
    return fcp_internal_fail(FCP_ERR_HIP, "hipStreamWaitEvent (shard step ring)");
  OneShot arena{s->arenas[k], (size_t)s->arena_bytes}, temp{s->temps[k], (size_t)s->temp_bytes};
  fcp_process_args_t a = *args;
  a.malloc_buff = oneshot_alloc;
  a.malloc_buff_ctx = &arena;
  a.malloc_temp = oneshot_alloc;
  a.malloc_temp_ctx = &temp;
  void *group_ptr[FCP_MAX_GROUPS_ABI] = {nullptr};
  int32_t group_shape[2 * FCP_MAX_GROUPS_ABI] = {0};
  fcp_process_result_t res;
  std::memset(&res, 0, sizeof(res));
  res.group_ptrs = group_ptr;
  res.group_shapes = group_shape;
  rc = fcp_internal_process(s->plan, &a, &res); // on a.stream itself: the exchange follows there
  if (rc) return rc;
  const int64_t rows = group_shape[2 * s->group];
  if (rows > s->max_rows) return fcp_internal_fail(FCP_ERR_SHAPE_MISMATCH, "more rows than the shard step was created for");
  int64_t begin = 0, count = 0;
  if (s->mode == FCP_PLACE_ROW_SHARD) {
    rc = fcp_shard_exchange(s->comm, group_ptr[s->group], rows, s->width, s->recvs[k], &begin, &count, a.stream);
    if (rc) return rc;
    rc = fcp_shard_finalize(s->plan, &a, s->group, s->recvs[k], s->comm->world, begin, count, s->outs[k], a.stream);
    if (rc) return rc;
  } else {
    rc = fcp_shard_exchange_columns(s->comm, group_ptr[s->group], rows, s->widths.data(), s->recvs[k], &begin, &count, a.stream);
    if (rc) return rc;
    std::vector<const void *> parts(s->comm->world);
    int64_t at = 0;
    for (int g = 0; g < s->comm->world; ++g) {
      parts[g] = static_cast<const float *>(s->recvs[k]) + at;
      at += count * s->widths[g];
    }
    rc = fcp_concat_outputs_scatter(parts.data(), s->widths.data(), s->col_offsets.data(), s->comm->world, count, (int32_t)s->width,
                                    s->outs[k], a.stream);
    if (rc) return rc;
  }
  if (hipEventRecord(s->used[k], static_cast<hipStream_t>(a.stream)) != hipSuccess)
    return fcp_internal_fail(FCP_ERR_HIP, "hipEventRecord (shard step ring)");
  s->used_on[k] = stream_key;
  *out = s->outs[k];
  if (row_begin) *row_begin = begin;
  if (row_count) *row_count = count;
  return FCP_OK;
}

} // extern "C"
