// fcp_env.h — every environment variable libfcp_hip.so reads, in one place (round 6; rounds 1-5 had grown 45 getenv calls
// scattered over the request path).
//
//   * SHIPPING switches — the thirteen fields of fcp::Env below, documented in INTEGRATION.md section 8.  None is needed for
//     normal operation.  fcp::read_env() is called ONCE per object — when a plan (fcp_plan_create*), a request stager
//     (fcp_stager_create*) or the RCCL binding is created — and the values live in that object: nothing on the request path
//     reads the environment.
//   * Everything else — diagnostics, tuning aids of the measurement scripts, test hooks — is ONE variable,
//       FCP_DIAG="key[=value],key[=value],..."      e.g.  FCP_DIAG="lane_fault_us=120,dyn_general"
//     looked up with fcp::diag("key") at the (rare or cached) place that wants it.  A bare key has the value "1".  The
//     keys are listed in INTEGRATION.md section 8; a deployment never sets FCP_DIAG.
//
// Header-only (fcp_pack.cc, fcp_graph.cc and pack_pool.h are plain C++ translation units without the HIP headers).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>

namespace fcp {

struct Env {
  // ---- request path (plan) ------------------------------------------------------------------------------------------------
  int64_t store_through_bytes = (int64_t)32 << 20; // FCP_STORE_THROUGH_BYTES: output size from which stores are write-through (`sc1 nt`)
  bool dyn_upload_kernel = false;                  // FCP_DYN_UPLOAD=kernel: column records uploaded by a kernel, not by host stores through the BAR
  bool seg_prepass = false;                        // FCP_SEG_PREPASS=1: segment-id columns always take the segment-offset pre-pass
  int64_t seg_search_max_pairs = 32768;            // FCP_SEG_SEARCH_MAX_PAIRS: up to how many (column, row) pairs the blocks search themselves
  // ---- private streams (experimental) -------------------------------------------------------------------------------------
  int64_t private_min_work_bytes = -1;             // FCP_PRIVATE_MIN_WORK_BYTES: requests below it stay on the caller's stream (-1: default 48 MiB)
  int32_t private_verify_budget_ms = -1;           // FCP_PRIVATE_VERIFY_BUDGET_MS: mapping search inside a request that had no warm-up (-1: 120)
  int32_t lane_supervise = -1;                     // FCP_LANE_SUPERVISE=0: no run-time supervisor (-1: on)
  int32_t lane_supervise_period = -1;              // FCP_LANE_SUPERVISE_PERIOD: the largest gap between evaluations, requests (-1: 8192)
  double lane_keep_ratio = -1.0;                   // FCP_LANE_KEEP_RATIO (-1: 0.97)
  // ---- request stager ---------------------------------------------------------------------------------------------------
  int32_t stager_copy = -1;                        // FCP_STAGER_COPY=kernel|sdma (-1: kernel)
  bool stager_zero_copy = false;                   // FCP_STAGER_ZERO_COPY=1: fcp_stager_create (without _ex) makes zero-copy stagers
  int32_t stager_groups = -1;                      // FCP_STAGER_GROUPS: groups a lone request is packed and shipped in (-1: 4)
  // ---- multi-GPU --------------------------------------------------------------------------------------------------------
  std::string rccl_path;                           // FCP_RCCL_PATH: which RCCL to dlopen (empty: librccl.so.1)
};

inline Env read_env() {
  Env e;
  if (const char *v = std::getenv("FCP_STORE_THROUGH_BYTES")) e.store_through_bytes = std::atoll(v);
  if (const char *v = std::getenv("FCP_DYN_UPLOAD")) e.dyn_upload_kernel = !std::strcmp(v, "kernel");
  if (std::getenv("FCP_SEG_PREPASS")) e.seg_prepass = true;
  if (const char *v = std::getenv("FCP_SEG_SEARCH_MAX_PAIRS")) e.seg_search_max_pairs = std::atoll(v);
  if (const char *v = std::getenv("FCP_PRIVATE_MIN_WORK_BYTES")) e.private_min_work_bytes = std::atoll(v);
  if (const char *v = std::getenv("FCP_PRIVATE_VERIFY_BUDGET_MS")) e.private_verify_budget_ms = std::atoi(v);
  if (const char *v = std::getenv("FCP_LANE_SUPERVISE")) e.lane_supervise = std::atoi(v) != 0;
  if (const char *v = std::getenv("FCP_LANE_SUPERVISE_PERIOD")) e.lane_supervise_period = std::atoi(v);
  if (const char *v = std::getenv("FCP_LANE_KEEP_RATIO")) e.lane_keep_ratio = std::atof(v);
  if (const char *v = std::getenv("FCP_STAGER_COPY")) e.stager_copy = std::strcmp(v, "sdma") != 0;
  if (std::getenv("FCP_STAGER_ZERO_COPY")) e.stager_zero_copy = true;
  if (const char *v = std::getenv("FCP_STAGER_GROUPS")) e.stager_groups = std::atoi(v);
  if (const char *v = std::getenv("FCP_RCCL_PATH")) e.rccl_path = v;
  return e;
}

// The value of `key` in FCP_DIAG, or nullptr.  The returned string lives in a thread-local buffer until this thread's
// next call.  Not for the request path: callers are plan / stager creation, rare paths (a descriptor miss, a verification)
// or function-local statics.
inline const char *diag(const char *key) {
  const char *list = std::getenv("FCP_DIAG");
  if (!list || !*list) return nullptr;
  thread_local char buf[96];
  const size_t klen = std::strlen(key);
  for (const char *p = list; *p;) {
    const char *e = std::strchr(p, ',');
    const size_t n = e ? (size_t)(e - p) : std::strlen(p);
    if (n >= klen && !std::strncmp(p, key, klen) && (n == klen || p[klen] == '=')) {
      if (n == klen) {
        buf[0] = '1';
        buf[1] = 0;
      } else {
        const size_t v = n - klen - 1 < sizeof(buf) - 1 ? n - klen - 1 : sizeof(buf) - 1;
        std::memcpy(buf, p + klen + 1, v);
        buf[v] = 0;
      }
      return buf;
    }
    p += n + (e ? 1 : 0);
  }
  return nullptr;
}
inline bool diag_on(const char *key) { return diag(key) != nullptr; }
inline long long diag_ll(const char *key, long long dflt) {
  const char *v = diag(key);
  return v ? std::atoll(v) : dflt;
}

} // namespace fcp
