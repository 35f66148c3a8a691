// pack_pool.h — worker pool of the request stager (fcp_stager_*, include/fcp_hip.h).
// Plain C++ (no HIP) so that tests/native/pack_pool_stress.cc can run it under
// ThreadSanitizer on the CPU.
#pragma once
#include "fcp_env.h"
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <thread>
#include <type_traits>
#include <vector>

namespace fcp {

// Persistent worker pool: parallel_for over [0, n_chunks).  Requests arrive every few tens of
// microseconds and a futex wake-up costs about as much as packing one request, so the workers
// are woken EARLY — `expect()` at the top of a staging call, several microseconds before the job
// is published — and spin only from then until the job arrives (bounded), never between requests:
// workers that spun through the gaps (round 2: ~1-2 ms after every job) kept the HIP runtime's own
// threads off their CPUs, and the hipMemcpyAsync that follows the pack blocked for ~50 us
// (profiles/r03_pcie_staging_api_call_timers.txt).  Chunk claims carry the job's epoch and chunk
// count together with the next chunk index in one 64-bit word and advance by compare-exchange, so
// a worker that is late leaving job e can neither run nor skip a chunk of job e+1.
class PackPool {
public:
  // `affinity` (optional): CPUs the workers may run on
  explicit PackPool(int n_threads, const cpu_set_t *affinity = nullptr) {
    if (affinity) {
      affinity_ = *affinity;
      pinned_ = true;
    }
    for (int t = 1; t < n_threads; ++t)
      workers_.emplace_back([this] {
        if (pinned_) (void)pthread_setaffinity_np(pthread_self(), sizeof(affinity_), &affinity_);
        loop();
      });
  }
  ~PackPool() {
    stop_.store(true, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu_);
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
  }
  // A job is about to be published: wake the workers now, so that their wake-up latency overlaps whatever
  // the caller still does before run().  Harmless if no job follows (they spin kSpins and sleep again).
  // Waking is a TREE: the caller wakes ONE sleeper (a futex wake costs the waker ~0.7 us per thread it wakes: 11 us for 15
  // workers, 19 us for 31 when this was notify_all — on the calling thread, before it packs), every worker that wakes
  // up wakes two more before it looks for work.
  void expect() {
    if (workers_.empty()) return;
    expected_.fetch_add(1, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu_);
    }
    wake();
  }
  // run() in two halves: start() publishes the job and returns at once — the workers pack while the caller does something
  // else (the stager: ships the groups of the request over PCIe as they complete) — finish() lets the caller help with what is
  // left and waits for the last chunk.  `fn` must stay alive until finish() has returned.  Returns false when the pool has
  // no workers (or the job a single chunk): nothing was started, the caller runs the chunks itself.
  template <typename F> bool start(int n_chunks, F &fn) {
    if (workers_.empty() || n_chunks <= 1) return false;
    if (n_chunks > kMaxChunks) n_chunks = kMaxChunks;
    call_ = [](void *p, int c) { (*static_cast<F *>(p))(c); };
    ctx_ = static_cast<void *>(&fn);
    const uint64_t e = epoch_.load(std::memory_order_relaxed) + 1;
    pending_.store(n_chunks, std::memory_order_relaxed);
    next_.store(pack(e, n_chunks, 0), std::memory_order_release);
    epoch_.store(e, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu_);
    }
    wake();
    return true;
  }
  void finish() {
    work(epoch_.load(std::memory_order_relaxed));
    while (pending_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
  }
  template <typename F> void run(int n_chunks, F &&fn) {
    if (workers_.empty() || n_chunks <= 1) {
      for (int c = 0; c < n_chunks; ++c) fn(c);
      return;
    }
    if (n_chunks > kMaxChunks) n_chunks = kMaxChunks; // never reached by the stager (<= 4 x threads)
    using Fn = typename std::remove_reference<F>::type;
    call_ = [](void *p, int c) { (*static_cast<Fn *>(p))(c); };
    ctx_ = const_cast<void *>(static_cast<const void *>(&fn));
    const uint64_t e = epoch_.load(std::memory_order_relaxed) + 1;
    pending_.store(n_chunks, std::memory_order_relaxed);
    // one word = job epoch | chunk count | next chunk: a claim can only succeed against the
    // job it was computed for; the release store publishes call_/ctx_ to every claimer
    next_.store(pack(e, n_chunks, 0), std::memory_order_release);
    epoch_.store(e, std::memory_order_release);
    {
      std::lock_guard<std::mutex> lk(mu_); // a worker between its predicate check and its wait holds mu_
    }
    wake(); // (sleepers left: the tree goes on from here; none: free)
    work(e); // the caller helps
    while (pending_.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();
  }

private:
  void wake() {
    if (kFanout <= 0) {
      cv_.notify_all();
    } else {
      for (int i = 0; i < kFanout; ++i) cv_.notify_one();
    }
  }
  // FCP_PACK_FANOUT (tuning aid): 0 = the caller wakes every sleeper itself (notify_all); k > 0 = the caller wakes k, every
  // woken worker k more
  const int kFanout = [] {
    return (int)fcp::diag_ll("pack_fanout", 2); // tuning aid
  }();
  static constexpr int kIdxBits = 20, kMaxChunks = (1 << kIdxBits) - 1;
  static uint64_t pack(uint64_t e, int n, int idx) {
    return (e << (2 * kIdxBits)) | ((uint64_t)n << kIdxBits) | (uint64_t)idx;
  }
  void work(uint64_t e) {
    const uint64_t tag = pack(e, 0, 0) >> (2 * kIdxBits);
    for (;;) {
      uint64_t cur = next_.load(std::memory_order_acquire);
      int idx;
      for (;;) {
        idx = (int)(cur & kMaxChunks);
        if ((cur >> (2 * kIdxBits)) != tag || idx >= (int)((cur >> kIdxBits) & kMaxChunks)) return;
        if (next_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
      }
      call_(ctx_, idx); // job e cannot end before this chunk is counted below
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  void loop() {
    uint64_t seen = 0, seen_expect = 0;
    for (;;) {
      uint64_t e;
      int spins = kSpins; // asleep until a job is announced (expect) or published (run)
      while ((e = epoch_.load(std::memory_order_acquire)) == seen) {
        if (stop_.load(std::memory_order_acquire)) return;
        if (++spins < kSpins) {
          __builtin_ia32_pause();
        } else {
          std::unique_lock<std::mutex> lk(mu_);
          cv_.wait(lk, [&] {
            return epoch_.load(std::memory_order_acquire) != seen || expected_.load(std::memory_order_acquire) != seen_expect ||
                   stop_.load(std::memory_order_acquire);
          });
          seen_expect = expected_.load(std::memory_order_acquire);
          lk.unlock();
          if (kFanout > 0)
            for (int i = 0; i < kFanout; ++i) cv_.notify_one(); // the wake-up tree: kFanout more sleepers each
          spins = 0; // announced: spin until the job is there, at most kSpins pauses
        }
      }
      seen = e;
      work(e);
    }
  }
  // how long a woken worker spins for the announced job: ~20-40 us of pause instructions, a few times the layout phase
  // that separates expect() from run() (FCP_DIAG=pack_spins=N: tuning aid)
  const int kSpins = [] {
    return (int)fcp::diag_ll("pack_spins", 1 << 10);
  }();
  cpu_set_t affinity_;
  bool pinned_ = false;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_;
  void (*call_)(void *, int) = nullptr;
  void *ctx_ = nullptr;
  std::atomic<uint64_t> next_{0}, epoch_{0}, expected_{0};
  std::atomic<int> pending_{0};
  std::atomic<bool> stop_{false};
};

} // namespace fcp
