// Stress test of the stager's worker pool (recom_amd/csrc/pack_pool.h), built with
// -fsanitize=thread by tests/test_host.py.  Thousands of back-to-back jobs of varying
// size, each checked for "every chunk ran exactly once"; jobs are short so that workers
// are regularly still leaving job e when job e+1 is published.
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>

#include "pack_pool.h"

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 4;
  const int jobs = argc > 2 ? std::atoi(argv[2]) : 20000;
  fcp::PackPool pool(threads);
  std::vector<std::atomic<int>> hits(256);
  uint64_t rng = 88172645463325252ull;
  long total = 0;
  for (int j = 0; j < jobs; ++j) {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    const int n = 1 + (int)(rng % 200);
    for (int c = 0; c < n; ++c) hits[c].store(0, std::memory_order_relaxed);
    if (rng & 64) pool.expect();                                       // announced jobs, unannounced jobs ...
    if ((rng & 0x3f00) == 0) { pool.expect(); std::this_thread::sleep_for(std::chrono::microseconds(150)); } // ... and announcements nothing follows in time
    auto chunk = [&](int c) {
      hits[c].fetch_add(1, std::memory_order_relaxed);
      if ((c & 7) == 0) {
        volatile int sink = 0;
        for (int k = 0; k < 200; ++k) sink = sink + k; // uneven chunk cost
      }
    };
    if (rng & 128) { // the two-halves form the stager uses when it ships groups while the workers pack (round 5)
      if (pool.start(n, chunk)) {
        while (hits[0].load(std::memory_order_relaxed) == 0 && (rng & 256)) std::this_thread::yield(); // the caller watches ...
        pool.finish();                                                                                  // ... then helps and waits
        if (rng & 512) pool.finish(); // (a second call finds nothing to do)
      } else {
        for (int c = 0; c < n; ++c) chunk(c);
      }
    } else {
      pool.run(n, chunk);
    }
    for (int c = 0; c < n; ++c) {
      if (hits[c].load(std::memory_order_relaxed) != 1) {
        std::printf("job %d: chunk %d ran %d times\n", j, c, hits[c].load());
        return 1;
      }
    }
    total += n;
    if ((j & 1023) == 0) std::this_thread::sleep_for(std::chrono::microseconds(400)); // let workers fall asleep
  }
  std::printf("ok %ld chunks\n", total);
  return 0;
}
