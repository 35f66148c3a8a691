// pack_probe.cc — host-side probe (no GPU): SparseTensor indices [nnz, 2] int64 -> row offsets, the loop the staged
// Addons>ConcatInputs spends its time in, in three forms on cold data, next to memcpy and a read-only pass of the same
// bytes.   g++ -O3 -std=c++17 scripts/probes/pack_probe.cc -o /tmp/pack_probe && /tmp/pack_probe
#include <immintrin.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

static void enter_rows(int32_t from, int32_t to, int32_t pos, int32_t *out) { for (int32_t k = from + 1; k <= to; ++k) out[k] = pos; }

__attribute__((target("avx2"))) static void boundaries_avx2(const int32_t *t, int n, int32_t base, int32_t *out) {
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256i cur = _mm256_loadu_si256((const __m256i *)(t + i)), prv = _mm256_loadu_si256((const __m256i *)(t + i - 1));
    unsigned m = ~(unsigned)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(cur, prv))) & 0xffu;
    while (m) { const int b = __builtin_ctz(m); m &= m - 1; enter_rows(t[i + b - 1], t[i + b], base + i + b, out); }
  }
  for (; i < n; ++i) if (t[i] != t[i - 1]) enter_rows(t[i - 1], t[i], base + i, out);
}

template <int VARIANT> __attribute__((target("avx2"))) static void seg_to_csr(const int64_t *p, int64_t nnz, int64_t rows, int32_t *out) {
  constexpr int kBlock = 2048;
  int32_t buf[kBlock + 8];
  int32_t *tmp = buf + 8;
  const int32_t hi = (int32_t)rows;
  if (VARIANT == 0) memset(out, 0, 4 * (rows + 1));
  int32_t dummy = 0, cur = INT32_MIN, run = 0;
  tmp[-1] = -1;
  for (int64_t i0 = 0; i0 < nnz; i0 += kBlock) {
    const int n = (int)(nnz - i0 < kBlock ? nnz - i0 : kBlock);
    const int64_t *q = p + 2 * i0;
    for (int i = 0; i < n; ++i) { const int64_t r = q[2 * i]; tmp[i] = (int32_t)(r < -1 ? -1 : (r > hi ? hi : r)); }
    if (VARIANT == 0) {
      for (int i = 0; i < n; ++i) {
        const int32_t r = tmp[i];
        run = (run & -(int32_t)(r == cur)) + 1;
        cur = r;
        const uintptr_t in = (uintptr_t)0 - (uintptr_t)(r < hi);
        int32_t *dst = (int32_t *)(((uintptr_t)(out + (r + 1)) & in) | ((uintptr_t)&dummy & ~in));
        *dst = run;
      }
    } else if (VARIANT == 1) {
      for (int i = 0; i < n; ++i) if (tmp[i] != tmp[i - 1]) enter_rows(tmp[i - 1], tmp[i], (int32_t)i0 + i, out);
    } else {
      boundaries_avx2(tmp, n, (int32_t)i0, out);
    }
    tmp[-1] = tmp[n - 1];
  }
  if (VARIANT == 0) { int32_t acc = 0; for (int64_t r = 0; r <= rows; ++r) { acc += out[r]; out[r] = acc; } }
  else enter_rows(tmp[-1], hi, (int32_t)nnz, out);
}

int main(int argc, char **argv) {
  const int cols = 512, rows = 256, R = 8, maxlen = argc > 1 ? atoi(argv[1]) : 10;
  std::vector<std::vector<std::vector<int64_t>>> idx(R, std::vector<std::vector<int64_t>>(cols));
  srand(1);
  for (int q = 0; q < R; ++q) for (int c = 0; c < cols; ++c)
    for (int r = 0; r < rows; ++r) { int l = rand() % (maxlen + 1); for (int k = 0; k < l; ++k) { idx[q][c].push_back(r); idx[q][c].push_back(k); } }
  std::vector<int32_t> out(rows + 1), ref(rows + 1);
  size_t total = 0;
  for (int q = 0; q < R; ++q) for (int c = 0; c < cols; ++c) total += idx[q][c].size() * 8;
  std::vector<char> big(total + 64);
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int rep = 0; rep < 3; ++rep) {
    double t[3];
    for (int v = 0; v < 3; ++v) {
      auto t0 = std::chrono::steady_clock::now();
      for (int q = 0; q < R; ++q) for (int c = 0; c < cols; ++c) {
        const int64_t *p = idx[q][c].data(); const int64_t n = idx[q][c].size() / 2;
        if (v == 0) seg_to_csr<0>(p, n, rows, out.data()); else if (v == 1) seg_to_csr<1>(p, n, rows, out.data()); else seg_to_csr<2>(p, n, rows, out.data());
        if (v == 0) ref = out; else if (q == R - 1 && c == cols - 1 && ref != out) printf("MISMATCH variant %d\n", v);
      }
      t[v] = us(t0, std::chrono::steady_clock::now()) / R;
    }
    auto t1 = std::chrono::steady_clock::now();
    int64_t sum = 0;
    for (int q = 0; q < R; ++q) for (int c = 0; c < cols; ++c) { const int64_t *p = idx[q][c].data(); size_t n = idx[q][c].size() / 2; for (size_t i = 0; i < n; ++i) sum += p[2 * i]; }
    auto t2 = std::chrono::steady_clock::now();
    size_t nb = 0;
    for (int q = 0; q < R; ++q) for (int c = 0; c < cols; ++c) { memcpy(big.data() + nb, idx[q][c].data(), idx[q][c].size() * 8); nb += idx[q][c].size() * 8; }
    auto t3 = std::chrono::steady_clock::now();
    printf("per request (%.1f MB of indices): run lengths %.0f us | boundaries scalar %.0f us | boundaries avx2 %.0f us | read-only %.0f us | memcpy %.0f us  [%ld]\n",
           nb / 1e6 / R, t[0], t[1], t[2], us(t1, t2) / R, us(t2, t3) / R, (long)(sum & 1));
  }
}
