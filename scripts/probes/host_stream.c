/* host_stream.c — round 6: what the GPU box's HOST memory does (context for bench.py's cpu_baseline, which gets slower beyond
 * 16 workers): OpenMP copy / triad over arrays far larger than the caches, and per-thread PRIVATE 12-MB buffers rewritten in
 * a loop (the shape of the baseline's per-worker output), at several thread counts.
 *   gcc -O2 -fopenmp scripts/probes/host_stream.c -o build/host_stream && ./build/host_stream */
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(void) {
  const size_t n = (size_t)1 << 29; /* 4 GiB per array of doubles */
  double *a = malloc(n * 8), *b = malloc(n * 8), *c = malloc(n * 8);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) { a[i] = 1.0; b[i] = 2.0; c[i] = 0.0; }
  int counts[] = {16, 32, 64, 128, 256};
  for (int k = 0; k < 5; ++k) {
    const int t = counts[k];
    if (t > omp_get_max_threads()) break;
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static) num_threads(t)
      for (size_t i = 0; i < n; ++i) c[i] = a[i] + 3.0 * b[i];
      const double el = omp_get_wtime() - t0;
      if (el < best) best = el;
    }
    printf("triad, %3d threads: %7.1f GB/s (3 x 4 GiB)\n", t, 3.0 * n * 8 / best / 1e9);
    /* private buffers: every thread rewrites its own 12 MB, 40 times (row-strided 64-byte pieces, like the baseline's
     * fused output: 512 rows x 24 KB, one 64-byte piece per row and pass) */
    const size_t rows = 512, width = 24576, piece = 64;
    double t0 = 0, t1 = 0;
#pragma omp parallel num_threads(t)
    {
      char *buf = calloc(rows * width, 1);
      memset(buf, 1, rows * width);
#pragma omp barrier
#pragma omp master
      t0 = omp_get_wtime();
#pragma omp barrier
      for (int pass = 0; pass < 40; ++pass)
        for (size_t off = 0; off < width; off += piece)
          for (size_t r = 0; r < rows; ++r) memset(buf + r * width + off, pass, piece);
#pragma omp barrier
#pragma omp master
      t1 = omp_get_wtime();
      free(buf);
    }
    printf("private 12-MB buffers rewritten in 64-byte row-strided pieces, %3d threads: %7.1f GB/s written in total\n", t,
           40.0 * rows * width * t / (t1 - t0) / 1e9);
  }
  return 0;
}
