// ref_bucketize_wrap.cc — C entry points around the REFERENCE's `Bucketize` template and its `alignmem` helper, whose text
// oracle/ref_extract.py takes from /root/reference/tensorflow_addons/graph_optimizers/cuda_emitter.cc:233-247 and :967-969
// into oracle/_ref/*.inc at build time (the text is never part of this repository).  Test infrastructure only.
#include <stdint.h>

#include <utility>

// the template is written for nvcc: the two qualifiers mean nothing to a host compiler
#define __device__
#define __forceinline__ inline
#include "bucketize_ref.inc"
// `alignmem` of the reference's generated host code (cuda_emitter.cc:967-969): plain C
#include "alignmem_ref.inc"

namespace {
constexpr int kMaxBoundaries = 1024; // instantiated for every count 1..kMaxBoundaries (NUM_BOUNDARIES is a template argument)
template <int N> int call(const float *b, float v) { return Bucketize<N, float>(*reinterpret_cast<const float(*)[N]>(b), v); }
using Fn = int (*)(const float *, float);
template <int... I> constexpr void fill(Fn *t, std::integer_sequence<int, I...>) { ((t[I] = &call<I + 1>), ...); }
struct Table {
  Fn fn[kMaxBoundaries];
  Table() { fill(fn, std::make_integer_sequence<int, kMaxBoundaries>()); }
};
const Table kTable;
} // namespace

extern "C" int ref_bucketize_max_boundaries(void) { return kMaxBoundaries; }
// bucket index of `value` among boundaries[0..n): the reference's own code; -1 if n is not instantiated
extern "C" int ref_bucketize(const float *boundaries, int n, float value) {
  if (n < 1 || n > kMaxBoundaries) return -1;
  return kTable.fn[n - 1](boundaries, value);
}
extern "C" void ref_bucketize_many(const float *boundaries, int n, const float *values, int64_t count, int32_t *out) {
  for (int64_t i = 0; i < count; ++i) out[i] = ref_bucketize(boundaries, n, values[i]);
}
// the arena alignment of the reference's generated host code (cuda_emitter.cc:2151-2179 sums / steps by it)
extern "C" int ref_alignmem(int x) { return alignmem(x); }
