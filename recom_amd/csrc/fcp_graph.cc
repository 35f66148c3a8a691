// fcp_graph.cc — plan builder from a TensorFlow GraphDef, behind the C ABI (fcp_graph_build, include/fcp_hip.h).
//
// The non-codegen half of the reference's CudaEmitter as ONE in-process call, so that the retained Grappler pass can
// replace `CudaEmitter::Optimize` (graph_optimizers/cuda_emitter.cc:80-116: generate CUDA text, run nvcc, cache the
// .so by md5, rewrite the graph) with: serialize the GraphDef, call fcp_graph_build, parse the rewritten GraphDef.
// No TensorFlow, no protobuf library, no Python: the GraphDef wire format is read and written here (public field
// numbers of TF 2.6.2's graph.proto / node_def.proto / attr_value.proto / tensor.proto / tensor_shape.proto; fields
// this file does not know survive byte for byte).
//
// What follows the reference, and where:
//   * tables (VariableV2 / Const / VarHandleOp whose consumers are only lookups)      graph_info.cc:209-259
//   * value node of a concat input through trailing Reshape / ExpandDims / Squeeze     FindFCOutputs, cuda_emitter.cc:1060-1069
//   * dispatch on GatherV2 / SparseSegment{Sum,Mean}WithNumSegments / ScatterNd / Sum  EmitSubgraphCode :1096-1152
//   * index operands through Reshape-likes, Cast, Bucketize, the [:, 0:1] StridedSlice, an identity SparseReshape
//     and the CPU id ops (SelectValue / GatherIndiceValue / GatherValueGenIndice)       EmitInputInline :1769-1949
//   * the three ops that replace the subgraphs, wired as Rewrite does                   :2496-2656
// The same walk exists in Python (recom_amd/graph/: the offline tool `python -m recom_amd.graph`);
// tests/test_graph_plan.py requires both builders to write identical plan files and equal rewritten graphs.
// Host-only plain C++ (compiled with g++ into libfcp_hip.so).
#include "fcp_env.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fcp_hip.h"

// diagnostic: FCP_DIAG=graph_debug prints the plan builder's decisions on stderr
static bool graph_debug() {
  static const bool on = fcp::diag_on("graph_debug");
  return on;
}

int fcp_internal_fail(int code, const std::string &msg); // fcp_plan.hip: sets fcp_last_error

namespace {

// ====================================================================================================
// protobuf wire format, generic
// ====================================================================================================
struct Field {
  uint32_t num = 0;
  uint8_t wt = 0;       // 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
  uint64_t v = 0;       // varint / fixed value
  std::string bytes;    // wire type 2 payload
};
struct Message {
  std::vector<Field> f;
};

struct ParseError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// "this subgraph is not one the fused path takes"
struct Unsupported : std::runtime_error {
  using std::runtime_error::runtime_error;
};

bool read_varint(const uint8_t *&p, const uint8_t *end, uint64_t *out) {
  uint64_t v = 0;
  for (int shift = 0; shift < 64 && p < end; shift += 7) {
    const uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) {
      *out = v;
      return true;
    }
  }
  return false;
}

Message parse_message(const std::string &s) {
  Message m;
  const uint8_t *p = reinterpret_cast<const uint8_t *>(s.data()), *end = p + s.size();
  while (p < end) {
    uint64_t key;
    if (!read_varint(p, end, &key)) throw ParseError("truncated field key");
    Field f;
    f.num = (uint32_t)(key >> 3);
    f.wt = (uint8_t)(key & 7);
    if (f.num == 0) throw ParseError("field number 0");
    switch (f.wt) {
    case 0:
      if (!read_varint(p, end, &f.v)) throw ParseError("truncated varint");
      break;
    case 1:
      if (end - p < 8) throw ParseError("truncated fixed64");
      memcpy(&f.v, p, 8);
      p += 8;
      break;
    case 5: {
      if (end - p < 4) throw ParseError("truncated fixed32");
      uint32_t x;
      memcpy(&x, p, 4);
      f.v = x;
      p += 4;
      break;
    }
    case 2: {
      uint64_t n;
      if (!read_varint(p, end, &n) || n > (uint64_t)(end - p)) throw ParseError("truncated length-delimited field");
      f.bytes.assign(reinterpret_cast<const char *>(p), (size_t)n);
      p += n;
      break;
    }
    default: throw ParseError("unsupported wire type (groups)");
    }
    m.f.push_back(std::move(f));
  }
  return m;
}

void put_varint(std::string &o, uint64_t v) {
  while (v >= 0x80) {
    o.push_back((char)(v | 0x80));
    v >>= 7;
  }
  o.push_back((char)v);
}

std::string serialize(const Message &m) {
  std::string o;
  for (const Field &f : m.f) {
    put_varint(o, ((uint64_t)f.num << 3) | f.wt);
    switch (f.wt) {
    case 0: put_varint(o, f.v); break;
    case 1: o.append(reinterpret_cast<const char *>(&f.v), 8); break;
    case 5: {
      const uint32_t x = (uint32_t)f.v;
      o.append(reinterpret_cast<const char *>(&x), 4);
      break;
    }
    default:
      put_varint(o, f.bytes.size());
      o += f.bytes;
    }
  }
  return o;
}

void add_varint(Message &m, uint32_t num, uint64_t v) {
  Field f;
  f.num = num;
  f.wt = 0;
  f.v = v;
  m.f.push_back(f);
}
void add_bytes(Message &m, uint32_t num, const std::string &b) {
  Field f;
  f.num = num;
  f.wt = 2;
  f.bytes = b;
  m.f.push_back(std::move(f));
}
const Field *find_field(const Message &m, uint32_t num) { // the LAST occurrence wins (protobuf scalar semantics)
  const Field *r = nullptr;
  for (const Field &f : m.f)
    if (f.num == num) r = &f;
  return r;
}
// repeated varint field, packed or not
std::vector<int64_t> repeated_varints(const Message &m, uint32_t num) {
  std::vector<int64_t> out;
  for (const Field &f : m.f) {
    if (f.num != num) continue;
    if (f.wt == 0) out.push_back((int64_t)f.v);
    else if (f.wt == 2) {
      const uint8_t *p = reinterpret_cast<const uint8_t *>(f.bytes.data()), *end = p + f.bytes.size();
      uint64_t v;
      while (p < end) {
        if (!read_varint(p, end, &v)) throw ParseError("truncated packed varints");
        out.push_back((int64_t)v);
      }
    }
  }
  return out;
}
template <typename T> std::vector<T> repeated_fixed(const Message &m, uint32_t num) { // float / double lists, packed or not
  std::vector<T> out;
  for (const Field &f : m.f) {
    if (f.num != num) continue;
    if (f.wt == 2) {
      if (f.bytes.size() % sizeof(T)) throw ParseError("packed fixed-width list of odd size");
      const size_t n = f.bytes.size() / sizeof(T);
      const size_t at = out.size();
      out.resize(at + n);
      memcpy(out.data() + at, f.bytes.data(), f.bytes.size());
    } else {
      T x;
      memcpy(&x, &f.v, sizeof(T));
      out.push_back(x);
    }
  }
  return out;
}
std::string packed_varints(const std::vector<int64_t> &v) {
  std::string o;
  for (int64_t x : v) put_varint(o, (uint64_t)x);
  return o;
}

// ====================================================================================================
// TensorFlow messages on top of it
// ====================================================================================================
enum { DT_FLOAT = 1, DT_DOUBLE = 2, DT_INT32 = 3, DT_UINT8 = 4, DT_INT16 = 5, DT_INT8 = 6, DT_STRING = 7, DT_INT64 = 9, DT_BOOL = 10 };

using Dim = std::optional<int64_t>;           // nullopt: dynamic
using Shape = std::optional<std::vector<Dim>>; // nullopt: even the rank is unknown

struct Node {
  Message raw;                         // the NodeDef as read (new nodes: as built)
  std::string name, op;
  std::vector<std::string> input;
  std::map<std::string, Message> attr; // AttrValue messages by name
};

Node parse_node(const std::string &bytes) {
  Node n;
  n.raw = parse_message(bytes);
  for (const Field &f : n.raw.f) {
    if (f.num == 1 && f.wt == 2) n.name = f.bytes;
    else if (f.num == 2 && f.wt == 2) n.op = f.bytes;
    else if (f.num == 3 && f.wt == 2) n.input.push_back(f.bytes);
    else if (f.num == 5 && f.wt == 2) {
      const Message e = parse_message(f.bytes);
      const Field *k = find_field(e, 1), *v = find_field(e, 2);
      n.attr[k ? k->bytes : std::string()] = v ? parse_message(v->bytes) : Message();
    }
  }
  return n;
}

bool has_attr(const Node &n, const char *key) { return n.attr.count(key) != 0; }
const Message &attr(const Node &n, const char *key) {
  static const Message empty;
  auto it = n.attr.find(key);
  return it == n.attr.end() ? empty : it->second;
}
int attr_type(const Node &n, const char *key) {
  const Field *f = find_field(attr(n, key), 6);
  return f ? (int)f->v : 0;
}
int64_t attr_i(const Node &n, const char *key) {
  const Field *f = find_field(attr(n, key), 3);
  return f ? (int64_t)f->v : 0;
}
bool attr_b(const Node &n, const char *key) {
  const Field *f = find_field(attr(n, key), 5);
  return f && f->v != 0;
}
std::string attr_s(const Node &n, const char *key) {
  const Field *f = find_field(attr(n, key), 2);
  return f ? f->bytes : std::string();
}
Message attr_list(const Node &n, const char *key) {
  const Field *f = find_field(attr(n, key), 1);
  return f ? parse_message(f->bytes) : Message();
}

Shape shape_of_proto(const Message &shape) { // TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 }; bool unknown_rank = 3 }
  const Field *unk = find_field(shape, 3);
  if (unk && unk->v) return std::nullopt;
  std::vector<Dim> out;
  for (const Field &f : shape.f) {
    if (f.num != 2 || f.wt != 2) continue;
    const Message dim = parse_message(f.bytes);
    const Field *sz = find_field(dim, 1);
    const int64_t v = sz ? (int64_t)sz->v : 0;
    out.push_back(v >= 0 ? Dim(v) : Dim());
  }
  return out;
}

// a small integer / float tensor as numbers (Const values the walk looks at: axes, slice bounds, reshape targets)
struct Array {
  int dtype = 0;
  std::vector<int64_t> shape;
  std::vector<double> v; // every supported dtype is exactly representable (ints up to 2^53: shapes and indices)
};

std::optional<Array> tensor_to_array(const Message &t) {
  Array a;
  const Field *dt = find_field(t, 1);
  a.dtype = dt ? (int)dt->v : 0;
  size_t elem = 0;
  switch (a.dtype) {
  case DT_FLOAT: case DT_INT32: elem = 4; break;
  case DT_DOUBLE: case DT_INT64: elem = 8; break;
  case DT_BOOL: case DT_INT8: case DT_UINT8: elem = 1; break;
  case DT_INT16: elem = 2; break;
  default: return std::nullopt;
  }
  if (const Field *sh = find_field(t, 2)) {
    const Shape s = shape_of_proto(parse_message(sh->bytes));
    if (!s) return std::nullopt;
    for (const Dim &d : *s) a.shape.push_back(d ? *d : 0);
  }
  // The walk only reads axes, slice bounds and reshape targets: a Const beyond kMaxConstElements is "not a constant we
  // look at" (a crafted shape such as [2^61 + 1] with a one-value splat list must neither overflow the product nor make
  // this reader allocate what the shape claims).
  constexpr size_t kMaxConstElements = (size_t)1 << 20;
  size_t n = 1;
  for (int64_t d : a.shape) {
    if (d < 0) return std::nullopt;
    if (d != 0 && n > kMaxConstElements / (size_t)d) return std::nullopt;
    n *= (size_t)d;
  }
  const Field *content = find_field(t, 4);
  if (content && !content->bytes.empty()) {
    if (content->bytes.size() < n * elem) return std::nullopt;
    const char *p = content->bytes.data();
    a.v.resize(n);
    for (size_t i = 0; i < n; ++i) {
      switch (a.dtype) {
      case DT_FLOAT: { float x; memcpy(&x, p + 4 * i, 4); a.v[i] = x; break; }
      case DT_DOUBLE: { double x; memcpy(&x, p + 8 * i, 8); a.v[i] = x; break; }
      case DT_INT32: { int32_t x; memcpy(&x, p + 4 * i, 4); a.v[i] = x; break; }
      case DT_INT64: { int64_t x; memcpy(&x, p + 8 * i, 8); a.v[i] = (double)x; break; }
      case DT_INT16: { int16_t x; memcpy(&x, p + 2 * i, 2); a.v[i] = x; break; }
      case DT_INT8: a.v[i] = (int8_t)p[i]; break;
      case DT_UINT8: a.v[i] = (uint8_t)p[i]; break;
      default: a.v[i] = p[i] != 0; break;
      }
    }
    return a;
  }
  // typed value lists; a short list repeats its last value (TensorFlow's MakeNdarray rule)
  std::vector<double> vals;
  switch (a.dtype) {
  case DT_FLOAT: for (float x : repeated_fixed<float>(t, 5)) vals.push_back(x); break;
  case DT_DOUBLE: for (double x : repeated_fixed<double>(t, 6)) vals.push_back(x); break;
  case DT_INT64: for (int64_t x : repeated_varints(t, 10)) vals.push_back((double)x); break;
  case DT_BOOL: for (int64_t x : repeated_varints(t, 11)) vals.push_back(x != 0); break;
  default: for (int64_t x : repeated_varints(t, 7)) vals.push_back((double)(int32_t)x); break; // int_val: int32, int16, int8, uint8
  }
  if (vals.empty()) vals.push_back(0.0);
  a.v.resize(n);
  for (size_t i = 0; i < n; ++i) a.v[i] = vals[std::min(i, vals.size() - 1)];
  return a;
}

std::pair<std::string, int> split_tensor(const std::string &t) { // "node:2" -> ("node", 2); "^node" -> ("node", -1)
  if (!t.empty() && t[0] == '^') return {t.substr(1), -1};
  const size_t c = t.find(':');
  if (c == std::string::npos) return {t, 0};
  return {t.substr(0, c), atoi(t.c_str() + c + 1)};
}
std::string tensor_name(const std::string &node, int port) { return port == 0 ? node : node + ":" + std::to_string(port); }

// ====================================================================================================
// read-only view of the graph (recom_amd/graph/view.py)
// ====================================================================================================
struct StridedSliceSpec {
  std::vector<int64_t> begin, end, strides;
  int64_t begin_mask = 0, end_mask = 0, ellipsis_mask = 0, new_axis_mask = 0, shrink_axis_mask = 0;
};
struct ElemSource { // where one element of a small integer tensor comes from
  bool is_const = false;
  int64_t value = 0;   // const
  std::string tensor;  // elem: a plain copy of element `index` of this tensor
  int64_t index = 0;
  bool operator==(const ElemSource &o) const { return is_const == o.is_const && value == o.value && tensor == o.tensor && index == o.index; }
};
using NodeRef = std::pair<const Node *, int>; // (producer, output port)

struct GraphView {
  std::vector<Node> nodes;                 // graph order
  std::map<std::string, int> index;        // name -> position
  std::vector<std::vector<std::pair<int, int>>> consumers; // per node: (consumer position, its input slot or -1 for control)

  explicit GraphView(const std::vector<Node> &ns) : nodes(ns) {
    for (size_t k = 0; k < nodes.size(); ++k) {
      if (index.count(nodes[k].name)) throw std::invalid_argument("duplicate node name " + nodes[k].name);
      index[nodes[k].name] = (int)k;
    }
    consumers.resize(nodes.size());
    for (size_t k = 0; k < nodes.size(); ++k)
      for (size_t i = 0; i < nodes[k].input.size(); ++i) {
        const auto [src, port] = split_tensor(nodes[k].input[i]);
        auto it = index.find(src);
        if (it == index.end()) throw std::invalid_argument("node " + nodes[k].name + ": input " + nodes[k].input[i] + " not in graph");
        consumers[it->second].push_back({(int)k, port >= 0 ? (int)i : -1});
      }
  }
  const Node *node(const std::string &name) const {
    auto it = index.find(name);
    return it == index.end() ? nullptr : &nodes[it->second];
  }
  std::vector<std::string> data_inputs(const Node &n) const {
    std::vector<std::string> out;
    for (const std::string &t : n.input)
      if (t.empty() || t[0] != '^') out.push_back(t);
    return out;
  }
  NodeRef input(const Node &n, int i) const {
    const std::vector<std::string> d = data_inputs(n);
    if (i < 0 || i >= (int)d.size()) throw Unsupported("node " + n.name + " has no data input " + std::to_string(i));
    const auto [name, port] = split_tensor(d[i]);
    return {node(name), port};
  }
  std::vector<std::pair<const Node *, int>> data_consumers(const std::string &name) const {
    std::vector<std::pair<const Node *, int>> out;
    auto it = index.find(name);
    if (it == index.end()) return out;
    for (const auto &c : consumers[it->second])
      if (c.second >= 0) out.push_back({&nodes[c.first], c.second});
    return out;
  }

  std::optional<Array> const_array(NodeRef r) const {
    const Node *n = r.first;
    int port = r.second;
    while ((n->op == "Identity" || n->op == "StopGradient") && port == 0) std::tie(n, port) = input(*n, 0);
    if (n->op != "Const" || port != 0) return std::nullopt;
    const Field *t = find_field(attr(*n, "value"), 8);
    if (!t) return std::nullopt;
    return tensor_to_array(parse_message(t->bytes));
  }

  int out_dtype(const Node &n, int port = 0) const {
    static const std::map<std::string, const char *> key = {
        {"Placeholder", "dtype"}, {"Const", "dtype"}, {"VariableV2", "dtype"}, {"GatherV2", "Tparams"}, {"ResourceGather", "dtype"},
        {"ReadVariableOp", "dtype"}, {"Cast", "DstT"}, {"Shape", "out_type"}, {"Size", "out_type"}};
    auto it = key.find(n.op);
    if (it != key.end()) return attr_type(n, it->second);
    if (n.op == "Bucketize") return DT_INT32;
    if (n.op == "SparseReshape" || n.op == "StringToHashBucketFast") return DT_INT64;
    if (n.op == "AsString") return DT_STRING;
    if (n.op == "Unique") return port == 0 ? attr_type(n, "T") : attr_type(n, "out_idx");
    if (has_attr(n, "T")) return attr_type(n, "T");
    if (has_attr(n, "dtype")) return attr_type(n, "dtype");
    throw Unsupported("cannot tell output dtype of " + n.name + " (" + n.op + ")");
  }

  std::optional<StridedSliceSpec> strided_slice_spec(const Node &n) const {
    StridedSliceSpec s;
    std::vector<int64_t> *dst[3] = {&s.begin, &s.end, &s.strides};
    for (int k = 0; k < 3; ++k) {
      const std::optional<Array> a = const_array(input(n, k + 1));
      if (!a) return std::nullopt;
      for (double v : a->v) dst[k]->push_back((int64_t)v);
    }
    s.begin_mask = attr_i(n, "begin_mask");
    s.end_mask = attr_i(n, "end_mask");
    s.ellipsis_mask = attr_i(n, "ellipsis_mask");
    s.new_axis_mask = attr_i(n, "new_axis_mask");
    s.shrink_axis_mask = attr_i(n, "shrink_axis_mask");
    return s;
  }

  static bool all_known(const std::vector<Dim> &s) {
    for (const Dim &d : s)
      if (!d) return false;
    return true;
  }

  // list of dims (nullopt = dynamic), or nullopt when even the rank is unknown
  Shape static_shape(NodeRef r, int depth = 0) const {
    const Node &n = *r.first;
    const int port = r.second;
    if (has_attr(n, "_output_shapes")) {
      const Message l = attr_list(n, "_output_shapes");
      int k = 0;
      for (const Field &f : l.f) {
        if (f.num != 7 || f.wt != 2) continue;
        if (k++ == port) {
          const Shape s = shape_of_proto(parse_message(f.bytes));
          if (s) return s;
          break;
        }
      }
    }
    if (depth > 64) return std::nullopt;
    const std::string &op = n.op;
    auto in_shape = [&](int i) { return static_shape(input(n, i), depth + 1); };
    if ((op == "Placeholder" || op == "VariableV2") && has_attr(n, "shape")) {
      const Field *s = find_field(attr(n, "shape"), 7);
      return s ? shape_of_proto(parse_message(s->bytes)) : Shape(std::vector<Dim>());
    }
    if (op == "Const") {
      const Field *t = find_field(attr(n, "value"), 8);
      if (!t) return std::nullopt;
      const Message tensor = parse_message(t->bytes);
      const Field *sh = find_field(tensor, 2);
      std::vector<Dim> out;
      if (sh) {
        const Message shape = parse_message(sh->bytes);
        for (const Field &f : shape.f)
          if (f.num == 2 && f.wt == 2) {
            const Message dim = parse_message(f.bytes);
            const Field *sz = find_field(dim, 1);
            out.push_back(Dim(sz ? (int64_t)sz->v : 0));
          }
      }
      return out;
    }
    if (op == "Identity" || op == "Cast" || op == "Bucketize" || op == "StopGradient" || op == "ZerosLike" || op == "AsString" ||
        op == "StringToHashBucketFast")
      return in_shape(0);
    if (op == "Reshape") {
      const std::optional<Array> tgt = const_array(input(n, 1));
      if (!tgt) return std::nullopt;
      std::vector<Dim> dims;
      for (double d : tgt->v) dims.push_back(d >= 0 ? Dim((int64_t)d) : Dim());
      const Shape src = in_shape(0);
      const int unknown = (int)std::count(dims.begin(), dims.end(), Dim());
      if (unknown == 1 && src && all_known(*src)) {
        int64_t known = 1;
        if (dims.size() > 1)
          for (const Dim &d : dims)
            if (d) known *= *d;
        if (known) {
          int64_t total = 1;
          for (const Dim &d : *src) total *= *d;
          *std::find(dims.begin(), dims.end(), Dim()) = total / known;
        }
      }
      return dims;
    }
    if (op == "ExpandDims") {
      const Shape src = in_shape(0);
      const std::optional<Array> ax = const_array(input(n, 1));
      if (!src || !ax || ax->v.empty()) return std::nullopt;
      int64_t a = (int64_t)ax->v[0];
      if (a < 0) a += (int64_t)src->size() + 1;
      if (a < 0 || a > (int64_t)src->size()) return std::nullopt;
      std::vector<Dim> out = *src;
      out.insert(out.begin() + a, Dim(1));
      return out;
    }
    if (op == "Squeeze") {
      const Shape src = in_shape(0);
      if (!src) return std::nullopt;
      std::vector<int64_t> dims = repeated_varints(attr_list(n, "squeeze_dims"), 3);
      for (int64_t &d : dims)
        if (d < 0) d += (int64_t)src->size();
      std::vector<Dim> out;
      if (dims.empty()) {
        if (!all_known(*src)) return std::nullopt;
        for (const Dim &d : *src)
          if (*d != 1) out.push_back(d);
        return out;
      }
      for (size_t k = 0; k < src->size(); ++k)
        if (std::find(dims.begin(), dims.end(), (int64_t)k) == dims.end()) out.push_back((*src)[k]);
      return out;
    }
    if (op == "StridedSlice") {
      const Shape src = in_shape(0);
      const std::optional<StridedSliceSpec> spec = strided_slice_spec(n);
      if (!src || !spec || spec->begin.size() > src->size()) return std::nullopt;
      std::vector<Dim> out;
      for (size_t k = 0; k < src->size(); ++k) {
        const Dim d = (*src)[k];
        if (k >= spec->begin.size()) {
          out.push_back(d);
          continue;
        }
        if (spec->shrink_axis_mask >> k & 1) continue;
        const int64_t b = spec->begin[k], e = spec->end[k], s = spec->strides[k];
        const bool bm = spec->begin_mask >> k & 1, em = spec->end_mask >> k & 1;
        if (s != 1) return std::nullopt;
        if (bm && em) out.push_back(d);
        else if (!d && (bm || em || b < 0 || e < 0)) out.push_back(Dim());
        else {
          const int64_t lo = bm ? 0 : (b < 0 ? b + *d : b);
          const int64_t hi = em ? *d : (e < 0 ? e + *d : e);
          out.push_back(Dim(std::max<int64_t>(0, std::min(hi, d ? *d : hi) - lo)));
        }
      }
      return out;
    }
    if (op == "VarHandleOp" || op == "ReadVariableOp") { // the variable's shape (attr of the handle op)
      const Node &h = op == "VarHandleOp" ? n : *input(n, 0).first;
      if (h.op != "VarHandleOp") return std::nullopt;
      const Field *s = find_field(attr(h, "shape"), 7);
      return s ? shape_of_proto(parse_message(s->bytes)) : Shape(std::vector<Dim>());
    }
    if (op == "GatherV2" || op == "ResourceGather") {
      const Shape p = in_shape(0), i = in_shape(1);
      if (!p || !i || p->empty()) return std::nullopt;
      std::vector<Dim> out = *i;
      out.insert(out.end(), p->begin() + 1, p->end());
      return out;
    }
    if (op == "Addons>SelectValue") return in_shape(0);
    if (op == "Addons>GatherIndiceValue") { // (surviving indices [n, k], surviving values [n])
      if (port != 0) return std::vector<Dim>{Dim()};
      const Shape idx = in_shape(0);
      return std::vector<Dim>{Dim(), idx && idx->size() == 2 ? (*idx)[1] : Dim()};
    }
    if (op == "Addons>GatherValueGenIndice") return port == 0 ? std::vector<Dim>{Dim(), Dim(1)} : std::vector<Dim>{Dim()};
    if (op == "SparseReshape") { // (output_indices [nnz, rank(new_shape)], output_shape [rank(new_shape)])
      const Shape idx = in_shape(0), nw = in_shape(2);
      if (!nw || nw->size() != 1 || !(*nw)[0]) return std::nullopt;
      if (port == 0) return std::vector<Dim>{idx && !idx->empty() ? (*idx)[0] : Dim(), (*nw)[0]};
      return std::vector<Dim>{(*nw)[0]};
    }
    if (op == "Pack") {
      const Shape first = in_shape(0);
      const int64_t axis = has_attr(n, "axis") ? attr_i(n, "axis") : 0;
      if (!first || axis != 0) return std::nullopt;
      std::vector<Dim> out{Dim((int64_t)data_inputs(n).size())};
      out.insert(out.end(), first->begin(), first->end());
      return out;
    }
    if (op == "Prod" || op == "Sum" || op == "Max" || op == "Min") {
      const Shape src = in_shape(0);
      const std::optional<Array> ax = const_array(input(n, 1));
      if (!src || !ax || (has_attr(n, "keep_dims") && attr_b(n, "keep_dims"))) return std::nullopt;
      std::set<int64_t> axes;
      for (double a : ax->v) axes.insert((int64_t)a < 0 ? (int64_t)a + (int64_t)src->size() : (int64_t)a);
      std::vector<Dim> out;
      for (size_t k = 0; k < src->size(); ++k)
        if (!axes.count((int64_t)k)) out.push_back((*src)[k]);
      return out;
    }
    return std::nullopt;
  }

  // element k (flat index) of a shape-like tensor, traced to a constant or to a plain copy of one element of another
  // tensor; nullopt when it is computed (lets the builder prove two shape entries equal without a symbolic engine)
  std::optional<ElemSource> elem_source(NodeRef r, int64_t k, int depth = 0) const {
    if (depth > 32) return std::nullopt;
    const Node &n = *r.first;
    const int port = r.second;
    const std::string &op = n.op;
    auto elem = [&]() {
      ElemSource e;
      e.tensor = tensor_name(n.name, port);
      e.index = k;
      return std::optional<ElemSource>(e);
    };
    if (op == "Const" && port == 0) {
      const std::optional<Array> a = const_array(r);
      if (!a || k < 0 || k >= (int64_t)a->v.size()) return std::nullopt;
      ElemSource e;
      e.is_const = true;
      e.value = (int64_t)a->v[k];
      return e;
    }
    if ((op == "Identity" || op == "StopGradient" || op == "Reshape" || op == "Squeeze" || op == "ExpandDims") && port == 0)
      return elem_source(input(n, 0), k, depth + 1); // flat order unchanged
    if (op == "Cast" && port == 0) {
      auto is_int = [](int t) { return t == DT_INT32 || t == DT_INT64; };
      if (is_int(attr_type(n, "SrcT")) && is_int(attr_type(n, "DstT"))) return elem_source(input(n, 0), k, depth + 1);
      return std::nullopt;
    }
    if (op == "Pack" && port == 0) {
      if ((has_attr(n, "axis") && attr_i(n, "axis") != 0) || k < 0 || k >= (int64_t)data_inputs(n).size()) return std::nullopt;
      const NodeRef src = input(n, (int)k);
      const Shape s = static_shape(src);
      return s && s->empty() ? elem_source(src, 0, depth + 1) : std::nullopt;
    }
    if (op == "ConcatV2" && port == 0) {
      const int n_in = (int)data_inputs(n).size();
      for (int i = 0; i + 1 < n_in; ++i) {
        const NodeRef src = input(n, i);
        const Shape s = static_shape(src);
        if (!s || s->size() != 1 || !(*s)[0]) return std::nullopt;
        if (k < *(*s)[0]) return elem_source(src, k, depth + 1);
        k -= *(*s)[0];
      }
      return std::nullopt;
    }
    if (op == "StridedSlice" && port == 0) {
      const std::optional<StridedSliceSpec> spec = strided_slice_spec(n);
      const NodeRef src = input(n, 0);
      const Shape s = static_shape(src);
      if (!spec || !s || s->size() != 1 || spec->begin.size() != 1 || spec->strides != std::vector<int64_t>{1}) return std::nullopt;
      if (spec->ellipsis_mask || spec->new_axis_mask) return std::nullopt;
      int64_t b = (spec->begin_mask & 1) ? 0 : spec->begin[0];
      if (b < 0) {
        if (!(*s)[0]) return std::nullopt;
        b += *(*s)[0];
      }
      return elem_source(src, b + k, depth + 1);
    }
    if (op == "GatherV2" && port == 0) {
      const std::optional<Array> idx = const_array(input(n, 1)), ax = const_array(input(n, 2));
      const NodeRef src = input(n, 0);
      const Shape s = static_shape(src);
      if (!idx || !ax || ax->v.empty() || (int64_t)ax->v[0] != 0 || !s || s->size() != 1) return std::nullopt;
      if (k < 0 || k >= (int64_t)idx->v.size()) return std::nullopt;
      int64_t i = (int64_t)idx->v[k];
      if (i < 0) {
        if (!(*s)[0]) return std::nullopt;
        i += *(*s)[0];
      }
      return elem_source(src, i, depth + 1);
    }
    if ((op == "Prod" || op == "Sum" || op == "Max" || op == "Min") && port == 0) {
      const NodeRef src = input(n, 0);
      const Shape s = static_shape(src);
      if (s && s->size() == 1 && (*s)[0] && *(*s)[0] == 1 && k == 0) return elem_source(src, 0, depth + 1); // a reduction of one element is that element
      return std::nullopt;
    }
    if (op == "SparseReshape" && port == 1) return elem_source(input(n, 2), k, depth + 1); // output_shape = new_shape (no -1 handled)
    if (op == "Placeholder" || op == "PlaceholderWithDefault" || port != 0 || data_inputs(n).empty()) return elem();
    return op == "Shape" ? elem() : std::nullopt;
  }
};

// ====================================================================================================
// the column plan (recom_amd/plan.py)
// ====================================================================================================
struct Column {
  int form = 0, dim = 0;
  int64_t vocab = 0;
  int combiner = 0, id_source = FCP_IDS_I64, table_input = -1, ids_input = -1, seg_input = -1, seg_kind = FCP_SEG_NONE, seg_stride = 1,
      rows_source = FCP_ROWS_FROM_IDS, rows_arg = 0;
  std::vector<float> boundaries;
  bool has_boundaries = false;
  int concat_group = 0, concat_slot = 0;
  int xform_mode = FCP_XFORM_NONE;
  std::vector<int64_t> xform_lo, xform_hi;
  int64_t xform_substitute = 0, hash_buckets = 0;
  // segment ids through a folded SparseReshape (fcp_column_ext_t::seg_map_*)
  std::vector<int64_t> seg_mul;
  int64_t seg_div = 1;
  int seg_sym = -1, seg_sym_slot = 0;
};
struct Plan {
  std::vector<Column> columns;
  std::vector<int> host_ranks, host_esizes;
  int n_device_inputs = 0, n_groups = 1, n_symbols = 0, layout = FCP_LAYOUT_CONCAT;
};
struct StageInfo {
  std::vector<int> modes, rows_symbol;
  int symbols_input = -1;
};

bool is_lookup(int form) { return form == FCP_FORM_GATHER || form == FCP_FORM_SEGMENT_REDUCE || form == FCP_FORM_GATHER_SCATTER; }

void validate_plan(const Plan &p) { // ColumnSpec.validate + PlanSpec.validate
  if (p.host_ranks.size() != p.host_esizes.size()) throw std::invalid_argument("host_input_ranks / elem_sizes length mismatch");
  std::set<std::pair<int, int>> seen;
  for (size_t k = 0; k < p.columns.size(); ++k) {
    const Column &c = p.columns[k];
    const std::string where = "column " + std::to_string(k) + ": ";
    if (c.form < 1 || c.form > 6) throw std::invalid_argument(where + "bad form");
    if (c.dim <= 0) throw std::invalid_argument(where + "dim must be positive");
    if ((c.form == FCP_FORM_EXTERNAL) != (c.rows_source == FCP_ROWS_FROM_GROUP))
      throw std::invalid_argument(where + "external slots (and only they) take their row count from their concat group");
    if (is_lookup(c.form)) {
      if (c.vocab <= 0 || c.table_input < 0 || c.ids_input < 0) throw std::invalid_argument(where + "lookup column needs vocab, table_input, ids_input");
      if (c.id_source == FCP_IDS_F32_BUCKETIZE && (!c.has_boundaries || c.boundaries.empty()))
        throw std::invalid_argument(where + "bucketize column needs boundaries");
    }
    if (c.form == FCP_FORM_SEGMENT_REDUCE || c.form == FCP_FORM_GATHER_SCATTER) {
      if (c.seg_kind == FCP_SEG_NONE || c.seg_input < 0) throw std::invalid_argument(where + "pooled/scatter column needs segment input");
      if (c.rows_source == FCP_ROWS_FROM_IDS) throw std::invalid_argument(where + "pooled/scatter column needs an explicit row count source");
    }
    if (c.form == FCP_FORM_SEGMENT_REDUCE && c.combiner != FCP_COMBINER_SUM && c.combiner != FCP_COMBINER_MEAN)
      throw std::invalid_argument(where + "segment-reduce column needs sum or mean");
    if (!c.seg_mul.empty()) {
      if (c.form != FCP_FORM_SEGMENT_REDUCE || (c.seg_kind != FCP_SEG_IDS_I32 && c.seg_kind != FCP_SEG_IDS_I64))
        throw std::invalid_argument(where + "a segment-id map needs a pooled column with segment ids");
      bool neg = false;
      for (int64_t v : c.seg_mul) neg = neg || v < 0;
      if (c.seg_mul.size() > 4 || c.seg_stride < (int)c.seg_mul.size() || c.seg_div < 1 || neg) throw std::invalid_argument(where + "bad segment-id map");
      if (c.seg_sym >= 0 && !(c.seg_sym_slot == 4 || (c.seg_sym_slot >= 0 && c.seg_sym_slot < (int)c.seg_mul.size())))
        throw std::invalid_argument(where + "bad segment-id map symbol slot");
    }
    if (c.hash_buckets && (!is_lookup(c.form) || c.hash_buckets < 0 || c.id_source == FCP_IDS_F32_BUCKETIZE))
      throw std::invalid_argument(where + "hash_buckets applies to the integer ids of lookup columns");
    if (c.xform_mode != FCP_XFORM_NONE) {
      if (!is_lookup(c.form)) throw std::invalid_argument(where + "id transforms apply to lookup columns only");
      if ((c.xform_mode != FCP_XFORM_SELECT && c.xform_mode != FCP_XFORM_FILTER) || c.xform_lo.size() != c.xform_hi.size())
        throw std::invalid_argument(where + "bad id transform");
      for (size_t i = 0; i < c.xform_lo.size(); ++i)
        if (c.xform_lo[i] > c.xform_hi[i]) throw std::invalid_argument(where + "empty id transform interval");
    }
    if (c.concat_group < 0 || c.concat_group >= p.n_groups) throw std::invalid_argument(where + "concat_group out of range");
    if (!seen.insert({c.concat_group, c.concat_slot}).second) throw std::invalid_argument(where + "duplicate concat slot");
    if (c.ids_input >= (int)p.host_ranks.size() || c.seg_input >= (int)p.host_ranks.size()) throw std::invalid_argument(where + "host input index out of range");
    if (c.table_input >= p.n_device_inputs) throw std::invalid_argument(where + "table_input out of range");
    if (c.rows_source == FCP_ROWS_FROM_SYMBOL && (c.rows_arg < 0 || c.rows_arg >= p.n_symbols)) throw std::invalid_argument(where + "symbol index out of range");
  }
}

// PlanSpec.narrowed(): every int64 id / segment-id input declared int32 where every reader agrees
std::pair<Plan, std::vector<bool>> narrowed(const Plan &p) {
  std::vector<bool> flags(p.host_ranks.size(), false);
  for (const Column &c : p.columns)
    if (is_lookup(c.form)) {
      if (c.id_source == FCP_IDS_I64 && c.vocab <= 0x7fffffff) flags[c.ids_input] = true;
      if (c.seg_kind == FCP_SEG_IDS_I64) flags[c.seg_input] = true;
    }
  for (const Column &c : p.columns)
    if (is_lookup(c.form) && (c.id_source != FCP_IDS_I64 || c.vocab > 0x7fffffff || c.hash_buckets || c.xform_mode != FCP_XFORM_NONE))
      if (c.ids_input >= 0 && p.host_esizes[c.ids_input] == 8) flags[c.ids_input] = false;
  Plan out = p;
  for (Column &c : out.columns) {
    if (c.ids_input >= 0 && flags[c.ids_input] && c.id_source == FCP_IDS_I64) c.id_source = FCP_IDS_I32;
    if (c.seg_input >= 0 && flags[c.seg_input] && c.seg_kind == FCP_SEG_IDS_I64) c.seg_kind = FCP_SEG_IDS_I32;
  }
  for (size_t i = 0; i < flags.size(); ++i)
    if (flags[i]) out.host_esizes[i] = 4;
  return {out, flags};
}

// PlanSpec.staged_for_concat_inputs(): the plan Addons>ConcatInputs produces its blob for, and its stage section
std::pair<Plan, StageInfo> staged_for_concat_inputs(const Plan &p) {
  auto [spec, flags] = narrowed(p);
  const int nh = (int)p.host_ranks.size();
  StageInfo st;
  st.modes.resize(nh);
  for (int i = 0; i < nh; ++i) st.modes[i] = flags[i] ? FCP_STAGE_NARROW_I64 : FCP_STAGE_COPY;
  std::vector<int> rows_col(nh, -1);
  // pooled columns only: their segment ids are sorted (TF's SparseSegment* contract); a ScatterNd column takes its row ids in any order
  std::map<int, std::vector<int>> users;
  std::vector<int> user_order;
  for (size_t k = 0; k < spec.columns.size(); ++k) {
    const Column &c = spec.columns[k];
    // (a column whose segment ids are computed from several coordinates keeps its index matrix: the pre-pass evaluates the map)
    if (c.form == FCP_FORM_SEGMENT_REDUCE && (c.seg_kind == FCP_SEG_IDS_I32 || c.seg_kind == FCP_SEG_IDS_I64) && c.seg_mul.empty()) {
      if (!users.count(c.seg_input)) user_order.push_back(c.seg_input);
      users[c.seg_input].push_back((int)k);
    }
  }
  for (int i : user_order) {
    const std::vector<int> &ks = users[i];
    const Column &c0 = p.columns[ks[0]];
    bool same = true;
    for (int k : ks) {
      const Column &c = p.columns[k];
      same = same && c.seg_stride == c0.seg_stride && c.rows_source == c0.rows_source && c.rows_arg == c0.rows_arg;
    }
    bool other = false;
    int readers = 0;
    for (const Column &c : p.columns) {
      other = other || c.ids_input == i || (c.rows_source == FCP_ROWS_FROM_INPUT_DIM0 && c.rows_arg == i);
      readers += c.seg_input == i;
    }
    if (same && !other && readers == (int)ks.size() && c0.rows_source == FCP_ROWS_FROM_SYMBOL) {
      st.modes[i] = FCP_STAGE_SEG_TO_CSR;
      rows_col[i] = ks[0];
    }
  }
  for (Column &c : spec.columns)
    if (c.seg_input >= 0 && st.modes[c.seg_input] == FCP_STAGE_SEG_TO_CSR) {
      c.seg_kind = FCP_SEG_CSR_I32;
      c.seg_stride = 1;
    }
  for (int i = 0; i < nh; ++i)
    if (st.modes[i] == FCP_STAGE_SEG_TO_CSR) {
      spec.host_ranks[i] = 1;
      spec.host_esizes[i] = 4;
    }
  st.rows_symbol.resize(nh);
  for (int i = 0; i < nh; ++i) st.rows_symbol[i] = rows_col[i] >= 0 ? p.columns[rows_col[i]].rows_arg : -1;
  if (std::find(st.modes.begin(), st.modes.end(), (int)FCP_STAGE_SEG_TO_CSR) != st.modes.end()) {
    st.symbols_input = nh; // the symbols vector rides along as one more (last) ConcatInputs input
    spec.host_ranks.push_back(1);
    spec.host_esizes.push_back(4);
    st.modes.push_back(FCP_STAGE_COPY);
    st.rows_symbol.push_back(-1);
  }
  validate_plan(spec);
  return {spec, st};
}

// repr(float(np.float32(x))): the shortest decimal string that reads back as the same double, in Python's layout
std::string py_float_repr(float x) {
  const double d = (double)x;
  if (std::isnan(d)) return "nan";
  if (std::isinf(d)) return d > 0 ? "inf" : "-inf";
  if (d == 0) return std::signbit(d) ? "-0.0" : "0.0";
  char buf[64];
  const auto res = std::to_chars(buf, buf + sizeof buf, d, std::chars_format::scientific); // shortest round-trip digits
  std::string s(buf, res.ptr);
  const bool neg = s[0] == '-';
  if (neg) s.erase(0, 1);
  const size_t e = s.find('e');
  std::string digits = s.substr(0, e);
  const int exp10 = atoi(s.c_str() + e + 1);
  digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
  std::string out;
  if (exp10 >= -4 && exp10 < 16) { // fixed notation, always with a fractional part
    if (exp10 < 0) out = "0." + std::string((size_t)(-exp10 - 1), '0') + digits;
    else if ((int)digits.size() <= exp10 + 1) out = digits + std::string((size_t)(exp10 + 1 - (int)digits.size()), '0') + ".0";
    else out = digits.substr(0, (size_t)exp10 + 1) + "." + digits.substr((size_t)exp10 + 1);
  } else {
    out = digits.substr(0, 1);
    if (digits.size() > 1) out += "." + digits.substr(1);
    char eb[16];
    snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', abs(exp10));
    out += eb;
  }
  return neg ? "-" + out : out;
}

// recom_amd.plan_io.save_plan, byte for byte
std::string plan_file_text(const Plan &p, const StageInfo *stage) {
  std::vector<size_t> maps;
  for (size_t k = 0; k < p.columns.size(); ++k)
    if (!p.columns[k].seg_mul.empty()) maps.push_back(k);
  std::string o = std::string("fcp_plan ") + (!maps.empty() ? "4" : stage ? "3" : "2") + "\n";
  o += "layout " + std::to_string(p.layout) + "\n";
  o += "groups " + std::to_string(p.n_groups) + " symbols " + std::to_string(p.n_symbols) + " device_inputs " + std::to_string(p.n_device_inputs) + "\n";
  o += "host_inputs " + std::to_string(p.host_ranks.size()) + "\n";
  for (size_t i = 0; i < p.host_ranks.size(); ++i) o += std::to_string(p.host_ranks[i]) + " " + std::to_string(p.host_esizes[i]) + "\n";
  o += "columns " + std::to_string(p.columns.size()) + "\n";
  for (const Column &c : p.columns) {
    const size_t nb = c.has_boundaries ? c.boundaries.size() : 0;
    const int64_t head[15] = {c.form, c.combiner, c.dim, c.id_source, c.vocab, c.table_input, c.ids_input, c.seg_input, c.seg_kind, c.seg_stride,
                              c.rows_source, c.rows_arg, c.concat_group, c.concat_slot, (int64_t)nb};
    for (int i = 0; i < 15; ++i) o += (i ? " " : "") + std::to_string(head[i]);
    for (size_t i = 0; i < nb; ++i) o += " " + py_float_repr(c.boundaries[i]);
    o += " " + std::to_string(c.xform_mode) + " " + std::to_string(c.xform_lo.size()) + " " + std::to_string(c.xform_substitute) + " " +
         std::to_string(c.hash_buckets);
    for (size_t i = 0; i < c.xform_lo.size(); ++i) o += " " + std::to_string(c.xform_lo[i]) + " " + std::to_string(c.xform_hi[i]);
    o += "\n";
  }
  if (!maps.empty()) { // version 4: segment-id maps — column, coordinates, symbol, symbol slot, mul0..mul3, div
    o += "segmaps " + std::to_string(maps.size()) + "\n";
    for (size_t k : maps) {
      const Column &c = p.columns[k];
      o += std::to_string(k) + " " + std::to_string(c.seg_mul.size()) + " " + std::to_string(c.seg_sym) + " " + std::to_string(c.seg_sym_slot);
      for (size_t i = 0; i < 4; ++i) o += " " + std::to_string(i < c.seg_mul.size() ? c.seg_mul[i] : 0);
      o += " " + std::to_string(c.seg_div) + "\n";
    }
  }
  if (stage) {
    o += "stage " + std::to_string(stage->modes.size()) + " symbols_input " + std::to_string(stage->symbols_input) + "\n";
    for (size_t i = 0; i < stage->modes.size(); ++i) o += std::to_string(stage->modes[i]) + " " + std::to_string(stage->rows_symbol[i]) + "\n";
  }
  return o;
}

// ====================================================================================================
// the plan builder (recom_amd/graph/plan_builder.py)
// ====================================================================================================
struct IndexSource { // where an index operand really comes from (EmitInputInline)
  std::string tensor; // graph tensor that ConcatInputs receives
  int dtype = 0, rank = 0, stride = 1;
  bool has_boundaries = false;
  std::vector<float> boundaries;
  int xform_mode = FCP_XFORM_NONE;
  std::vector<int64_t> xform_lo, xform_hi;
  int64_t xform_substitute = 0, hash_buckets = 0;
  std::string filter_node; // the Gather* node whose (indices, values) pair this operand belongs to ("" = none)
  bool generated_rows = false;
  // a SparseReshape folded into the index expression (cuda_emitter.cc:1874-1916): `reshaped` while the operand is the op's
  // whole output_indices matrix; the [:, 0:1] slice turns it into the segment-id map (seg_mul non-empty)
  bool reshaped = false, has_seg_sym = false;
  int map_rank = 0;
  std::vector<int64_t> seg_mul;
  int64_t seg_div = 1;
  std::pair<std::string, int> seg_sym; // (tensor, flat index) whose value multiplies one factor
  int seg_sym_slot = 0;
};
struct SymbolDef {
  std::string tensor;
  int index;
  int last_stride = 0; // > 0: the row count of a plain SparseSegmentSum / Mean: max(reshape(tensor, [-1])[::last_stride], -1) + 1
};
struct HostInput {
  std::string tensor;
  int dtype, rank;
};
struct ColumnInfo {
  std::string value_tensor, concat_input;
  int concat_index;
};
struct GroupInfo {
  std::string concat_node;
  int dtype, n_inputs;
  std::vector<int> columns;
};
struct BuiltPlan {
  Plan spec;
  std::vector<HostInput> host_inputs, device_inputs;
  std::vector<SymbolDef> symbols;
  std::vector<GroupInfo> groups;
  std::vector<ColumnInfo> columns;
  std::vector<std::pair<std::string, std::string>> skipped;
};

bool reshape_like(const std::string &op) { return op == "Reshape" || op == "ExpandDims" || op == "Squeeze"; } // IsReshape, cuda_emitter.cc:62-73

struct PlanBuilder {
  const GraphView &g;
  bool external; // host_concat == "external"
  std::map<std::string, std::pair<int64_t, int64_t>> tables;
  std::map<std::string, int> host_ix, dev_ix;
  std::vector<HostInput> host_list, dev_list;
  std::map<std::pair<std::string, int>, int> sym_ix;
  std::vector<SymbolDef> sym_list;

  PlanBuilder(const GraphView &view, bool ext) : g(view), external(ext) { find_tables(); }

  // ---- tables (graph_info.cc:209-259) ------------------------------------------------------------
  void find_tables() {
    for (const Node &n : g.nodes) {
      if (n.op != "VariableV2" && n.op != "Const" && n.op != "VarHandleOp") continue;
      Shape shape;
      if (n.op == "VarHandleOp") {
        if (attr_type(n, "dtype") != DT_FLOAT) continue;
        const Field *s = find_field(attr(n, "shape"), 7);
        shape = s ? shape_of_proto(parse_message(s->bytes)) : Shape(std::vector<Dim>());
      } else {
        int dt = 0;
        try {
          dt = g.out_dtype(n);
        } catch (const Unsupported &) {
          continue;
        }
        if (dt != DT_FLOAT) continue;
        shape = g.static_shape({&n, 0});
      }
      if (graph_debug()) fprintf(stderr, "fcp_graph: table candidate %s (%s): rank %d\n", n.name.c_str(), n.op.c_str(), shape ? (int)shape->size() : -1);
      if (!shape || shape->size() != 2 || !GraphView::all_known(*shape) || std::min(*(*shape)[0], *(*shape)[1]) <= 0) continue;
      std::vector<std::string> stack{n.name};
      int lookups = 0;
      bool ok = true;
      while (!stack.empty() && ok) {
        const std::string cur = stack.back();
        stack.pop_back();
        for (const auto &[c, i] : g.data_consumers(cur)) {
          if (c->op == "Identity" || c->op == "ReadVariableOp") stack.push_back(c->name);
          else if (c->op == "Assign" || c->op == "SaveV2" || c->op == "AssignVariableOp" || c->op == "VarIsInitializedOp") continue;
          else if ((c->op.find("Gather") != std::string::npos || c->op.find("SparseSegment") != std::string::npos) && i == 0) ++lookups;
          else {
            ok = false;
            break;
          }
        }
      }
      if (graph_debug()) fprintf(stderr, "fcp_graph:   -> ok %d lookups %d\n", (int)ok, lookups);
      if (ok && lookups) tables[n.name] = {*(*shape)[0], *(*shape)[1]};
    }
  }
  struct Table {
    std::string name;
    int64_t vocab, dim;
  };
  Table table_of(NodeRef r) const {
    const Node *n = r.first;
    int port = r.second;
    while ((n->op == "Identity" || n->op == "ReadVariableOp") && port == 0) std::tie(n, port) = g.input(*n, 0);
    auto it = tables.find(n->name);
    if (it == tables.end() || port != 0) throw Unsupported(n->name + " (" + n->op + ") is not an embedding table");
    return {n->name, it->second.first, it->second.second};
  }

  // ---- operand bookkeeping ------------------------------------------------------------------------
  int host_input(const std::string &tensor, int dtype, int rank) {
    auto it = host_ix.find(tensor);
    if (it != host_ix.end()) return it->second;
    host_ix[tensor] = (int)host_list.size();
    host_list.push_back({tensor, dtype, rank});
    return (int)host_list.size() - 1;
  }
  int device_input(const std::string &tensor) {
    auto it = dev_ix.find(tensor);
    if (it != dev_ix.end()) return it->second;
    std::string name = tensor;
    const Node *n = g.node(tensor);
    if (n && n->op == "VarHandleOp") { // a resource variable's handle is not its data
      name = tensor + "/fcp_read";
      for (const auto &[c, i] : g.data_consumers(tensor))
        if (c->op == "ReadVariableOp" && i == 0) {
          name = c->name;
          break;
        }
    }
    dev_ix[tensor] = (int)dev_list.size();
    dev_list.push_back({name, DT_FLOAT, 2});
    return (int)dev_list.size() - 1;
  }
  int symbol(const std::string &tensor, int index, int last_stride = 0) {
    auto key = std::make_pair(tensor + (last_stride ? "\x01last" + std::to_string(last_stride) : std::string()), index);
    auto it = sym_ix.find(key);
    if (it != sym_ix.end()) return it->second;
    sym_ix[key] = (int)sym_list.size();
    sym_list.push_back({tensor, index, last_stride});
    return (int)sym_list.size() - 1;
  }

  // ---- EmitInputInline (cuda_emitter.cc:1769-1949) ------------------------------------------------
  IndexSource terminal(NodeRef r) const {
    const int dtype = g.out_dtype(*r.first, r.second);
    const Shape s = g.static_shape(r);
    if (!s) throw Unsupported("rank of " + r.first->name + " unknown");
    IndexSource src;
    src.tensor = tensor_name(r.first->name, r.second);
    src.dtype = dtype;
    src.rank = (int)s->size();
    return src;
  }
  IndexSource trace_index(NodeRef r) const {
    try {
      return trace_inline(r);
    } catch (const Unsupported &) {
      return terminal(r);
    }
  }
  static bool is_int(int t) { return t == DT_INT32 || t == DT_INT64; }

  IndexSource trace_inline(NodeRef r) const {
    const Node &n = *r.first;
    const int port = r.second;
    if (n.op.rfind("Addons>", 0) == 0) return trace_id_filter(r);
    if (port != 0) throw Unsupported("not an inlinable op");
    if (n.op == "StringToHashBucketFast") {
      const NodeRef a = g.input(n, 0);
      if (a.first->op != "AsString" || a.second != 0) throw Unsupported("StringToHashBucketFast over a string tensor");
      const Node &an = *a.first;
      if (!is_int(attr_type(an, "T")) || (has_attr(an, "width") && attr_i(an, "width") != -1 && attr_i(an, "width") != 0) ||
          (has_attr(an, "fill") && !attr_s(an, "fill").empty()) || (has_attr(an, "scientific") && attr_b(an, "scientific")) ||
          (has_attr(an, "shortest") && attr_b(an, "shortest")))
        throw Unsupported("AsString with formatting");
      IndexSource src = trace_index(g.input(an, 0));
      if (src.has_boundaries || src.xform_mode != FCP_XFORM_NONE || src.hash_buckets || src.stride != 1 || !src.filter_node.empty() || !is_int(src.dtype))
        throw Unsupported("hash of a transformed id stream");
      src.hash_buckets = attr_i(n, "num_buckets");
      return src;
    }
    if (n.op == "SparseReshape") {
      const ReshapeMap m = sparse_reshape_map(n);
      IndexSource src = trace_index(g.input(n, 0));
      if (src.reshaped || src.stride != 1 || src.has_boundaries || src.rank != 2) throw Unsupported("SparseReshape over a transformed index matrix");
      src.reshaped = true;
      src.map_rank = m.rank;
      src.seg_mul = m.mul;
      src.seg_div = m.div;
      src.has_seg_sym = m.has_sym;
      src.seg_sym = m.sym;
      src.seg_sym_slot = m.slot;
      return src;
    }
    if (reshape_like(n.op) || n.op == "Identity") return trace_index(g.input(n, 0)); // flat element index unchanged
    if (n.op == "Cast") {
      IndexSource src = trace_index(g.input(n, 0));
      const int dst = attr_type(n, "DstT");
      if (is_int(dst) && (is_int(src.dtype) || src.has_boundaries)) return src;
      throw Unsupported("cast changes the value");
    }
    if (n.op == "Bucketize") {
      IndexSource src = trace_index(g.input(n, 0));
      if (src.dtype != DT_FLOAT || src.has_boundaries || src.stride != 1) throw Unsupported("Bucketize over a non-float32 operand");
      const std::vector<float> b = repeated_fixed<float>(attr_list(n, "boundaries"), 4);
      if (b.empty()) throw Unsupported("Bucketize without boundaries");
      src.boundaries = b;
      src.has_boundaries = true;
      return src;
    }
    if (n.op == "StridedSlice") {
      // the one case the reference inlines: [n, k] -> column 0 (:1864-1873)
      const std::optional<StridedSliceSpec> spec = g.strided_slice_spec(n);
      const NodeRef in = g.input(n, 0);
      const Shape in_shape = g.static_shape(in);
      if (!spec || !in_shape || in_shape->size() != 2 || !(*in_shape)[1]) throw Unsupported("StridedSlice operand shape unknown");
      if (spec->ellipsis_mask || spec->new_axis_mask || (spec->shrink_axis_mask != 0 && spec->shrink_axis_mask != 2)) throw Unsupported("StridedSlice masks");
      if (spec->begin.size() != 2 || spec->strides != std::vector<int64_t>{1, 1}) throw Unsupported("StridedSlice is not a 2-D unit-stride slice");
      if (!(((spec->begin_mask & 1) || spec->begin[0] == 0) && (spec->end_mask & 1))) throw Unsupported("StridedSlice does not keep all rows");
      if ((spec->begin_mask & 2) || (spec->end_mask & 2) || spec->begin[1] != 0 || spec->end[1] != 1) throw Unsupported("StridedSlice does not select column 0");
      IndexSource src = trace_index(in);
      if (src.has_boundaries) throw Unsupported("slice of bucketized values");
      if (src.reshaped) { // column 0 of a SparseReshape's output: the row coordinate of the map
        src.reshaped = false;
        src.stride *= src.map_rank;
        if (src.seg_mul == std::vector<int64_t>{1} && src.seg_div == 1 && !src.has_seg_sym) src.seg_mul.clear(); // the identity on idx0 needs no map
        return src;
      }
      src.stride *= (int)*(*in_shape)[1];
      return src;
    }
    throw Unsupported("not an inlinable op");
  }

  // the CPU id ops PreLookupOptimizer leaves in front of a lookup (pre_lookup_optimizer.cc:596-654)
  IndexSource trace_id_filter(NodeRef r) const {
    const Node &n = *r.first;
    const int port = r.second;
    const std::vector<int64_t> lo = repeated_varints(attr_list(n, "left_boundaries"), 3), hi = repeated_varints(attr_list(n, "right_boundaries"), 3);
    if (lo.size() != hi.size()) throw Unsupported("malformed interval attrs");
    for (size_t i = 0; i < lo.size(); ++i)
      if (lo[i] > hi[i]) throw Unsupported("malformed interval attrs");
    auto absorb = [&](int k, int mode, int64_t sub) {
      const NodeRef in = g.input(n, k);
      IndexSource src = trace_index(in);
      if (src.xform_mode != FCP_XFORM_NONE || src.stride != 1 || src.generated_rows || !src.filter_node.empty()) src = terminal(in);
      src.xform_mode = mode;
      src.xform_lo = lo;
      src.xform_hi = hi;
      src.xform_substitute = sub;
      return src;
    };
    if (n.op == "Addons>SelectValue" && port == 0) return absorb(0, FCP_XFORM_SELECT, attr_i(n, "substitute"));
    if (n.op == "Addons>GatherIndiceValue") {
      IndexSource src;
      if (port == 1) src = absorb(1, FCP_XFORM_FILTER, 0);
      else if (port == 0) {
        src = trace_index(g.input(n, 0));
        if (!src.filter_node.empty() || src.xform_mode != FCP_XFORM_NONE) src = terminal(g.input(n, 0));
      } else
        throw Unsupported("not an inlinable op");
      src.filter_node = n.name;
      return src;
    }
    if (n.op == "Addons>GatherValueGenIndice") {
      IndexSource src;
      if (port == 1) src = absorb(0, FCP_XFORM_FILTER, 0);
      else if (port == 0) {
        src = terminal(g.input(n, 0));
        src.generated_rows = true;
      } else
        throw Unsupported("not an inlinable op");
      src.filter_node = n.name;
      return src;
    }
    throw Unsupported("not an inlinable op");
  }

  // SparseReshape(indices [nnz, r], shape [r], new_shape [q]): the row coordinate of the reshaped element is
  // (sum_k idx_k * prod(shape[k+1:])) / prod(new_shape[1:]) (the reference's flat-index algebra, cuda_emitter.cc:1874-1916,
  // offset 0).  Every shape entry must trace to a constant or to a copy of a tensor element; a trailing coordinate drops
  // out when its dimension divides the denominator (idx < dim); what remains may hold ONE run-time factor (a symbol).
  struct ReshapeMap {
    int rank = 0;
    std::vector<int64_t> mul;
    int64_t div = 1;
    bool has_sym = false;
    std::pair<std::string, int> sym;
    int slot = 0;
  };
  ReshapeMap sparse_reshape_map(const Node &n) const {
    const NodeRef shape = g.input(n, 1), nw = g.input(n, 2);
    const Shape s1 = g.static_shape(shape), s2 = g.static_shape(nw);
    auto rank1 = [](const Shape &s) { return s && s->size() == 1 && (*s)[0]; };
    if (!rank1(s1) || !rank1(s2)) throw Unsupported("SparseReshape: ranks unknown");
    const int r = (int)*(*s1)[0], q = (int)*(*s2)[0];
    if (r < 1 || r > 4 || q < 1) throw Unsupported("SparseReshape: more than 4 input coordinates");
    std::vector<ElemSource> ins, outs;
    for (int k = 1; k < r; ++k) {
      const std::optional<ElemSource> e = g.elem_source(shape, k);
      if (!e || (e->is_const && e->value <= 0)) throw Unsupported("SparseReshape: a dimension is computed (or inferred at run time)");
      ins.push_back(*e);
    }
    for (int k = 1; k < q; ++k) {
      const std::optional<ElemSource> e = g.elem_source(nw, k);
      if (!e || (e->is_const && e->value <= 0)) throw Unsupported("SparseReshape: a dimension is computed (or inferred at run time)");
      outs.push_back(*e);
    }
    using Sym = std::pair<std::string, int>;
    struct Factor {
      int64_t c = 1;
      std::vector<Sym> syms;
    };
    auto product = [](const std::vector<ElemSource> &v, size_t from) {
      Factor f;
      for (size_t i = from; i < v.size(); ++i) {
        if (v[i].is_const) f.c *= v[i].value;
        else f.syms.push_back({v[i].tensor, (int)v[i].index});
      }
      return f;
    };
    std::vector<Factor> mul;
    for (int k = 0; k < r; ++k) mul.push_back(product(ins, (size_t)k)); // multiplier of coordinate k: prod(shape[k+1:])
    Factor div = product(outs, 0);
    auto remove_one = [](std::vector<Sym> &v, const Sym &key) { v.erase(std::find(v.begin(), v.end(), key)); };
    int nk = r;
    while (nk >= 2) { // drop trailing coordinates whose dimension divides `div`
      const ElemSource &last = ins[(size_t)nk - 2]; // shape[nk-1]
      if (last.is_const) {
        if (div.c % last.value) break;
        div.c /= last.value;
        for (int k = 0; k < nk - 1; ++k) mul[k].c /= last.value;
      } else {
        const Sym key{last.tensor, (int)last.index};
        if (std::find(div.syms.begin(), div.syms.end(), key) == div.syms.end()) break;
        remove_one(div.syms, key);
        for (int k = 0; k < nk - 1; ++k) remove_one(mul[k].syms, key);
      }
      --nk;
    }
    mul.resize((size_t)nk);
    ReshapeMap m;
    m.rank = r;
    int runtime = 0;
    for (int k = 0; k <= nk; ++k) {
      const Factor &f = k < nk ? mul[k] : div;
      if (f.syms.empty()) continue;
      if (++runtime > 1 || f.syms.size() != 1) throw Unsupported("SparseReshape: more than one run-time factor");
      m.has_sym = true;
      m.sym = f.syms[0];
      m.slot = k < nk ? k : 4;
    }
    for (const Factor &f : mul) m.mul.push_back(f.c);
    m.div = div.c;
    return m;
  }

  struct IdsOperand {
    int host, id_source;
    bool has_boundaries;
    std::vector<float> boundaries;
    int xform_mode;
    std::vector<int64_t> lo, hi;
    int64_t sub, hash_buckets;
    std::string filter_node;
  };
  IdsOperand ids_operand(NodeRef r) {
    IndexSource src = trace_index(r);
    if (src.stride != 1 || src.generated_rows || src.reshaped) src = terminal(r); // no strided id source in the column record
    int id_source;
    if (src.has_boundaries) id_source = FCP_IDS_F32_BUCKETIZE;
    else if (src.dtype == DT_INT32) id_source = FCP_IDS_I32;
    else if (src.dtype == DT_INT64) id_source = FCP_IDS_I64;
    else throw Unsupported("ids tensor " + src.tensor + " has dtype " + std::to_string(src.dtype));
    return {host_input(src.tensor, src.dtype, src.rank), id_source, src.has_boundaries, src.boundaries, src.xform_mode, src.xform_lo, src.xform_hi,
            src.xform_substitute, src.hash_buckets, src.filter_node};
  }
  struct SegOperand {
    int host, kind, stride;
    std::vector<int64_t> seg_mul;
    int64_t seg_div = 1;
    int seg_sym = -1, seg_sym_slot = 0;
  };
  SegOperand seg_operand(NodeRef r, const std::string &filter_node, bool allow_map = false) {
    IndexSource src = trace_index(r);
    if (src.has_boundaries || src.xform_mode != FCP_XFORM_NONE || src.generated_rows || src.filter_node != filter_node || src.reshaped ||
        (!src.seg_mul.empty() && !allow_map)) {
      if (!filter_node.empty()) throw Unsupported("segment ids do not come from the id filter's indices output");
      src = terminal(r);
    }
    int kind;
    if (src.dtype == DT_INT32) kind = FCP_SEG_IDS_I32;
    else if (src.dtype == DT_INT64) kind = FCP_SEG_IDS_I64;
    else throw Unsupported("segment ids " + src.tensor + " have dtype " + std::to_string(src.dtype));
    SegOperand o{0, kind, src.stride, {}, 1, -1, 0};
    if (!src.seg_mul.empty()) { // a SparseReshape folded in: the row coordinate as an expression of the original ones
      o.seg_mul = src.seg_mul;
      o.seg_div = src.seg_div;
      o.seg_sym_slot = src.seg_sym_slot;
      o.seg_sym = src.has_seg_sym ? symbol(src.seg_sym.first, src.seg_sym.second) : -1;
    }
    o.host = host_input(src.tensor, src.dtype, src.rank);
    return o;
  }

  static void apply_ids(Column &c, const IdsOperand &o) {
    c.id_source = o.id_source;
    c.ids_input = o.host;
    c.has_boundaries = o.has_boundaries;
    c.boundaries = o.boundaries;
    c.xform_mode = o.xform_mode;
    c.xform_lo = o.lo;
    c.xform_hi = o.hi;
    c.xform_substitute = o.sub;
    c.hash_buckets = o.hash_buckets;
  }

  // ---- EmitSubgraphCode dispatch (cuda_emitter.cc:1096-1152) ---------------------------------------
  struct GatherMatch {
    Table table;
    IdsOperand ids;
  };
  GatherMatch match_gather(const Node &n) {
    const std::optional<Array> axis = g.const_array(g.input(n, 2));
    if (!axis || axis->v.empty() || (int64_t)axis->v[0] != 0) throw Unsupported("GatherV2 axis is not 0");
    if (has_attr(n, "batch_dims") && attr_i(n, "batch_dims") != 0) throw Unsupported("GatherV2 batch_dims");
    const Table t = table_of(g.input(n, 0));
    return {t, ids_operand(g.input(n, 1))};
  }

  Column match_column(NodeRef r, int group, int slot) {
    const Node &n = *r.first;
    if (r.second != 0) throw Unsupported("value is not output 0");
    Column c;
    c.concat_group = group;
    c.concat_slot = slot;
    if (n.op == "ResourceGather" || n.op == "GatherV2") { // EmitGatherRows :1246-1330 (ResourceGather: GatherV2 over a resource variable)
      Table t;
      std::optional<IdsOperand> ids;
      if (n.op == "ResourceGather") {
        if (has_attr(n, "batch_dims") && attr_i(n, "batch_dims") != 0) throw Unsupported("ResourceGather batch_dims");
        t = table_of(g.input(n, 0));
        ids = ids_operand(g.input(n, 1));
      } else {
        GatherMatch m = match_gather(n);
        t = m.table;
        ids = m.ids;
      }
      if (!ids->filter_node.empty()) throw Unsupported(n.op + " over filtered values"); // compacted values without their indices: rows are lost
      c.form = FCP_FORM_GATHER;
      c.dim = (int)t.dim;
      c.vocab = t.vocab;
      c.combiner = FCP_COMBINER_NONE;
      c.table_input = device_input(t.name);
      apply_ids(c, *ids);
      c.rows_source = FCP_ROWS_FROM_IDS;
      return c;
    }
    if (n.op == "SparseSegmentSumWithNumSegments" || n.op == "SparseSegmentMeanWithNumSegments") { // EmitSparseSegmentReduce* :1444-1760
      const Table t = table_of(g.input(n, 0));
      const IdsOperand ids = ids_operand(g.input(n, 1));
      const SegOperand seg = seg_operand(g.input(n, 2), ids.filter_node, /*allow_map=*/true);
      NodeRef nn = g.input(n, 3);
      while (reshape_like(nn.first->op) && nn.second == 0) nn = g.input(*nn.first, 0); // Squeeze(num_segments) lookup_optimizer.cc:248-254
      const int sym = symbol(tensor_name(nn.first->name, nn.second), 0);
      c.form = FCP_FORM_SEGMENT_REDUCE;
      c.dim = (int)t.dim;
      c.vocab = t.vocab;
      c.combiner = n.op == "SparseSegmentSumWithNumSegments" ? FCP_COMBINER_SUM : FCP_COMBINER_MEAN;
      c.table_input = device_input(t.name);
      apply_ids(c, ids);
      c.seg_input = seg.host;
      c.seg_kind = seg.kind;
      c.seg_stride = seg.stride;
      c.seg_mul = seg.seg_mul;
      c.seg_div = seg.seg_div;
      c.seg_sym = seg.seg_sym;
      c.seg_sym_slot = seg.seg_sym_slot;
      c.rows_source = FCP_ROWS_FROM_SYMBOL;
      c.rows_arg = sym;
      return c;
    }
    if (n.op == "SparseSegmentSum" || n.op == "SparseSegmentMean") {
      // no num_segments (the emitter takes these too, cuda_emitter.cc:1096-1113; its row count is a SymEngine symbol,
      // :1444-1622): rows = last segment id + 1, computed by the rewritten graph from the sorted segment ids the host ships
      const Table t = table_of(g.input(n, 0));
      const IdsOperand ids = ids_operand(g.input(n, 1));
      if (!ids.filter_node.empty()) throw Unsupported("row count of a plain SparseSegment op over filtered ids depends on what the filter keeps");
      const SegOperand seg = seg_operand(g.input(n, 2), std::string());
      const int sym = symbol(host_list[seg.host].tensor, 0, seg.stride);
      c.form = FCP_FORM_SEGMENT_REDUCE;
      c.dim = (int)t.dim;
      c.vocab = t.vocab;
      c.combiner = n.op == "SparseSegmentSum" ? FCP_COMBINER_SUM : FCP_COMBINER_MEAN;
      c.table_input = device_input(t.name);
      apply_ids(c, ids);
      c.seg_input = seg.host;
      c.seg_kind = seg.kind;
      c.seg_stride = seg.stride;
      c.rows_source = FCP_ROWS_FROM_SYMBOL;
      c.rows_arg = sym;
      return c;
    }
    if (n.op == "ScatterNd") { // EmitGatherScatterRows :1332-1442
      const NodeRef upd = g.input(n, 1);
      if (upd.first->op != "GatherV2" || upd.second != 0) throw Unsupported("ScatterNd updates are not a GatherV2");
      const GatherMatch m = match_gather(*upd.first);
      const IndexSource rows = trace_index(g.input(n, 0));
      c.dim = (int)m.table.dim;
      c.vocab = m.table.vocab;
      c.combiner = FCP_COMBINER_NONE;
      if (!m.ids.filter_node.empty() && rows.generated_rows && rows.filter_node == m.ids.filter_node) {
        // ScatterNd(GatherValueGenIndice:0, GatherV2(table, GatherValueGenIndice:1)): a one-hot gather whose dropped ids leave zero rows
        c.form = FCP_FORM_GATHER;
        c.table_input = device_input(m.table.name);
        apply_ids(c, m.ids);
        c.rows_source = FCP_ROWS_FROM_IDS;
        return c;
      }
      const SegOperand seg = seg_operand(g.input(n, 0), m.ids.filter_node);
      const NodeRef shp = g.input(n, 2);
      const int sym = symbol(tensor_name(shp.first->name, shp.second), 0);
      c.form = FCP_FORM_GATHER_SCATTER;
      c.table_input = device_input(m.table.name);
      apply_ids(c, m.ids);
      c.seg_input = seg.host;
      c.seg_kind = seg.kind;
      c.seg_stride = seg.stride;
      c.rows_source = FCP_ROWS_FROM_SYMBOL;
      c.rows_arg = sym;
      return c;
    }
    if (n.op == "Sum") { // EmitBatchColReduction :1180-1244
      const std::optional<Array> axis = g.const_array(g.input(n, 1));
      if (!axis || axis->v.size() != 1 || (int64_t)axis->v[0] != 1) throw Unsupported("Sum is not over axis 1");
      if (has_attr(n, "keep_dims") && attr_b(n, "keep_dims")) throw Unsupported("Sum keep_dims");
      const NodeRef x = g.input(n, 0);
      const Shape s = g.static_shape(x);
      if (g.out_dtype(*x.first, x.second) != DT_FLOAT || !s || s->size() != 3 || !(*s)[2]) throw Unsupported("Sum operand is not a float32 [b, r, c] tensor with static c");
      const int i = host_input(tensor_name(x.first->name, x.second), DT_FLOAT, 3);
      c.form = FCP_FORM_BATCH_COL_REDUCTION;
      c.dim = (int)*(*s)[2];
      c.id_source = FCP_IDS_I32;
      c.ids_input = i;
      c.rows_source = FCP_ROWS_FROM_INPUT_DIM0;
      c.rows_arg = i;
      return c;
    }
    throw Unsupported("op " + n.op + " is not a lookup");
  }

  Column host_column(const std::string &tensor, int group, int slot) { // passthrough / external: a concat input that is not a lookup
    const auto [name, port] = split_tensor(tensor);
    const Node *n = g.node(name);
    const Shape s = n ? g.static_shape({n, port}) : Shape();
    if (!n || g.out_dtype(*n, port) != DT_FLOAT || !s || s->size() != 2 || !(*s)[1]) throw Unsupported(tensor + ": not a float32 [rows, dim] tensor with static dim");
    Column c;
    c.dim = (int)*(*s)[1];
    c.id_source = FCP_IDS_I32;
    c.concat_group = group;
    c.concat_slot = slot;
    if (external) { // the reference's wiring: the tensor reaches Addons>ConcatOutputs as a host input, the plan only reserves its slot
      c.form = FCP_FORM_EXTERNAL;
      c.rows_source = FCP_ROWS_FROM_GROUP;
    } else {
      const int i = host_input(tensor, DT_FLOAT, 2);
      c.form = FCP_FORM_PASSTHROUGH;
      c.ids_input = i;
      c.rows_source = FCP_ROWS_FROM_INPUT_DIM0;
      c.rows_arg = i;
    }
    return c;
  }

  std::vector<std::string> upstream_tables(const Node &start, size_t limit = 256) const {
    std::set<std::string> seen;
    std::vector<const Node *> stack{&start};
    std::vector<std::string> found;
    while (!stack.empty() && seen.size() < limit) {
      const Node *n = stack.back();
      stack.pop_back();
      if (!seen.insert(n->name).second) continue;
      if (tables.count(n->name)) found.push_back(n->name);
      const int k = (int)g.data_inputs(*n).size();
      for (int i = 0; i < k; ++i) stack.push_back(g.input(*n, i).first);
    }
    return found;
  }

  // ---- the walk -----------------------------------------------------------------------------------
  BuiltPlan build() {
    BuiltPlan out;
    std::vector<Column> columns;
    for (const Node &concat : g.nodes) {
      if (concat.op != "ConcatV2") continue;
      const std::vector<std::string> ins = g.data_inputs(concat);
      const int n = has_attr(concat, "N") ? (int)attr_i(concat, "N") : (int)ins.size() - 1;
      if (n < 0 || n >= (int)ins.size()) continue;
      const std::optional<Array> axis = g.const_array(g.input(concat, n));
      int dtype = 0;
      try {
        dtype = g.out_dtype(concat);
      } catch (const Unsupported &) {
        continue;
      }
      if (!axis || axis->v.empty() || ((int64_t)axis->v[0] != 1 && (int64_t)axis->v[0] != -1) || dtype != DT_FLOAT) {
        if (graph_debug()) fprintf(stderr, "fcp_graph: concat %s skipped (axis const %d, dtype %d)\n", concat.name.c_str(), axis ? 1 : 0, dtype);
        continue;
      }
      // snapshot: a group that turns out unusable must not leave operands behind
      const auto snap = std::make_tuple(host_ix, host_list, dev_ix, dev_list, sym_ix, sym_list);
      const int group = (int)out.groups.size();
      std::vector<Column> cols;
      std::vector<ColumnInfo> cinfo;
      int lookups = 0;
      try {
        for (int i = 0; i < n; ++i) {
          NodeRef r = g.input(concat, i);
          while (reshape_like(r.first->op) && r.second == 0) r = g.input(*r.first, 0); // FindFCOutputs :1060-1066
          Column col;
          std::string value;
          try {
            col = match_column(r, group, i);
            ++lookups;
            value = tensor_name(r.first->name, r.second);
          } catch (const Unsupported &why) {
            if (graph_debug()) fprintf(stderr, "fcp_graph: %s input %d (%s, %s): %s\n", concat.name.c_str(), i, r.first->name.c_str(), r.first->op.c_str(), why.what());
            col = host_column(ins[i], group, i);
            value = ins[i];
            if (tables.count(r.first->name) || !upstream_tables(*r.first).empty()) out.skipped.push_back({r.first->name, why.what()});
          }
          cols.push_back(col);
          cinfo.push_back({value, ins[i], i});
        }
        if (lookups == 0) throw Unsupported("no lookup column converges here");
      } catch (const Unsupported &why) {
        if (graph_debug()) fprintf(stderr, "fcp_graph: concat %s given up: %s\n", concat.name.c_str(), why.what());
        std::tie(host_ix, host_list, dev_ix, dev_list, sym_ix, sym_list) = snap;
        if (lookups) out.skipped.push_back({concat.name, why.what()});
        continue;
      }
      GroupInfo gi{concat.name, dtype, n, {}};
      for (int i = 0; i < n; ++i) gi.columns.push_back((int)columns.size() + i);
      columns.insert(columns.end(), cols.begin(), cols.end());
      out.columns.insert(out.columns.end(), cinfo.begin(), cinfo.end());
      out.groups.push_back(gi);
    }
    if (out.groups.empty()) throw Unsupported("no ConcatV2 with embedding lookups found");
    out.spec.columns = columns;
    for (const HostInput &h : host_list) {
      out.spec.host_ranks.push_back(h.rank);
      if (h.dtype != DT_FLOAT && h.dtype != DT_INT32 && h.dtype != DT_INT64) throw Unsupported("host input " + h.tensor + " has an unsupported dtype");
      out.spec.host_esizes.push_back(h.dtype == DT_INT64 ? 8 : 4);
    }
    out.spec.n_device_inputs = (int)dev_list.size();
    out.spec.n_groups = (int)out.groups.size();
    out.spec.n_symbols = (int)sym_list.size();
    validate_plan(out.spec);
    out.host_inputs = host_list;
    out.device_inputs = dev_list;
    out.symbols = sym_list;
    return out;
  }
};

std::string describe(const BuiltPlan &b) {
  static const char *names[] = {"", "gather", "segment-reduce", "gather-scatter", "passthrough", "batch-col-reduction", "external (ConcatOutputs host input)"};
  std::string o = std::to_string(b.groups.size()) + " concat group(s), " + std::to_string(b.spec.columns.size()) + " column(s), " +
                  std::to_string(b.host_inputs.size()) + " host input(s), " + std::to_string(b.device_inputs.size()) + " table(s), " +
                  std::to_string(b.symbols.size()) + " symbol(s)";
  for (size_t g = 0; g < b.groups.size(); ++g) {
    std::map<std::string, int> forms;
    int width = 0;
    for (int k : b.groups[g].columns) {
      ++forms[names[b.spec.columns[k].form]];
      width += b.spec.columns[k].dim;
    }
    o += "\n  group " + std::to_string(g) + ": " + b.groups[g].concat_node + "  width " + std::to_string(width) + "  ";
    bool first = true;
    for (const auto &[k, v] : forms) {
      o += (first ? "" : ", ") + std::to_string(v) + "x " + k;
      first = false;
    }
  }
  for (const auto &[node, why] : b.skipped) o += "\n  skipped " + node + ": " + why;
  return o;
}

// ====================================================================================================
// the graph rewrite (recom_amd/graph/rewrite.py; CudaEmitter::Rewrite, cuda_emitter.cc:2496-2656)
// ====================================================================================================
// AttrValue builders.  oneof members are written even when they hold the default value (the field is "set").
Message av_type(int t) {
  Message m;
  add_varint(m, 6, (uint64_t)t);
  return m;
}
Message av_i(int64_t v) {
  Message m;
  add_varint(m, 3, (uint64_t)v);
  return m;
}
Message av_b(bool v) {
  Message m;
  add_varint(m, 5, v ? 1 : 0);
  return m;
}
Message av_s(const std::string &s) {
  Message m;
  add_bytes(m, 2, s);
  return m;
}
Message av_list_types(const std::vector<int64_t> &t) {
  Message l, m;
  if (!t.empty()) add_bytes(l, 6, packed_varints(t));
  add_bytes(m, 1, serialize(l));
  return m;
}
Message av_list_ints(const std::vector<int64_t> &v) {
  Message l, m;
  if (!v.empty()) add_bytes(l, 3, packed_varints(v));
  add_bytes(m, 1, serialize(l));
  return m;
}
Message av_tensor_i32(const std::vector<int32_t> &values, bool scalar) { // numpy_to_tensor: dtype, dims, tensor_content
  Message t;
  add_varint(t, 1, DT_INT32);
  if (!scalar) {
    Message dim, shape;
    add_varint(dim, 1, values.size());
    add_bytes(shape, 2, serialize(dim));
    add_bytes(t, 2, serialize(shape));
  }
  add_bytes(t, 4, std::string(reinterpret_cast<const char *>(values.data()), values.size() * 4));
  Message m;
  add_bytes(m, 8, serialize(t));
  return m;
}

struct OutNode { // a node of the rewritten graph: an original one (raw bytes kept) or a new one
  std::string name, op;
  std::vector<std::string> input;
  std::vector<std::pair<std::string, Message>> attrs; // new nodes only, in creation order
  const Node *orig = nullptr;
  bool renamed = false;
};

std::string serialize_node(const OutNode &n) {
  if (n.orig) {
    if (!n.renamed) return serialize(n.orig->raw);
    Message m = n.orig->raw;
    bool done = false;
    for (Field &f : m.f)
      if (f.num == 1 && f.wt == 2) {
        f.bytes = n.name;
        done = true;
      }
    if (!done) {
      Field f;
      f.num = 1;
      f.wt = 2;
      f.bytes = n.name;
      m.f.insert(m.f.begin(), f);
    }
    return serialize(m);
  }
  Message m;
  add_bytes(m, 1, n.name);
  add_bytes(m, 2, n.op);
  for (const std::string &i : n.input) add_bytes(m, 3, i);
  for (const auto &[k, v] : n.attrs) {
    Message e;
    add_bytes(e, 1, k);
    add_bytes(e, 2, serialize(v));
    add_bytes(m, 5, serialize(e));
  }
  return serialize(m);
}

constexpr int kBlockThreads = 64; // CudaEmitter(graph_info, 1 << 28, 64), fc_optimize_pass.cc:71

std::string rewrite_graph(const Message &graph, const GraphView &view, const BuiltPlan &built, const std::string &plan_path, bool prune,
                          const StageInfo *stage) {
  for (const char *reserved : {"ConcatInputs", "FeatureColumnProcess"})
    if (view.node(reserved)) throw std::invalid_argument(std::string("graph already has a node named ") + reserved);
  std::vector<OutNode> nodes;
  for (const Node &n : view.nodes) {
    OutNode o;
    o.name = n.name;
    o.op = n.op;
    o.input = n.input;
    o.orig = &n;
    nodes.push_back(o);
  }
  auto add_node = [&](const std::string &name, const std::string &op) -> size_t {
    OutNode o;
    o.name = name;
    o.op = op;
    nodes.push_back(o);
    return nodes.size() - 1;
  };
  auto add_const = [&](const std::string &name, const std::vector<int32_t> &v, bool scalar) {
    const size_t k = add_node(name, "Const");
    nodes[k].attrs.push_back({"dtype", av_type(DT_INT32)});
    nodes[k].attrs.push_back({"value", av_tensor_i32(v, scalar)});
  };

  {
    const size_t ci = add_node("ConcatInputs", "Addons>ConcatInputs");
    std::vector<int64_t> types, ranks;
    for (const HostInput &h : built.host_inputs) {
      nodes[ci].input.push_back(h.tensor);
      types.push_back(h.dtype);
      ranks.push_back(h.rank);
    }
    if (stage && stage->symbols_input >= 0) {
      if (stage->symbols_input != (int)built.host_inputs.size() || built.symbols.empty())
        throw std::invalid_argument("the stage section expects the symbols vector as the last ConcatInputs input");
      nodes[ci].input.push_back("FeatureColumnProcess/symbols");
      types.push_back(DT_INT32);
      ranks.push_back(1);
    }
    nodes[ci].attrs.push_back({"T", av_list_types(types)});
    nodes[ci].attrs.push_back({"ranks", av_list_ints(ranks)});
    if (stage) nodes[ci].attrs.push_back({"_fcp_plan", av_s(plan_path)});
  }
  const size_t fuse = add_node("FeatureColumnProcess", "Addons>FeatureColumnProcess");
  nodes[fuse].input = {"ConcatInputs", "ConcatInputs:1", "ConcatInputs:2"};
  {
    std::vector<int64_t> types, ranks;
    for (const HostInput &d : built.device_inputs) {
      const std::string suffix = "/fcp_read";
      if (d.tensor.size() > suffix.size() && d.tensor.compare(d.tensor.size() - suffix.size(), suffix.size(), suffix) == 0 && !view.node(d.tensor)) {
        const size_t rd = add_node(d.tensor, "ReadVariableOp"); // a resource variable nobody reads as a tensor yet
        nodes[rd].input.push_back(d.tensor.substr(0, d.tensor.size() - suffix.size()));
        nodes[rd].attrs.push_back({"dtype", av_type(d.dtype)});
      }
      nodes[fuse].input.push_back(d.tensor);
      types.push_back(d.dtype);
      ranks.push_back(d.rank);
    }
    nodes[fuse].attrs.push_back({"dlpath", av_s(plan_path)});
    nodes[fuse].attrs.push_back({"input_types", av_list_types(types)});
    nodes[fuse].attrs.push_back({"input_ranks", av_list_ints(ranks)});
  }
  std::map<int, int> out_index; // plan column -> index among FeatureColumnProcess outputs
  for (size_t k = 0; k < built.spec.columns.size(); ++k)
    if (built.spec.columns[k].form != FCP_FORM_EXTERNAL) {
      const int idx = (int)out_index.size();
      out_index[(int)k] = idx;
    }
  nodes[fuse].attrs.push_back({"output_types", av_list_types(std::vector<int64_t>(out_index.size(), DT_FLOAT))}); // every column output is [prefix, dim]
  nodes[fuse].attrs.push_back({"output_ranks", av_list_ints(std::vector<int64_t>(out_index.size(), 2))});

  if (!built.symbols.empty()) {
    add_const("FeatureColumnProcess/symbols/flat_shape", {-1}, false);
    add_const("FeatureColumnProcess/symbols/axis", {0}, true);
    const size_t pack = add_node("FeatureColumnProcess/symbols", "Pack");
    nodes[pack].attrs.push_back({"N", av_i((int64_t)built.symbols.size())});
    nodes[pack].attrs.push_back({"T", av_type(DT_INT32)});
    nodes[pack].attrs.push_back({"axis", av_i(0)});
    for (size_t k = 0; k < built.symbols.size(); ++k) {
      const SymbolDef &sym = built.symbols[k];
      const auto [src, port] = split_tensor(sym.tensor);
      const int dtype = view.out_dtype(*view.node(src), port);
      const std::string base = "FeatureColumnProcess/symbols/s" + std::to_string(k);
      const size_t flat = add_node(base + "/flat", "Reshape");
      nodes[flat].input = {sym.tensor, "FeatureColumnProcess/symbols/flat_shape"};
      nodes[flat].attrs.push_back({"T", av_type(dtype)});
      nodes[flat].attrs.push_back({"Tshape", av_type(DT_INT32)});
      if (sym.last_stride > 0) {
        // rows of a plain SparseSegmentSum / Mean = max(sorted segment ids, -1) + 1 (0 rows without ids)
        std::string last = base + "/flat";
        if (sym.last_stride > 1) { // column 0 of an [nnz, k] index matrix: flat[::k]
          add_const(base + "/zero", {0}, false);
          add_const(base + "/stride", {sym.last_stride}, false);
          const size_t col = add_node(base + "/col", "StridedSlice");
          nodes[col].input = {base + "/flat", base + "/zero", base + "/zero", base + "/stride"};
          nodes[col].attrs.push_back({"T", av_type(dtype)});
          nodes[col].attrs.push_back({"Index", av_type(DT_INT32)});
          nodes[col].attrs.push_back({"begin_mask", av_i(1)});
          nodes[col].attrs.push_back({"end_mask", av_i(1)});
          nodes[col].attrs.push_back({"ellipsis_mask", av_i(0)});
          nodes[col].attrs.push_back({"new_axis_mask", av_i(0)});
          nodes[col].attrs.push_back({"shrink_axis_mask", av_i(0)});
          last = base + "/col";
        }
        const size_t c32 = add_node(base + "/cast", "Cast");
        nodes[c32].input = {last};
        nodes[c32].attrs.push_back({"SrcT", av_type(dtype)});
        nodes[c32].attrs.push_back({"DstT", av_type(DT_INT32)});
        add_const(base + "/none", {-1}, false);
        const size_t cat = add_node(base + "/cat", "ConcatV2");
        nodes[cat].input = {base + "/cast", base + "/none", "FeatureColumnProcess/symbols/axis"};
        nodes[cat].attrs.push_back({"N", av_i(2)});
        nodes[cat].attrs.push_back({"T", av_type(DT_INT32)});
        nodes[cat].attrs.push_back({"Tidx", av_type(DT_INT32)});
        const size_t mx = add_node(base + "/max", "Max");
        nodes[mx].input = {base + "/cat", "FeatureColumnProcess/symbols/axis"};
        nodes[mx].attrs.push_back({"T", av_type(DT_INT32)});
        nodes[mx].attrs.push_back({"Tidx", av_type(DT_INT32)});
        nodes[mx].attrs.push_back({"keep_dims", av_b(false)});
        add_const(base + "/one", {1}, true);
        const size_t add = add_node(base, "AddV2");
        nodes[add].input = {base + "/max", base + "/one"};
        nodes[add].attrs.push_back({"T", av_type(DT_INT32)});
        nodes[pack].input.push_back(base);
        continue;
      }
      add_const(base + "/index", {sym.index}, true);
      const size_t pick = add_node(base + "/pick", "GatherV2");
      nodes[pick].input = {base + "/flat", base + "/index", "FeatureColumnProcess/symbols/axis"};
      nodes[pick].attrs.push_back({"Tparams", av_type(dtype)});
      nodes[pick].attrs.push_back({"Tindices", av_type(DT_INT32)});
      nodes[pick].attrs.push_back({"Taxis", av_type(DT_INT32)});
      nodes[pick].attrs.push_back({"batch_dims", av_i(0)});
      const size_t cast = add_node(base, "Cast");
      nodes[cast].input = {base + "/pick"};
      nodes[cast].attrs.push_back({"SrcT", av_type(dtype)});
      nodes[cast].attrs.push_back({"DstT", av_type(DT_INT32)});
      nodes[pack].input.push_back(base);
    }
    nodes[fuse].op = "Addons>FeatureColumnProcessWithSymbols";
    nodes[fuse].input.push_back("FeatureColumnProcess/symbols");
  }

  std::vector<std::string> removed;
  for (const GroupInfo &gi : built.groups) {
    size_t orig = nodes.size();
    for (size_t k = 0; k < nodes.size(); ++k)
      if (nodes[k].orig && !nodes[k].renamed && nodes[k].name == gi.concat_node) orig = k;
    if (orig == nodes.size()) throw std::invalid_argument("concat node " + gi.concat_node + " vanished");
    std::vector<int> host_pos;
    for (size_t pos = 0; pos < gi.columns.size(); ++pos)
      if (!out_index.count(gi.columns[pos])) host_pos.push_back((int)pos);
    const size_t nw = add_node(gi.concat_node, host_pos.empty() ? "Addons>ConcatOutputsNoHost" : "Addons>ConcatOutputs");
    int first = -1;
    for (int col : gi.columns)
      if (out_index.count(col)) {
        first = out_index[col];
        break;
      }
    std::vector<int64_t> dev_concat, dev_input, host_concat, dims, buffer_types;
    nodes[nw].input = {"FeatureColumnProcess", "FeatureColumnProcess:1"};
    for (size_t pos = 0; pos < gi.columns.size(); ++pos) {
      const int col = gi.columns[pos];
      if (out_index.count(col)) {
        dev_concat.push_back((int64_t)pos);
        dev_input.push_back(out_index[col]);
      } else { // the original concat input, untouched (:2597-2602)
        host_concat.push_back((int64_t)pos);
        nodes[nw].input.push_back(built.columns[col].concat_input);
      }
      dims.push_back(built.spec.columns[col].dim);
    }
    // tensor_buffers: keep blob, tables and arena alive until the concat output is consumed
    nodes[nw].input.push_back("ConcatInputs");
    buffer_types.push_back(DT_INT8);
    for (const HostInput &d : built.device_inputs) {
      nodes[nw].input.push_back(d.tensor);
      buffer_types.push_back(d.dtype);
    }
    nodes[nw].input.push_back("FeatureColumnProcess:2");
    buffer_types.push_back(DT_INT8);
    auto &a = nodes[nw].attrs;
    a.push_back({"T", av_type(gi.dtype)});
    a.push_back({"BLOCK_THREADS", av_i(kBlockThreads)});
    a.push_back({"prefix_begin", av_i(2 * first)}); // index into output_shapes (rank 2 per output)
    a.push_back({"prefix_end", av_i(2 * first + 1)});
    a.push_back({"output_dir", av_s("")});
    a.push_back({"N", av_i((int64_t)host_pos.size())});
    a.push_back({"host_concat_indices", av_list_ints(host_concat)});
    if (!dev_concat.empty()) {
      a.push_back({"device_concat_indices", av_list_ints(dev_concat)});
      a.push_back({"device_input_indices", av_list_ints(dev_input)});
    }
    a.push_back({"embedd_dims", av_list_ints(dims)});
    a.push_back({"buffer_types", av_list_types(buffer_types)});
    nodes[orig].name = gi.concat_node + "_removed";
    nodes[orig].renamed = true;
    removed.push_back(nodes[orig].name);
  }

  std::vector<bool> dead(nodes.size(), false);
  if (prune) { // drop the removed concats and every node that only fed them
    std::map<std::string, size_t> by_name;
    for (size_t k = 0; k < nodes.size(); ++k) by_name[nodes[k].name] = k;
    std::vector<int> uses(nodes.size(), 0);
    for (const OutNode &n : nodes)
      for (const std::string &t : n.input) {
        auto it = by_name.find(split_tensor(t).first);
        if (it != by_name.end()) ++uses[it->second];
      }
    std::vector<std::string> stack = removed;
    while (!stack.empty()) {
      const std::string name = stack.back();
      stack.pop_back();
      auto it = by_name.find(name);
      if (it == by_name.end()) continue;
      const size_t k = it->second;
      if (dead[k] || uses[k] != 0) continue;
      if (nodes[k].op == "Placeholder" || nodes[k].op == "VariableV2" || nodes[k].op == "VarHandleOp") continue; // graph interface stays
      dead[k] = true;
      for (const std::string &t : nodes[k].input) {
        const std::string src = split_tensor(t).first;
        auto jt = by_name.find(src);
        if (jt != by_name.end()) --uses[jt->second];
        stack.push_back(src);
      }
    }
  }
  // GraphDef: the nodes, then every other top-level field as it was (versions, function library, ...)
  Message out;
  for (size_t k = 0; k < nodes.size(); ++k)
    if (!dead[k]) add_bytes(out, 1, serialize_node(nodes[k]));
  for (const Field &f : graph.f)
    if (f.num != 1) out.f.push_back(f);
  return serialize(out);
}

char *dup_string(const std::string &s) {
  char *p = static_cast<char *>(malloc(s.size() + 1));
  if (p) memcpy(p, s.c_str(), s.size() + 1);
  return p;
}

} // namespace

extern "C" {

int fcp_graph_build(const void *graphdef, size_t n_bytes, uint32_t flags, const char *plan_path, void **rewritten, size_t *rewritten_bytes,
                    char **description) {
  if ((!graphdef && n_bytes) || !plan_path) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "null argument");
  if (flags & ~(uint32_t)(FCP_GRAPH_HOST_CONCAT_EXTERNAL | FCP_GRAPH_STAGED | FCP_GRAPH_NO_PRUNE)) return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "unknown fcp_graph_build flags");
  if (rewritten) *rewritten = nullptr;
  if (rewritten_bytes) *rewritten_bytes = 0;
  if (description) *description = nullptr;
  try {
    const Message graph = parse_message(std::string(static_cast<const char *>(graphdef), n_bytes));
    std::vector<Node> nodes;
    for (const Field &f : graph.f)
      if (f.num == 1 && f.wt == 2) nodes.push_back(parse_node(f.bytes));
    const GraphView view(nodes);
    PlanBuilder builder(view, (flags & FCP_GRAPH_HOST_CONCAT_EXTERNAL) != 0);
    const BuiltPlan built = builder.build();
    std::string text, desc = describe(built);
    std::optional<StageInfo> stage;
    if (flags & FCP_GRAPH_STAGED) {
      auto [spec, st] = staged_for_concat_inputs(built.spec);
      text = plan_file_text(spec, &st);
      static const char *what[] = {"copied", "int64 -> int32", "row ids -> row offsets"};
      int counts[3] = {0, 0, 0};
      for (int m : st.modes) ++counts[m];
      std::string line;
      for (int k = 0; k < 3; ++k)
        if (counts[k]) line += (line.empty() ? "" : ", ") + std::to_string(counts[k]) + "x " + what[k];
      desc += "\n  staged ConcatInputs: " + line;
      stage = st;
    } else {
      text = plan_file_text(built.spec, nullptr);
    }
    FILE *f = fopen(plan_path, "w");
    if (!f || fwrite(text.data(), 1, text.size(), f) != text.size() || fclose(f) != 0) {
      if (f) fclose(f);
      return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, std::string("cannot write column plan ") + plan_path);
    }
    if (rewritten) {
      const std::string out = rewrite_graph(graph, view, built, plan_path, !(flags & FCP_GRAPH_NO_PRUNE), stage ? &*stage : nullptr);
      void *buf = malloc(out.size() ? out.size() : 1);
      if (!buf) return fcp_internal_fail(FCP_ERR_ALLOC, "out of host memory");
      memcpy(buf, out.data(), out.size());
      *rewritten = buf;
      if (rewritten_bytes) *rewritten_bytes = out.size();
    }
    if (description) *description = dup_string(desc);
    return FCP_OK;
  } catch (const Unsupported &why) {
    return fcp_internal_fail(FCP_ERR_UNSUPPORTED, std::string("nothing to fuse: ") + why.what());
  } catch (const ParseError &why) {
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, std::string("not a GraphDef: ") + why.what());
  } catch (const std::invalid_argument &why) {
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, why.what());
  } catch (const std::bad_alloc &) {
    return fcp_internal_fail(FCP_ERR_ALLOC, "out of host memory");
  } catch (const std::exception &why) { // nothing may cross the C boundary (std::length_error, out_of_range, ... of a crafted graph)
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, std::string("malformed GraphDef: ") + why.what());
  } catch (...) {
    return fcp_internal_fail(FCP_ERR_INVALID_ARGUMENT, "malformed GraphDef");
  }
}

void fcp_graph_free(void *p) { free(p); }

} // extern "C"
