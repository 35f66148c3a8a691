// tf_mock.h — a MINIMAL mock of the TensorFlow C++ op-kernel API surface that
// recom_amd/tf_shim/fcp_tf_ops.cc uses.  TEST SCAFFOLDING for this repository's own shim only:
// TensorFlow is absent from the image, so tests/test_host.py compiles the shim against these
// declarations with `g++ -fsyntax-only` to catch typos and type errors.  Nothing here implements
// TensorFlow behaviour, and nothing of the reference is built with it.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <string>
#include <vector>

namespace tensorflow {
using int64 = long long;
using int32 = int;
using int8 = signed char;

enum DataType { DT_INVALID = 0, DT_FLOAT = 1, DT_INT32 = 3, DT_INT8 = 6, DT_INT64 = 9 };
int DataTypeSize(DataType);
template <typename T> struct DataTypeToEnum;
template <> struct DataTypeToEnum<float> { static constexpr DataType value = DT_FLOAT; };
template <> struct DataTypeToEnum<int> { static constexpr DataType value = DT_INT32; };

class Status {
public:
  static Status OK();
  bool ok() const;
};
namespace errors {
template <typename... A> Status InvalidArgument(A...);
template <typename... A> Status Internal(A...);
template <typename... A> Status Unimplemented(A...);
template <typename... A> Status ResourceExhausted(A...);
template <typename... A> Status NotFound(A...);
template <typename... A> Status Aborted(A...);
} // namespace errors

class TensorShape {
public:
  TensorShape();
  TensorShape(std::initializer_list<int64>);
  void AddDim(int64);
  int64 num_elements() const;
};

template <typename T> struct Flat {
  T *data() const;
  T &operator()(int64) const;
};

class Tensor {
public:
  Tensor();
  void *data() const;
  int64 NumElements() const;
  int dims() const;
  int64 dim_size(int) const;
  DataType dtype() const;
  template <typename T> Flat<T> flat();
  template <typename T> Flat<const T> flat() const;
  Tensor Slice(int64, int64) const;
  Status BitcastFrom(const Tensor &, DataType, const TensorShape &);
};

namespace stream_executor {
struct StreamInterface { void **GpuStreamMemberHack(); };
struct Stream { StreamInterface *implementation(); };
} // namespace stream_executor
namespace se = stream_executor;
struct DeviceContext { se::Stream *stream(); };
struct DeviceBase {
  struct GpuDeviceInfo { int gpu_id; };
  const GpuDeviceInfo *tensorflow_gpu_device_info() const;
};

struct AllocatorAttributes {
  void set_gpu_compatible(bool);
};

class OpKernelConstruction {
public:
  template <typename T> Status GetAttr(const char *, T *);
  void CtxFailure(const Status &);
  DeviceBase *device() const;
};
class OpKernelContext {
public:
  const Tensor &input(int);
  int num_inputs() const;
  Status allocate_output(int, const TensorShape &, Tensor **);
  Status allocate_output(int, const TensorShape &, Tensor **, AllocatorAttributes);
  Status allocate_temp(DataType, const TensorShape &, Tensor *);
  void set_output(int, const Tensor &);
  DeviceContext *op_device_context();
  DeviceBase *device() const;
  void CtxFailure(const Status &);
};
class OpKernel {
public:
  explicit OpKernel(OpKernelConstruction *);
  virtual ~OpKernel();
  virtual void Compute(OpKernelContext *) = 0;
};

#define OP_REQUIRES_OK(CTX, ...)           \
  do {                                     \
    ::tensorflow::Status s_(__VA_ARGS__);  \
    if (!s_.ok()) {                        \
      (CTX)->CtxFailure(s_);               \
      return;                              \
    }                                      \
  } while (0)
#define OP_REQUIRES(CTX, EXP, STATUS) \
  do {                                \
    if (!(EXP)) {                     \
      (CTX)->CtxFailure(STATUS);      \
      return;                         \
    }                                 \
  } while (0)

struct OpDefBuilderMock {
  OpDefBuilderMock &Input(const char *);
  OpDefBuilderMock &Output(const char *);
  OpDefBuilderMock &Attr(const char *);
};
struct KernelDefBuilderMock {
  KernelDefBuilderMock &Device(const char *);
  KernelDefBuilderMock &HostMemory(const char *);
  template <typename T> KernelDefBuilderMock &TypeConstraint(const char *);
};
KernelDefBuilderMock Name(const char *);
constexpr const char *DEVICE_CPU = "CPU";
constexpr const char *DEVICE_GPU = "GPU";
OpDefBuilderMock RegisterOpMock(const char *);
template <typename K> int RegisterKernelMock(const KernelDefBuilderMock &);
#define TF_MOCK_CAT2(a, b) a##b
#define TF_MOCK_CAT(a, b) TF_MOCK_CAT2(a, b)
#define REGISTER_OP(NAME) static ::tensorflow::OpDefBuilderMock TF_MOCK_CAT(op_reg_, __COUNTER__) = ::tensorflow::RegisterOpMock(NAME)
#define REGISTER_KERNEL_BUILDER(BUILDER, ...) \
  static int TF_MOCK_CAT(kernel_reg_, __COUNTER__) = ::tensorflow::RegisterKernelMock<__VA_ARGS__>(BUILDER)
} // namespace tensorflow
