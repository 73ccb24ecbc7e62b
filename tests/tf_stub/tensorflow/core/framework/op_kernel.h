// TEST INFRASTRUCTURE -- a stand-in for the few TensorFlow declarations warprnnt_op.cc uses, so that the op source meets a
// compiler and its Compute() methods can be EXECUTED in an image that has no TensorFlow (tests/test_tensorflow_stub.py,
// tests/tf_stub/run_op.cpp).  Signatures follow tensorflow/core/framework/{op_kernel.h, tensor.h, tensor_shape.h, op.h,
// shape_inference.h} of TF 2.x closely enough that code written against them compiles against the real headers too; nothing
// here is part of the product, and nothing here is TensorFlow code.
//   TF_STUB_DEVICE_MEMORY: tensors of the "GPU" device live in hipMalloc'ed memory (the harness is then compiled by hipcc).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#ifdef TF_STUB_DEVICE_MEMORY
#include <hip/hip_runtime.h>
#endif

namespace Eigen {
struct GpuDevice {
    void* stream_ = nullptr;
#ifdef TF_STUB_DEVICE_MEMORY
    hipStream_t stream() const { return static_cast<hipStream_t>(stream_); }
#else
    void* stream() const { return stream_; }
#endif
};
struct ThreadPoolDevice {};
}  // namespace Eigen

namespace tensorflow {
using int32 = std::int32_t;
using int64 = long long;
using uint8 = std::uint8_t;
enum DataType { DT_FLOAT = 1, DT_INT32 = 3, DT_UINT8 = 4 };
constexpr const char* DEVICE_CPU = "CPU";
constexpr const char* DEVICE_GPU = "GPU";

class Status {
public:
    Status() = default;
    Status(int code, std::string msg) : code_(code), msg_(std::move(msg)) {}
    bool ok() const { return code_ == 0; }
    const std::string& message() const { return msg_; }
    std::string ToString() const { return ok() ? "OK" : msg_; }
    static Status OK() { return Status(); }
private:
    int code_ = 0;
    std::string msg_;
};
namespace errors {
template <typename... A> Status make_(int code, const A&... a) { std::ostringstream os; (void)std::initializer_list<int>{(os << a, 0)...}; return Status(code, os.str()); }
template <typename... A> Status InvalidArgument(const A&... a) { return make_(3, a...); }
template <typename... A> Status Internal(const A&... a) { return make_(13, a...); }
}  // namespace errors

class TensorShape {
public:
    TensorShape() = default;
    TensorShape(std::initializer_list<int64> d) : d_(d) {}
    explicit TensorShape(std::vector<int64> d) : d_(std::move(d)) {}
    int dims() const { return static_cast<int>(d_.size()); }
    int64 dim_size(int i) const { return d_[i]; }
    int64 num_elements() const { int64 n = 1; for (int64 v : d_) n *= v; return n; }
    bool operator==(const TensorShape& o) const { return d_ == o.d_; }
private:
    std::vector<int64> d_;
};
struct TensorShapeUtils {
    static bool IsVector(const TensorShape& s) { return s.dims() == 1; }
    static bool IsMatrix(const TensorShape& s) { return s.dims() == 2; }
};

template <typename T> struct FlatMap {      // what Tensor::flat<T>() hands back: only data() and size() are used
    T* p; int64 n;
    T* data() const { return p; }
    int64 size() const { return n; }
    void setZero() const { std::memset(p, 0, sizeof(T) * static_cast<size_t>(n)); }
};

class Tensor {
public:
    Tensor() = default;
    Tensor(DataType dt, const TensorShape& s, bool on_device) : dt_(dt), shape_(s), device_(on_device) {
        const size_t bytes = static_cast<size_t>(s.num_elements()) * (dt == DT_UINT8 ? 1 : 4);
        void* p = nullptr;
#ifdef TF_STUB_DEVICE_MEMORY
        if (on_device) {
            if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) std::abort();
            buf_ = std::shared_ptr<void>(p, [](void* q) { (void)hipFree(q); });
            return;
        }
#endif
        p = std::calloc(bytes ? bytes : 1, 1);
        buf_ = std::shared_ptr<void>(p, std::free);
    }
    int dims() const { return shape_.dims(); }
    int64 dim_size(int i) const { return shape_.dim_size(i); }
    const TensorShape& shape() const { return shape_; }
    int64 NumElements() const { return shape_.num_elements(); }
    DataType dtype() const { return dt_; }
    bool on_device() const { return device_; }
    void* raw() const { return buf_.get(); }
    template <typename T> FlatMap<T> flat() { return {static_cast<T*>(buf_.get()), NumElements()}; }
    template <typename T> FlatMap<const T> flat() const { return {static_cast<const T*>(buf_.get()), NumElements()}; }
private:
    DataType dt_ = DT_FLOAT;
    TensorShape shape_;
    bool device_ = false;
    std::shared_ptr<void> buf_;
};

struct CpuWorkerThreads { int num_threads = 2; };
class DeviceBase {
public:
    const CpuWorkerThreads* tensorflow_cpu_worker_threads() const { return &workers_; }
private:
    CpuWorkerThreads workers_;
};

class OpKernelConstruction {
public:
    std::map<std::string, int> int_attrs;
    Status status;
    template <typename T> Status GetAttr(const std::string& name, T* v) const {
        auto it = int_attrs.find(name);
        if (it == int_attrs.end()) return errors::InvalidArgument("no attr ", name);
        *v = static_cast<T>(it->second);
        return Status();
    }
    void SetStatus(const Status& s) { status = s; }
    void CtxFailure(const Status& s) { status = s; }
    void CtxFailure(const char*, int, const Status& s) { status = s; }
    void CtxFailureWithWarning(const char*, int, const Status& s) { status = s; }
};

class OpKernelContext {
public:
    std::vector<Tensor> inputs;
    std::vector<std::string> input_names;
    std::vector<std::unique_ptr<Tensor>> outputs;
    std::vector<bool> output_on_host;            // HostMemory("...") of the kernel registration
    bool gpu = false;
    Eigen::GpuDevice gpu_device;
    DeviceBase dev;
    Status status_;
    const Tensor& input(int i) const { return inputs[i]; }
    Status input(const std::string& name, const Tensor** t) const {
        for (size_t i = 0; i < input_names.size(); ++i)
            if (input_names[i] == name) { *t = &inputs[i]; return Status(); }
        return errors::InvalidArgument("no input ", name);
    }
    Status allocate_output(int i, const TensorShape& s, Tensor** t) {
        if (static_cast<int>(outputs.size()) <= i) outputs.resize(i + 1);
        const bool host = static_cast<int>(output_on_host.size()) > i && output_on_host[i];
        outputs[i].reset(new Tensor(DT_FLOAT, s, gpu && !host));
        *t = outputs[i].get();
        return Status();
    }
    Status allocate_temp(DataType dt, const TensorShape& s, Tensor* t) { *t = Tensor(dt, s, gpu); return Status(); }
    template <typename D> const D& eigen_device() const;
    DeviceBase* device() { return &dev; }
    const Status& status() const { return status_; }
    void SetStatus(const Status& s) { status_ = s; }
    void CtxFailure(const Status& s) { status_ = s; }
    void CtxFailure(const char*, int, const Status& s) { status_ = s; }
    void CtxFailureWithWarning(const char*, int, const Status& s) { status_ = s; }
};
template <> inline const Eigen::GpuDevice& OpKernelContext::eigen_device<Eigen::GpuDevice>() const { return gpu_device; }

class OpKernel {
public:
    explicit OpKernel(OpKernelConstruction*) {}
    virtual ~OpKernel() = default;
    virtual void Compute(OpKernelContext* ctx) = 0;
};

#define OP_REQUIRES(CTX, EXP, STATUS) do { if (!(EXP)) { (CTX)->CtxFailure(__FILE__, __LINE__, (STATUS)); return; } } while (0)
#define OP_REQUIRES_OK(CTX, ...) do { ::tensorflow::Status s_(__VA_ARGS__); if (!s_.ok()) { (CTX)->CtxFailureWithWarning(__FILE__, __LINE__, s_); return; } } while (0)
#define TF_RETURN_IF_ERROR(...) do { ::tensorflow::Status s_ = (__VA_ARGS__); if (!s_.ok()) return s_; } while (0)

// ---- kernel registry
struct KernelDef {
    std::string op, device;
    std::vector<std::string> host_memory;
    std::function<OpKernel*(OpKernelConstruction*)> make;
};
inline std::vector<KernelDef>& kernel_registry() { static std::vector<KernelDef> r; return r; }
class Name {
public:
    explicit Name(const char* op) { def_.op = op; }
    Name& Device(const char* d) { def_.device = d; return *this; }
    Name& HostMemory(const char* arg) { def_.host_memory.push_back(arg); return *this; }
    template <typename T> Name& TypeConstraint(const char*) { return *this; }
    KernelDef def_;
};
struct KernelRegistrar {
    KernelRegistrar(const Name& n, std::function<OpKernel*(OpKernelConstruction*)> make) {
        KernelDef d = n.def_;
        d.make = std::move(make);
        kernel_registry().push_back(d);
    }
};
#define TF_STUB_CAT2(a, b) a##b
#define TF_STUB_CAT(a, b) TF_STUB_CAT2(a, b)
#define REGISTER_KERNEL_BUILDER(NAME, ...) \
    static ::tensorflow::KernelRegistrar TF_STUB_CAT(tf_stub_kernel_, __COUNTER__)((::tensorflow::NAME), \
        [](::tensorflow::OpKernelConstruction* c) -> ::tensorflow::OpKernel* { return new __VA_ARGS__(c); })
}  // namespace tensorflow
