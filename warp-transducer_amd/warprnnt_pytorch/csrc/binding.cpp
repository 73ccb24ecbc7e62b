// binding.cpp -- `warprnnt_pytorch._warp_rnnt_ext`: the compiled PyTorch extension module over libwarprnnt.so.
//
// The reference ships its binding as a pybind11 module (pytorch_binding/src/binding.cpp:12-162: cpu_rnnt / gpu_rnnt with
// eight arguments, dims read off `acts`, a temporary workspace from the framework's allocator, the current stream, a dtype
// switch that prints to stderr and returns -1).  This is that module for the MI355X library: the same two entry points with
// the same signatures and behaviour, the extension entry points the Python wrappers use (two-phase forward / backward), and
// -- what the ctypes loader cannot give -- the whole loss as a C++ autograd function (`rnnt_loss`): argument checks,
// allocations, both library calls and the reduction run without returning to Python, and the backward node is executed by
// the autograd engine without the GIL.  Host code only (no kernels here): compiled with g++ against torch and linked to
// libwarprnnt.so; the hot path stays behind the C-ABI of include/rnnt.h.
#include <torch/extension.h>

#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPCachingAllocatorMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <map>
#include <tuple>

#include "rnnt.h"

namespace {

// (PyTorch on ROCm keeps the device type "cuda": the guards and the allocator hooks are the ...MasqueradingAsCUDA forms)
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
using OptionalDeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

// ---------------------------------------------------------------------------------------------------------- helpers
int dtype_code(const at::Tensor& t) {              // include/rnnt.h: 0 fp32, 1 fp64, 2 bf16, 3 fp16
    switch (t.scalar_type()) {
        case at::kFloat: return 0;
        case at::kDouble: return 1;
        case at::kBFloat16: return 2;
        case at::kHalf: return 3;
        default: return -1;
    }
}

size_t elem_size_for_workspace(int code) { return code == 1 ? 8 : (code == 0 ? 4 : 2); }

// The GPU location dereferences labels and both length vectors ON THE DEVICE of the activations: a host tensor (or one of
// another GPU) there is a GPU fault, not an exception -- unless it is caught here.
void check_on_device_of(const at::Tensor& acts, const at::Tensor& t, const char* name) {
    TORCH_CHECK_VALUE(t.is_cuda() && t.device() == acts.device(), name, " must be on the device of the activations (", acts.device(),
                      "), got ", t.device());
}
void check_gpu_arguments(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& lengths, const at::Tensor& label_lengths) {
    if (labels.numel() > 0) check_on_device_of(acts, labels, "labels");
    check_on_device_of(acts, lengths, "lengths");
    check_on_device_of(acts, label_lengths, "label_lengths");
}
// A caller-owned workspace: on the device of the activations and at least as large as get_workspace_size says TODAY (the
// layout is private and has grown between versions: a size cached from an older library must not be silently overrun).
void check_workspace(const at::Tensor& acts, const at::Tensor& ws, size_t need) {
    TORCH_CHECK_VALUE(ws.is_cuda() && ws.device() == acts.device(), "workspace must be on the device of the activations");
    TORCH_CHECK_VALUE(ws.is_contiguous() && ws.nbytes() >= need, "workspace of ", ws.nbytes(), " bytes, get_workspace_size asks for ", need);
}

void check_status(rnntStatus_t st, const char* what) {
    if (st != RNNT_STATUS_SUCCESS)
        throw std::runtime_error(std::string(what) + " failed: " + rnntGetStatusString(st) + " (status " + std::to_string(static_cast<int>(st)) + ")");
}

// get_workspace_size per (T, U, N, element size): a handful of shapes per process, asked once each
size_t workspace_bytes(int T, int U, int N, bool gpu, size_t esz) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, size_t> cache;
    const auto key = std::make_tuple(T, U, N, (esz == 8 ? 2 : 0) | (gpu ? 1 : 0));
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    size_t n = 0;
    check_status(get_workspace_size(T, U, N, gpu, &n, esz), "get_workspace_size");
    std::lock_guard<std::mutex> g(mu);
    cache[key] = n;
    return n;
}

rnntOptions make_options(rnntComputeLocation loc, const at::Tensor& acts, int blank, int num_threads, hipStream_t stream) {
    rnntOptions o{};                                // zero-initialised, as the reference asks (include/rnnt.h:43-64)
    o.loc = loc;
    o.num_threads = num_threads > 0 ? static_cast<unsigned>(num_threads) : 0u;
    o.stream = reinterpret_cast<CUstream>(stream);
    o.blank_label = blank;
    o.maxT = static_cast<int>(acts.size(1));
    o.maxU = static_cast<int>(acts.size(2));
    o.batch_first = true;
    return o;
}

const int* iptr(const at::Tensor& t) { return t.numel() > 0 ? t.data_ptr<int>() : nullptr; }

// ---------------------------------------------------------------------------------------------------------- the reference's module
// int cpu_rnnt(acts, labels, input_lengths, label_lengths, costs, grads, blank_label, num_threads)
// (pytorch_binding/src/binding.cpp:12-82): acts are LOG-PROBS on the host, grads the sparse d/d(log-probs); 0 / -1.
int cpu_rnnt(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& input_lengths, const at::Tensor& label_lengths,
             at::Tensor costs, at::Tensor grads, int blank_label, int num_threads) {
    const int N = static_cast<int>(acts.size(0)), T = static_cast<int>(acts.size(1)), U = static_cast<int>(acts.size(2)),
              A = static_cast<int>(acts.size(3));
    const int code = dtype_code(acts);
    if (code != 0 && code != 1) {
        std::fprintf(stderr, "warp_rnnt.cpu_rnnt: unsupported data type %s\n", c10::toString(acts.scalar_type()));
        return -1;
    }
    const size_t esz = code == 1 ? 8 : 4;
    at::Tensor ws = at::empty({static_cast<long>(workspace_bytes(T, U, N, false, esz))}, at::TensorOptions().dtype(at::kByte));
    const rnntOptions opt = make_options(RNNT_CPU, acts, blank_label, num_threads, nullptr);
    void* g = grads.numel() > 0 ? grads.data_ptr() : nullptr;
    rnntStatus_t st;
    if (code == 0)
        st = compute_rnnt_loss(acts.data_ptr<float>(), static_cast<float*>(g), iptr(labels), iptr(label_lengths), iptr(input_lengths), A, N,
                               costs.data_ptr<float>(), ws.data_ptr(), opt);
    else
        st = compute_rnnt_loss_fp64(acts.data_ptr<double>(), static_cast<double*>(g), iptr(labels), iptr(label_lengths), iptr(input_lengths),
                                    A, N, costs.data_ptr<double>(), ws.data_ptr(), opt);
    check_status(st, "compute_rnnt_loss (RNNT_CPU)");
    return 0;
}

// int gpu_rnnt(...) (binding.cpp:84-154): acts are raw LOGITS on the device, labels / lengths device tensors, costs a HOST
// tensor; current stream (binding.cpp:104), device of `acts` (:118), temporary workspace from the caching allocator (:120,128).
int gpu_rnnt(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& input_lengths, const at::Tensor& label_lengths,
             at::Tensor costs, at::Tensor grads, int blank_label, int num_threads, c10::optional<at::Tensor> workspace) {
    TORCH_CHECK_VALUE(acts.is_cuda(), "gpu_rnnt needs device tensors");
    const int N = static_cast<int>(acts.size(0)), T = static_cast<int>(acts.size(1)), U = static_cast<int>(acts.size(2)),
              A = static_cast<int>(acts.size(3));
    const int code = dtype_code(acts);
    if (code < 0) {
        std::fprintf(stderr, "warp_rnnt.gpu_rnnt: unsupported data type %s\n", c10::toString(acts.scalar_type()));
        return -1;
    }
    const DeviceGuard guard(acts.device());
    const hipStream_t stream = c10::hip::getCurrentHIPStream(acts.device().index()).stream();
    check_gpu_arguments(acts, labels, input_lengths, label_lengths);
    const size_t need = workspace_bytes(T, U, N, true, elem_size_for_workspace(code));
    if (workspace.has_value()) check_workspace(acts, *workspace, need);
    at::Tensor ws = workspace.has_value() ? *workspace : at::empty({static_cast<long>(need)}, acts.options().dtype(at::kByte));
    const rnntOptions opt = make_options(RNNT_GPU, acts, blank_label, num_threads, stream);
    void* g = grads.numel() > 0 ? grads.data_ptr() : nullptr;
    rnntStatus_t st;
    switch (code) {
        case 0: st = compute_rnnt_loss(static_cast<const float*>(acts.data_ptr()), static_cast<float*>(g), iptr(labels), iptr(label_lengths),
                                       iptr(input_lengths), A, N, costs.data_ptr<float>(), ws.data_ptr(), opt); break;
        case 1: st = compute_rnnt_loss_fp64(static_cast<const double*>(acts.data_ptr()), static_cast<double*>(g), iptr(labels), iptr(label_lengths),
                                            iptr(input_lengths), A, N, costs.data_ptr<double>(), ws.data_ptr(), opt); break;
        case 2: st = compute_rnnt_loss_bf16(static_cast<const uint16_t*>(acts.data_ptr()), static_cast<uint16_t*>(g), iptr(labels), iptr(label_lengths),
                                            iptr(input_lengths), A, N, costs.data_ptr<float>(), ws.data_ptr(), opt); break;
        default: st = compute_rnnt_loss_fp16(static_cast<const uint16_t*>(acts.data_ptr()), static_cast<uint16_t*>(g), iptr(labels), iptr(label_lengths),
                                             iptr(input_lengths), A, N, costs.data_ptr<float>(), ws.data_ptr(), opt); break;
    }
    check_status(st, "compute_rnnt_loss (RNNT_GPU)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------- two-phase entries
// Forward phase (compute_rnnt_loss_fwd[_fastemit]): enqueue only; returns the workspace, allocated from the caching
// allocator on the current stream (the one the kernels are enqueued on).
at::Tensor gpu_rnnt_fwd(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& input_lengths, const at::Tensor& label_lengths,
                        at::Tensor costs_device, int blank_label, bool prepare_backward, double fastemit_lambda) {
    const int N = static_cast<int>(acts.size(0)), T = static_cast<int>(acts.size(1)), U = static_cast<int>(acts.size(2)),
              A = static_cast<int>(acts.size(3));
    const int code = dtype_code(acts);
    TORCH_CHECK_TYPE(code >= 0, "rnnt_loss: unsupported dtype ", c10::toString(acts.scalar_type()), " for the GPU location");
    check_gpu_arguments(acts, labels, input_lengths, label_lengths);
    const int index = acts.device().index();
    OptionalDeviceGuard guard;                  // (entered only when the tensors do not live on the current device)
    if (at::hip::current_device() != index) guard.set_index(index);
    const hipStream_t stream = c10::hip::getCurrentHIPStream(index).stream();
    at::Tensor ws = at::empty({static_cast<long>(workspace_bytes(T, U, N, true, elem_size_for_workspace(code)))}, acts.options().dtype(at::kByte));
    const rnntOptions opt = make_options(RNNT_GPU, acts, blank_label, 0, stream);
    const int* lab = labels.numel() > 0 ? labels.data_ptr<int>() : reinterpret_cast<const int*>(costs_device.data_ptr());   // maxU == 1: never read
    rnntStatus_t st;
    if (fastemit_lambda != 0.0)
        st = compute_rnnt_loss_fwd_fastemit(acts.data_ptr(), lab, iptr(label_lengths), iptr(input_lengths), A, N, costs_device.data_ptr(),
                                            ws.data_ptr(), opt, code, prepare_backward ? 1 : 0, static_cast<float>(fastemit_lambda));
    else
        st = compute_rnnt_loss_fwd(acts.data_ptr(), lab, iptr(label_lengths), iptr(input_lengths), A, N, costs_device.data_ptr(), ws.data_ptr(),
                                   opt, code, prepare_backward ? 1 : 0);
    check_status(st, "compute_rnnt_loss_fwd");
    return ws;
}

// Gradient phase (compute_rnnt_loss_bwd) from the workspace of gpu_rnnt_fwd.  `fwd_stream`: the raw handle of the stream the
// forward ran on (0 = unknown): a backward on another stream tells the allocator that the workspace is in use there.
void gpu_rnnt_bwd(const at::Tensor& acts, at::Tensor grads, const c10::optional<at::Tensor>& grad_scale, const at::Tensor& workspace,
                  int blank_label, long fwd_stream) {
    const int N = static_cast<int>(acts.size(0)), A = static_cast<int>(acts.size(3));
    const int code = dtype_code(acts);
    check_workspace(acts, workspace, workspace_bytes(static_cast<int>(acts.size(1)), static_cast<int>(acts.size(2)), N, true, elem_size_for_workspace(code)));
    const int index = acts.device().index();
    OptionalDeviceGuard guard;
    if (at::hip::current_device() != index) guard.set_index(index);
    const c10::hip::HIPStream cur = c10::hip::getCurrentHIPStream(index);
    if (fwd_stream != 0 && reinterpret_cast<long>(cur.stream()) != fwd_stream)
        c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(workspace.storage().data_ptr(),
                                                                                      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(index));
    const rnntOptions opt = make_options(RNNT_GPU, acts, blank_label, 0, cur.stream());
    const void* scale = grad_scale.has_value() && grad_scale->numel() > 0 ? grad_scale->data_ptr() : nullptr;
    check_status(compute_rnnt_loss_bwd(acts.data_ptr(), grads.data_ptr(), scale, A, N, workspace.data_ptr(), opt, code), "compute_rnnt_loss_bwd");
}

// ---------------------------------------------------------------------------------------------------------- argument checks
// T == max(lengths), U == max(label_lengths) + 1 need the VALUES: the two small vectors are copied to pinned host memory
// behind whatever the stream holds, one synchronisation, maxima on the host (the reference reads them back with two
// .item() calls: two reductions and two synchronisations)
void read_max_lengths(const at::Tensor& lengths, const at::Tensor& label_lengths, int* max_t_out, int* max_l_out) {
    const long n = lengths.numel();
    int max_t = 0, max_l = 0;
    if (lengths.is_cuda() && label_lengths.is_cuda() && lengths.device() == label_lengths.device()) {
        static thread_local at::Tensor pinned;
        if (!pinned.defined() || pinned.numel() < 2 * n)
            pinned = at::empty({2 * n > 256 ? 2 * n : 256}, at::TensorOptions().dtype(at::kInt).pinned_memory(true));
        const DeviceGuard guard(lengths.device());
        const hipStream_t stream = c10::hip::getCurrentHIPStream(lengths.device().index()).stream();
        int* h = pinned.data_ptr<int>();
        C10_HIP_CHECK(hipMemcpyAsync(h, lengths.data_ptr<int>(), sizeof(int) * n, hipMemcpyDeviceToHost, stream));
        C10_HIP_CHECK(hipMemcpyAsync(h + n, label_lengths.data_ptr<int>(), sizeof(int) * n, hipMemcpyDeviceToHost, stream));
        C10_HIP_CHECK(hipStreamSynchronize(stream));
        for (long i = 0; i < n; ++i) { max_t = h[i] > max_t || i == 0 ? h[i] : max_t; max_l = h[n + i] > max_l || i == 0 ? h[n + i] : max_l; }
    } else {
        max_t = lengths.max().item<int>();
        max_l = label_lengths.max().item<int>();
    }
    *max_t_out = max_t;
    *max_l_out = max_l;
}

// The reference binding's checks, in its order, with its exception types and message texts (incl. the spelling "lenghts"):
// pytorch_binding/warprnnt_pytorch/__init__.py:103-140 (the Python twin of this function is warprnnt_pytorch/_checks.py).
void certify_inputs(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& lengths, const at::Tensor& label_lengths,
                    bool read_lengths) {
    TORCH_CHECK_TYPE(labels.scalar_type() == at::kInt, "labels must be torch.int32");
    TORCH_CHECK_TYPE(label_lengths.scalar_type() == at::kInt, "label_lengths must be torch.int32");
    TORCH_CHECK_TYPE(lengths.scalar_type() == at::kInt, "lengths must be torch.int32");
    TORCH_CHECK_VALUE(acts.is_contiguous(), "log_probs must be contiguous");
    TORCH_CHECK_VALUE(labels.is_contiguous(), "labels must be contiguous");
    TORCH_CHECK_VALUE(label_lengths.is_contiguous(), "label_lengths must be contiguous");
    TORCH_CHECK_VALUE(lengths.is_contiguous(), "lengths must be contiguous");
    TORCH_CHECK_VALUE(acts.dim() >= 1 && lengths.dim() >= 1 && lengths.size(0) == acts.size(0), "must have a length per example.");
    TORCH_CHECK_VALUE(label_lengths.dim() >= 1 && label_lengths.size(0) == acts.size(0), "must have a label length per example.");
    TORCH_CHECK_VALUE(acts.dim() == 4, "log_probs must be 4D");
    TORCH_CHECK_VALUE(labels.dim() == 2, "labels must be 2D");
    TORCH_CHECK_VALUE(lengths.dim() == 1, "lenghts must be 1D");
    TORCH_CHECK_VALUE(label_lengths.dim() == 1, "label_lenghts must be 1D");
    if (!read_lengths) return;
    int max_t = 0, max_l = 0;
    read_max_lengths(lengths, label_lengths, &max_t, &max_l);
    TORCH_CHECK_VALUE(acts.size(1) == max_t, "Input length mismatch");
    TORCH_CHECK_VALUE(acts.size(2) == max_l + 1, "Output length mismatch");
}

// ---------------------------------------------------------------------------------------------------------- the loss as a C++ autograd function
// GPU tensors, two-phase route (what warprnnt_pytorch._RNNT does in Python): forward = statistics + lattice (+ the
// gradient-coefficient table when a gradient is wanted) and ONE reduction kernel; backward = one elementwise kernel for the
// per-sample factor (grad_output, 1/N of 'mean') and the gradient kernel.  Between the two only the workspace is kept --
// no gradient tensor (the reference keeps one and rescales it twice: __init__.py:24,36-50).
struct RNNTFunction : public torch::autograd::Function<RNNTFunction> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& acts, const at::Tensor& labels,
                              const at::Tensor& act_lens, const at::Tensor& label_lens, int64_t blank, int64_t reduction,
                              double fastemit_lambda, bool validate) {
        certify_inputs(acts, labels, act_lens, label_lens, validate);
        const long n = acts.size(0);
        const bool need_grad = acts.requires_grad();
        const at::ScalarType cost_dtype = acts.scalar_type() == at::kDouble ? at::kDouble : at::kFloat;
        at::Tensor costs = at::empty({n}, acts.options().dtype(cost_dtype));
        at::Tensor ws = gpu_rnnt_fwd(acts, labels, act_lens, label_lens, costs, static_cast<int>(blank), need_grad, fastemit_lambda);
        if (need_grad) {
            ctx->save_for_backward({acts, ws});
            ctx->saved_data["blank"] = blank;
            ctx->saved_data["mean_scale"] = reduction == 2 ? 1.0 / static_cast<double>(n) : 1.0;
            ctx->saved_data["stream"] = static_cast<int64_t>(reinterpret_cast<long>(c10::hip::getCurrentHIPStream(acts.device().index()).stream()));
        }
        if (reduction == 1) return costs.sum(0, /*keepdim=*/true);      // reference __init__.py:36-40: sum -> shape (1,)
        if (reduction == 2) return costs.mean(0, /*keepdim=*/true);     // 'mean' = sum / N
        return costs;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor& acts = saved[0];
        const at::Tensor& ws = saved[1];
        const long n = acts.size(0);
        const at::ScalarType sdt = acts.scalar_type() == at::kDouble ? at::kDouble : at::kFloat;
        at::Tensor g = grad_outputs[0].reshape({-1});
        if (g.scalar_type() != sdt || g.device() != acts.device()) g = g.to(acts.device(), sdt);
        // the per-sample factor of the gradient kernel in ONE elementwise kernel (a (1,) grad_output of 'sum' / 'mean' is
        // broadcast; the 1/N of 'mean' folded in)
        const at::Tensor scale = g.expand({n}) * ctx->saved_data["mean_scale"].toDouble();
        at::Tensor grads = at::empty_like(acts);
        gpu_rnnt_bwd(acts, grads, scale, ws, static_cast<int>(ctx->saved_data["blank"].toInt()), ctx->saved_data["stream"].toInt());
        return {grads, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// reduction: 0 'none', 1 'sum', 2 'mean'
at::Tensor rnnt_loss(const at::Tensor& acts, const at::Tensor& labels, const at::Tensor& act_lens, const at::Tensor& label_lens, int64_t blank,
                     int64_t reduction, double fastemit_lambda, bool validate) {
    TORCH_CHECK_VALUE(acts.is_cuda(), "the compiled rnnt_loss serves GPU tensors (the CPU location goes through cpu_rnnt)");
    return RNNTFunction::apply(acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate);
}

// ---------------------------------------------------------------------------------------------------------- additive joint
// RNNTLossAdd as a C++ autograd function (the Python twin: warprnnt_pytorch/add_network.py): the loss of the joint
// trans_acts[:, :, None] + pred_acts[:, None] without forming it (compute_rnnt_loss_add_fwd_dt / _bwd_dt; the reference's
// add_network call shape, pytorch_binding/test/test_time.py:51-70).  Checks in the order and with the texts of add_network._certify.
size_t workspace_bytes_add(int T, int U, int N) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, size_t> cache;
    const auto key = std::make_tuple(T, U, N);
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    size_t n = 0;
    check_status(get_workspace_size_add(T, U, N, &n), "get_workspace_size_add");
    std::lock_guard<std::mutex> g(mu);
    cache[key] = n;
    return n;
}

void certify_add(const at::Tensor& f, const at::Tensor& g, const at::Tensor& labels, const at::Tensor& lengths, const at::Tensor& label_lengths,
                 bool read_lengths) {
    TORCH_CHECK_TYPE(labels.scalar_type() == at::kInt, "labels must be torch.int32");
    TORCH_CHECK_TYPE(label_lengths.scalar_type() == at::kInt, "label_lengths must be torch.int32");
    TORCH_CHECK_TYPE(lengths.scalar_type() == at::kInt, "lengths must be torch.int32");
    TORCH_CHECK_VALUE(f.is_contiguous(), "trans_acts must be contiguous");
    TORCH_CHECK_VALUE(g.is_contiguous(), "pred_acts must be contiguous");
    TORCH_CHECK_VALUE(labels.is_contiguous(), "labels must be contiguous");
    TORCH_CHECK_VALUE(lengths.is_contiguous(), "lengths must be contiguous");
    TORCH_CHECK_VALUE(label_lengths.is_contiguous(), "label_lengths must be contiguous");
    TORCH_CHECK_VALUE(f.dim() == 3, "trans_acts must be 3D");
    TORCH_CHECK_VALUE(g.dim() == 3, "pred_acts must be 3D");
    TORCH_CHECK_VALUE(labels.dim() == 2, "labels must be 2D");
    TORCH_CHECK_VALUE(lengths.dim() == 1, "lengths must be 1D");
    TORCH_CHECK_VALUE(label_lengths.dim() == 1, "label_lengths must be 1D");
    TORCH_CHECK_VALUE(f.is_cuda() && g.is_cuda(), "the additive-joint loss runs on the GPU only");
    const int code = dtype_code(f);
    TORCH_CHECK_TYPE((code == 0 || code == 2 || code == 3) && g.scalar_type() == f.scalar_type(),
                     "trans_acts and pred_acts must both be torch.float32, torch.bfloat16 or torch.float16");
    TORCH_CHECK_VALUE(g.size(0) == f.size(0) && g.size(2) == f.size(2), "trans_acts (B,T,V) and pred_acts (B,U+1,V) disagree");
    TORCH_CHECK_VALUE(lengths.size(0) == f.size(0) && label_lengths.size(0) == f.size(0), "must have a length per example.");
    if (!read_lengths) return;
    int max_t = 0, max_l = 0;
    read_max_lengths(lengths, label_lengths, &max_t, &max_l);
    TORCH_CHECK_VALUE(f.size(1) == max_t, "Input length mismatch");
    TORCH_CHECK_VALUE(g.size(1) == max_l + 1, "Output length mismatch");
}

rnntOptions make_options_add(const at::Tensor& f, const at::Tensor& g, int blank, hipStream_t stream) {
    rnntOptions o{};
    o.loc = RNNT_GPU;
    o.stream = reinterpret_cast<CUstream>(stream);
    o.blank_label = blank;
    o.maxT = static_cast<int>(f.size(1));
    o.maxU = static_cast<int>(g.size(1));
    o.batch_first = true;
    return o;
}

struct RNNTAddFunction : public torch::autograd::Function<RNNTAddFunction> {
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& f, const at::Tensor& g, const at::Tensor& labels,
                              const at::Tensor& act_lens, const at::Tensor& label_lens, int64_t blank, int64_t reduction,
                              double fastemit_lambda, bool validate) {
        certify_add(f, g, labels, act_lens, label_lens, validate);
        check_on_device_of(f, g, "pred_acts");
        check_gpu_arguments(f, labels, act_lens, label_lens);
        const long n = f.size(0);
        const int T = static_cast<int>(f.size(1)), U = static_cast<int>(g.size(1)), V = static_cast<int>(f.size(2));
        const bool need_grad = f.requires_grad() || g.requires_grad();
        const int index = f.device().index();
        OptionalDeviceGuard guard;
        if (at::hip::current_device() != index) guard.set_index(index);
        const hipStream_t stream = c10::hip::getCurrentHIPStream(index).stream();
        at::Tensor costs = at::empty({n}, f.options().dtype(at::kFloat));
        at::Tensor ws = at::empty({static_cast<long>(workspace_bytes_add(T, U, static_cast<int>(n)))}, f.options().dtype(at::kByte));
        const rnntOptions opt = make_options_add(f, g, static_cast<int>(blank), stream);
        const int* lab = labels.numel() > 0 ? labels.data_ptr<int>() : reinterpret_cast<const int*>(costs.data_ptr());   // maxU == 1: never read
        check_status(compute_rnnt_loss_add_fwd_dt(f.data_ptr(), g.data_ptr(), lab, iptr(label_lens), iptr(act_lens), V, static_cast<int>(n),
                                                  costs.data_ptr<float>(), ws.data_ptr(), opt, dtype_code(f), need_grad ? 1 : 0,
                                                  static_cast<float>(fastemit_lambda)),
                     "compute_rnnt_loss_add_fwd");
        if (need_grad) {
            ctx->save_for_backward({f, g, labels, act_lens, label_lens, ws});
            ctx->saved_data["blank"] = blank;
            ctx->saved_data["mean_scale"] = reduction == 2 ? 1.0 / static_cast<double>(n) : 1.0;
            ctx->saved_data["stream"] = static_cast<int64_t>(reinterpret_cast<long>(stream));
        }
        if (reduction == 1) return costs.sum(0, /*keepdim=*/true);
        if (reduction == 2) return costs.mean(0, /*keepdim=*/true);
        return costs;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_outputs) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &f = saved[0], &g = saved[1], &labels = saved[2], &act_lens = saved[3], &label_lens = saved[4], &ws = saved[5];
        const long n = f.size(0);
        const int index = f.device().index();
        OptionalDeviceGuard guard;
        if (at::hip::current_device() != index) guard.set_index(index);
        const c10::hip::HIPStream cur = c10::hip::getCurrentHIPStream(index);
        const long fwd_stream = ctx->saved_data["stream"].toInt();
        if (fwd_stream != 0 && reinterpret_cast<long>(cur.stream()) != fwd_stream)
            c10::hip::HIPCachingAllocatorMasqueradingAsCUDA::recordStreamMasqueradingAsCUDA(ws.storage().data_ptr(),
                                                                                          c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(index));
        at::Tensor go = grad_outputs[0].reshape({-1});
        if (go.scalar_type() != at::kFloat || go.device() != f.device()) go = go.to(f.device(), at::kFloat);
        const at::Tensor scale = go.expand({n}) * ctx->saved_data["mean_scale"].toDouble();
        at::Tensor df = at::empty_like(f), dg = at::empty_like(g);
        const rnntOptions opt = make_options_add(f, g, static_cast<int>(ctx->saved_data["blank"].toInt()), cur.stream());
        const int* lab = labels.numel() > 0 ? labels.data_ptr<int>() : reinterpret_cast<const int*>(scale.data_ptr());
        check_status(compute_rnnt_loss_add_bwd_dt(f.data_ptr(), g.data_ptr(), df.data_ptr(), dg.data_ptr(), scale.data_ptr<float>(), lab,
                                                  iptr(label_lens), iptr(act_lens), static_cast<int>(f.size(2)), static_cast<int>(n),
                                                  ws.data_ptr(), opt, dtype_code(f)),
                     "compute_rnnt_loss_add_bwd");
        return {df, dg, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor rnnt_loss_add(const at::Tensor& trans_acts, const at::Tensor& pred_acts, const at::Tensor& labels, const at::Tensor& act_lens,
                         const at::Tensor& label_lens, int64_t blank, int64_t reduction, double fastemit_lambda, bool validate) {
    return RNNTAddFunction::apply(trans_acts, pred_acts, labels, act_lens, label_lens, blank, reduction, fastemit_lambda, validate);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "warp-transducer for MI355X: compiled PyTorch extension over libwarprnnt.so";
    m.def("cpu_rnnt", &cpu_rnnt, "RNNT CPU version (log-probs in, sparse log-prob gradients out)");
    m.def("gpu_rnnt", &gpu_rnnt, "RNNT GPU version (logits in, dense logit gradients out; host costs)", py::arg("acts"), py::arg("labels"),
          py::arg("input_lengths"), py::arg("label_lengths"), py::arg("costs"), py::arg("grads"), py::arg("blank_label"), py::arg("num_threads"),
          py::arg("workspace") = py::none());
    m.def("gpu_rnnt_fwd", &gpu_rnnt_fwd, "two-phase forward (compute_rnnt_loss_fwd): returns the workspace", py::arg("acts"), py::arg("labels"),
          py::arg("input_lengths"), py::arg("label_lengths"), py::arg("costs_device"), py::arg("blank_label"), py::arg("prepare_backward"),
          py::arg("fastemit_lambda") = 0.0);
    m.def("gpu_rnnt_bwd", &gpu_rnnt_bwd, "two-phase backward (compute_rnnt_loss_bwd)", py::arg("acts"), py::arg("grads"), py::arg("grad_scale"),
          py::arg("workspace"), py::arg("blank_label"), py::arg("fwd_stream") = 0);
    m.def("certify_inputs", &certify_inputs, "the reference binding's argument checks", py::arg("log_probs"), py::arg("labels"), py::arg("lengths"),
          py::arg("label_lengths"), py::arg("read_lengths") = true);
    m.def("rnnt_loss", &rnnt_loss, "RNN-T loss of GPU tensors as a C++ autograd function (reduction: 0 none, 1 sum, 2 mean)", py::arg("acts"),
          py::arg("labels"), py::arg("act_lens"), py::arg("label_lens"), py::arg("blank") = 0, py::arg("reduction") = 2,
          py::arg("fastemit_lambda") = 0.0, py::arg("validate") = true);
    m.def("rnnt_loss_add", &rnnt_loss_add, "RNN-T loss of the additive joint trans_acts[:, :, None] + pred_acts[:, None] as a C++ autograd function",
          py::arg("trans_acts"), py::arg("pred_acts"), py::arg("labels"), py::arg("act_lens"), py::arg("label_lens"), py::arg("blank") = 0,
          py::arg("reduction") = 2, py::arg("fastemit_lambda") = 0.0, py::arg("validate") = true);
    m.def("library_version", []() { return get_warprnnt_version(); });
}
