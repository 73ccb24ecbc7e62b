// rnnt_gpu.hip -- host driver of the gfx950 path and the exported C-ABI.
//
// Replaces, behaviour for behaviour, the reference's
//   src/rnnt_entrypoint.cpp:14-185      (exports, validation, dispatch, workspace size)
//   include/detail/gpu_rnnt.h:82-253    (GpuRNNT::compute_cost_and_score / cost_and_grad /
//                                        score_forward: workspace carve, launches, D2H of costs)
// The library never allocates memory: everything lives in the caller's workspace
// (reference README.md:36-37).  The only host<->device traffic is the N-element copy of the
// costs to the caller's HOST array followed by one stream synchronisation, which the
// reference contract requires (gpu_rnnt.h:208-213).  (One opt-in exception, off by default and
// host memory only: the pinned staging buffer of rnnt_host_staging(), see stage_acquire below.)
#define RNNT_GPU_INSTANTIATE_F32 1
#include "rnnt_gpu_impl.h"

#include <link.h>
#include <limits.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>


namespace rnnt {

Profile g_prof;
std::mutex g_prof_mu;
Ranges g_ranges;
// state shared by the instantiations of run_gpu (rnnt_gpu_impl.h)
std::mutex g_stage_mu;
std::vector<HostStage*> g_stage_all;
std::atomic<int> g_stage_mode{-1};
std::atomic<long long> g_stage_bytes{0};
thread_local AuxStream t_aux;

template rnntStatus_t run_gpu<F32>(const float*, float*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                   const rnntOptions&, int, int, float, const long long*, long long);

}  // namespace rnnt

using namespace rnnt;

// The library is compiled with -fvisibility=hidden: only the C entry points of include/rnnt.h are exported
// (the reference exports exactly its five: nm -D of its libwarprnnt.so).
#pragma GCC visibility push(default)
extern "C" {

int get_warprnnt_version() { return 1; }
int get_warprnnt_extension_version(void) { return 5; }

const char* rnntGetStatusString(rnntStatus_t status) {
    // Same strings as the reference (src/rnnt_entrypoint.cpp:18-35) so log scrapers keep working.
    // A caller (a C program, ctypes) can hand over ANY int; loading a value outside the enumerators' range through the enum
    // type is undefined behaviour in C++ (found by the UBSan run of `make asan`: tests/test_sanitizers.py), so the bits are
    // read as the int they are.
    int code = 0;
    static_assert(sizeof(code) == sizeof(status), "rnntStatus_t is int-sized");
    std::memcpy(&code, &status, sizeof(code));
    switch (code) {
        case RNNT_STATUS_SUCCESS: return "no error";
        case RNNT_STATUS_MEMOPS_FAILED: return "cuda memcpy or memset failed";
        case RNNT_STATUS_INVALID_VALUE: return "invalid value";
        case RNNT_STATUS_EXECUTION_FAILED: return "execution failed";
        case RNNT_STATUS_UNKNOWN_ERROR:
        default: return "unknown error";
    }
}

rnntStatus_t get_workspace_size(int maxT, int maxU, int minibatch, bool gpu, size_t* size_bytes,
                                size_t dtype_size) {
    if (minibatch <= 0 || maxT <= 0 || maxU <= 0 || size_bytes == nullptr) return RNNT_STATUS_INVALID_VALUE;
    const size_t lat = dtype_size >= 8 ? 8 : 4;
    if (gpu)
        *size_bytes = make_layout(maxT, maxU, minibatch, lat, false).total;
    else
        *size_bytes = cpu_workspace_bytes(maxT, maxU, minibatch, lat);
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t get_workspace_size_add(int maxT, int maxU, int minibatch, size_t* size_bytes) {
    if (minibatch <= 0 || maxT <= 0 || maxU <= 0 || size_bytes == nullptr) return RNNT_STATUS_INVALID_VALUE;
    *size_bytes = make_layout(maxT, maxU, minibatch, 4, true).total;
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t compute_rnnt_loss(const float* const activations, float* gradients, const int* const flat_labels,
                               const int* const label_lengths, const int* const input_lengths,
                               int alphabet_size, int minibatch, float* costs, void* workspace,
                               rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options))
        return RNNT_STATUS_INVALID_VALUE;
    if (loc_of(options) == RNNT_CPU)
        return cpu_rnnt_f32(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, workspace, options);
    if (loc_of(options) == RNNT_GPU)
        return run_gpu<F32>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, nullptr, nullptr, workspace, options);
    return RNNT_STATUS_INVALID_VALUE;
}

rnntStatus_t compute_rnnt_loss_fp64(const double* const activations, double* gradients,
                                    const int* const flat_labels, const int* const label_lengths,
                                    const int* const input_lengths, int alphabet_size, int minibatch,
                                    double* costs, void* workspace, rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options))
        return RNNT_STATUS_INVALID_VALUE;
    if (loc_of(options) == RNNT_CPU)
        return cpu_rnnt_f64(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, workspace, options);
    if (loc_of(options) == RNNT_GPU)
        return run_gpu<F64>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, nullptr, nullptr, workspace, options);
    return RNNT_STATUS_INVALID_VALUE;
}

rnntStatus_t compute_rnnt_loss_bf16(const uint16_t* const activations, uint16_t* gradients,
                                    const int* const flat_labels, const int* const label_lengths,
                                    const int* const input_lengths, int alphabet_size, int minibatch,
                                    float* costs, void* workspace, rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu<BF16>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                         minibatch, costs, nullptr, nullptr, workspace, options);
}

rnntStatus_t compute_rnnt_loss_fp16(const uint16_t* const activations, uint16_t* gradients,
                                    const int* const flat_labels, const int* const label_lengths,
                                    const int* const input_lengths, int alphabet_size, int minibatch,
                                    float* costs, void* workspace, rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu<F16>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                        minibatch, costs, nullptr, nullptr, workspace, options);
}

// Dispatch of the enqueue-only forms on the dtype code (0 fp32, 1 fp64, 2 bf16, 3 fp16).
static rnntStatus_t run_async(const void* acts, void* grads, const int* labels, const int* label_lengths,
                              const int* input_lengths, int A, int N, void* costs_device, const void* scale,
                              void* workspace, const rnntOptions& o, int dtype_code, int phases, int want_grad,
                              float fastemit = 0.0f, const long long* offsets = nullptr, long long packed_rows = 0) {
    switch (dtype_code) {
        case 0:
            return run_gpu<F32>(static_cast<const float*>(acts), static_cast<float*>(grads), labels, label_lengths,
                                input_lengths, A, N, nullptr, static_cast<float*>(costs_device),
                                static_cast<const float*>(scale), workspace, o, phases, want_grad, fastemit, offsets, packed_rows);
        case 1:
            return run_gpu<F64>(static_cast<const double*>(acts), static_cast<double*>(grads), labels, label_lengths,
                                input_lengths, A, N, nullptr, static_cast<double*>(costs_device),
                                static_cast<const double*>(scale), workspace, o, phases, want_grad, fastemit, offsets, packed_rows);
        case 2:
            return run_gpu<BF16>(static_cast<const uint16_t*>(acts), static_cast<uint16_t*>(grads), labels,
                                 label_lengths, input_lengths, A, N, nullptr, static_cast<float*>(costs_device),
                                 static_cast<const float*>(scale), workspace, o, phases, want_grad, fastemit, offsets, packed_rows);
        case 3:
            return run_gpu<F16>(static_cast<const uint16_t*>(acts), static_cast<uint16_t*>(grads), labels,
                                label_lengths, input_lengths, A, N, nullptr, static_cast<float*>(costs_device),
                                static_cast<const float*>(scale), workspace, o, phases, want_grad, fastemit, offsets, packed_rows);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t compute_rnnt_loss_async(const void* activations, void* gradients, const int* const flat_labels,
                                     const int* const label_lengths, const int* const input_lengths,
                                     int alphabet_size, int minibatch, void* costs_device,
                                     const void* grad_scale_device, void* workspace, rnntOptions options,
                                     int dtype_code) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, grad_scale_device, workspace, options, dtype_code, 3, -1);
}

// RCCL is looked up at run time: the library links against no collective library.  Which copy matters: an ncclComm_t is
// only meaningful to the RCCL that created it, and a process can hold more than one (PyTorch ships its own librccl.so
// under torch/lib next to the ROCm one).  Order: (1) the function the caller registered with rnnt_set_rccl_all_reduce();
// (2) the ONE librccl that is already mapped into the process (dl_iterate_phdr; opened with RTLD_NOLOAD, so nothing new is
// loaded) -- two different mapped copies and no registered function is an error, reported on stderr, never a guess;
// (3) nothing mapped: librccl.so.1 / librccl.so by name (a caller that creates its communicator later, from the same name).
struct Rccl {
    using AllReduce = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
    std::mutex mu;
    AllReduce registered = nullptr;                 // rnnt_set_rccl_all_reduce
    bool tried = false;
    AllReduce found = nullptr;
    char from[512] = {0};                           // where `found` came from (rnnt_rccl_source)
    // communicators introduced through rnnt_sharded_prepare, each with the ncclAllReduce resolved for it THEN: a sharded
    // step never has to look RCCL up, so it has no way to fail in front of its collective
    std::vector<std::pair<void*, AllReduce>> prepared;
};
static Rccl& rccl_state() { static Rccl r; return r; }

static int rccl_visit(struct dl_phdr_info* info, size_t, void* data) {
    auto* paths = static_cast<std::vector<std::string>*>(data);
    if (info->dlpi_name == nullptr) return 0;
    const char* base = strrchr(info->dlpi_name, '/');
    base = base ? base + 1 : info->dlpi_name;
    if (strncmp(base, "librccl.so", 10) != 0) return 0;
    char real[PATH_MAX];
    std::string path = realpath(info->dlpi_name, real) != nullptr ? real : info->dlpi_name;   // (symlinks of one file count once)
    for (const auto& q : *paths)
        if (q == path) return 0;
    paths->push_back(path);
    return 0;
}

static Rccl::AllReduce rccl_all_reduce() {
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    if (r.registered != nullptr) return r.registered;
    if (r.tried) return r.found;
    r.tried = true;
    std::vector<std::string> mapped;
    dl_iterate_phdr(rccl_visit, &mapped);
    if (mapped.size() > 1) {
        fprintf(stderr, "warprnnt: %zu different RCCL libraries are mapped into this process (", mapped.size());
        for (size_t i = 0; i < mapped.size(); ++i) fprintf(stderr, "%s%s", i ? ", " : "", mapped[i].c_str());
        fprintf(stderr, "): compute_rnnt_loss_sharded cannot tell which one made the communicator -- pass its ncclAllReduce "
                        "to rnnt_set_rccl_all_reduce()\n");
        return nullptr;
    }
    auto take = [&](const char* name, int flags) {
        void* h = dlopen(name, flags);
        if (h == nullptr) return false;
        r.found = reinterpret_cast<Rccl::AllReduce>(dlsym(h, "ncclAllReduce"));
        if (r.found != nullptr) snprintf(r.from, sizeof(r.from), "%s", name);
        return r.found != nullptr;
    };
    if (mapped.size() == 1) { (void)take(mapped[0].c_str(), RTLD_NOW | RTLD_NOLOAD); return r.found; }
    for (const char* name : {"librccl.so.1", "librccl.so"})
        if (take(name, RTLD_NOW | RTLD_GLOBAL)) break;
    return r.found;
}

void rnnt_set_rccl_all_reduce(void* nccl_all_reduce_fn) {
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    r.registered = reinterpret_cast<Rccl::AllReduce>(nccl_all_reduce_fn);
    if (nccl_all_reduce_fn == nullptr) { r.tried = false; r.found = nullptr; r.from[0] = 0; }   // NULL: forget, look again at the next call
}

rnntStatus_t rnnt_sharded_prepare(void* rccl_comm) {
    if (rccl_comm == nullptr) return RNNT_STATUS_INVALID_VALUE;
    const Rccl::AllReduce fn = rccl_all_reduce();                 // (takes the lock itself)
    if (fn == nullptr) return RNNT_STATUS_EXECUTION_FAILED;       // no (unambiguous) RCCL in this process: said so on stderr
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    for (auto& e : r.prepared)
        if (e.first == rccl_comm) { e.second = fn; return RNNT_STATUS_SUCCESS; }
    r.prepared.emplace_back(rccl_comm, fn);
    return RNNT_STATUS_SUCCESS;
}

void rnnt_sharded_release(void* rccl_comm) {
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    for (size_t i = 0; i < r.prepared.size(); ++i)
        if (r.prepared[i].first == rccl_comm) { r.prepared.erase(r.prepared.begin() + static_cast<long>(i)); return; }
}

static Rccl::AllReduce prepared_all_reduce(void* rccl_comm) {
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    for (const auto& e : r.prepared)
        if (e.first == rccl_comm) return e.second;
    return nullptr;
}

const char* rnnt_rccl_source(void) {
    if (rccl_all_reduce() == nullptr) return "";
    Rccl& r = rccl_state();
    std::lock_guard<std::mutex> g(r.mu);
    return r.registered != nullptr ? "registered by the caller" : r.from;
}

rnntStatus_t compute_rnnt_loss_sharded(const void* activations, void* gradients, const int* const flat_labels,
                                       const int* const label_lengths, const int* const input_lengths,
                                       int alphabet_size, int minibatch, void* costs_device,
                                       const void* grad_scale_device, double* loss_sum_count_device, void* rccl_comm,
                                       void* workspace, rnntOptions options, int dtype_code) {
    // (argument errors every rank of a job makes alike return before anything is enqueued)
    if (loss_sum_count_device == nullptr || loc_of(options) != RNNT_GPU) return RNNT_STATUS_INVALID_VALUE;
    // A communicator must have been introduced with rnnt_sharded_prepare(): RCCL was resolved THERE, where a failure is an
    // ordinary error on that rank before any collective exists.  An unprepared communicator is a programming error every
    // rank of a job makes alike (same code), like the two above: INVALID_VALUE, nothing enqueued.  From here on nothing
    // can stop this rank short of the collective.
    Rccl::AllReduce all_reduce = nullptr;
    if (rccl_comm != nullptr && (all_reduce = prepared_all_reduce(rccl_comm)) == nullptr) return RNNT_STATUS_INVALID_VALUE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(options.stream);
    rnntStatus_t st = bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                               alphabet_size, minibatch, options)
                          ? RNNT_STATUS_INVALID_VALUE
                          : run_async(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                                      minibatch, costs_device, grad_scale_device, workspace, options, dtype_code, 3, -1);
    (void)hipGetLastError();
    if (st == RNNT_STATUS_SUCCESS) {
        if (dtype_code == 1)
            hipLaunchKernelGGL((loss_sum_kernel<double>), dim3(1), dim3(256), 0, stream, static_cast<const double*>(costs_device),
                               minibatch, loss_sum_count_device);
        else
            hipLaunchKernelGGL((loss_sum_kernel<float>), dim3(1), dim3(256), 0, stream, static_cast<const float*>(costs_device),
                               minibatch, loss_sum_count_device);
        if (hipGetLastError() != hipSuccess) st = RNNT_STATUS_EXECUTION_FAILED;
    }
    // ALL RANKS OR NONE: a rank whose local part failed (a shard shape only this rank has, a launch error) still joins
    // the collective -- with a NaN pair, so every rank's reduced loss is NaN and every rank can see the step failed --
    // instead of leaving its peers blocked in ncclAllReduce for ever; it then returns its own status.
    if (st != RNNT_STATUS_SUCCESS && rccl_comm != nullptr &&
        hipMemsetAsync(loss_sum_count_device, 0xff, 2 * sizeof(double), stream) != hipSuccess) {
        (void)hipGetLastError();
        return st;                                       // (cannot even mark the pair: nothing sane is left to send)
    }
    if (rccl_comm != nullptr &&
        all_reduce(loss_sum_count_device, loss_sum_count_device, 2, /*ncclFloat64*/ 8, /*ncclSum*/ 0, rccl_comm, stream) != 0)
        return st != RNNT_STATUS_SUCCESS ? st : RNNT_STATUS_EXECUTION_FAILED;
    return st;
}

rnntStatus_t compute_rnnt_loss_fwd(const void* activations, const int* const flat_labels,
                                   const int* const label_lengths, const int* const input_lengths,
                                   int alphabet_size, int minibatch, void* costs_device, void* workspace,
                                   rnntOptions options, int dtype_code, int prepare_backward) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, nullptr, workspace, options, dtype_code, 1, prepare_backward ? 1 : 0);
}

rnntStatus_t compute_rnnt_loss_bwd(const void* activations, void* gradients, const void* grad_scale_device,
                                   int alphabet_size, int minibatch, void* workspace, rnntOptions options,
                                   int dtype_code) {
    if (activations == nullptr || gradients == nullptr || workspace == nullptr || alphabet_size <= 0 ||
        minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, nullptr, nullptr, nullptr, alphabet_size, minibatch, nullptr,
                     grad_scale_device, workspace, options, dtype_code, 2, 1);
}

rnntStatus_t compute_rnnt_loss_likelihoods(const void* workspace, int minibatch, rnntOptions options, int dtype_code,
                                           double* ll_forward_host, double* ll_backward_host) {
    if (workspace == nullptr || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU ||
        ll_forward_host == nullptr || ll_backward_host == nullptr || dtype_code < 0 || dtype_code > 3)
        return RNNT_STATUS_INVALID_VALUE;
    const Layout lay = make_layout(options.maxT, options.maxU, minibatch, dtype_code == 1 ? 8 : 4, false);
    const char* ws = reinterpret_cast<const char*>(align_up(reinterpret_cast<size_t>(workspace)));
    hipStream_t stream = reinterpret_cast<hipStream_t>(options.stream);
    const size_t bytes = sizeof(double) * static_cast<size_t>(minibatch);
    if (hipMemcpyAsync(ll_forward_host, ws + lay.llf, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipMemcpyAsync(ll_backward_host, ws + lay.llb, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess)
        return RNNT_STATUS_MEMOPS_FAILED;
    if (hipStreamSynchronize(stream) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
    for (int b = 0; b < minibatch; ++b) ll_forward_host[b] *= kLn2;       // the lattice keeps the forward one in base 2
    return RNNT_STATUS_SUCCESS;
}

rnntStatus_t compute_rnnt_loss_lattice_dump(const void* workspace, const int* const label_lengths, const int* const input_lengths,
                                            int minibatch, int sample, rnntOptions options, int dtype_code, double* alpha_device,
                                            double* beta_device) {
    if (workspace == nullptr || label_lengths == nullptr || input_lengths == nullptr || alpha_device == nullptr || beta_device == nullptr ||
        minibatch <= 0 || sample < 0 || sample >= minibatch || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU ||
        dtype_code < 0 || dtype_code > 3)
        return RNNT_STATUS_INVALID_VALUE;
    const int A = options.blank_label >= 0 ? options.blank_label + 1 : 1;           // (the plan only checks the blank against it)
    const unsigned cells = static_cast<unsigned>(options.maxT) * static_cast<unsigned>(options.maxU);
    const dim3 grid((cells + 255) / 256);
    if (dtype_code == 1) {
        Plan<double> p;
        if (!make_plan(p, A, minibatch, options, const_cast<void*>(workspace), nullptr, label_lengths, input_lengths, static_cast<double*>(nullptr)))
            return RNNT_STATUS_INVALID_VALUE;
        hipLaunchKernelGGL((lattice_dump_kernel<double>), grid, dim3(256), 0, p.stream, p.alpha, p.beta, p.offa, p.offb, input_lengths,
                           label_lengths, sample, p.maxT, p.maxU, p.Up, p.lat_w, p.lat_sh, alpha_device, beta_device, p.padflag + 2);
    } else {
        Plan<float> p;
        if (!make_plan(p, A, minibatch, options, const_cast<void*>(workspace), nullptr, label_lengths, input_lengths, static_cast<float*>(nullptr)))
            return RNNT_STATUS_INVALID_VALUE;
        hipLaunchKernelGGL((lattice_dump_kernel<float>), grid, dim3(256), 0, p.stream, p.alpha, p.beta, p.offa, p.offb, input_lengths,
                           label_lengths, sample, p.maxT, p.maxU, p.Up, p.lat_w, p.lat_sh, alpha_device, beta_device, p.padflag + 2);
    }
    return hipGetLastError() == hipSuccess ? RNNT_STATUS_SUCCESS : RNNT_STATUS_EXECUTION_FAILED;
}

rnntStatus_t compute_rnnt_loss_fastemit(const void* activations, void* gradients, const int* const flat_labels,
                                        const int* const label_lengths, const int* const input_lengths,
                                        int alphabet_size, int minibatch, void* costs_device,
                                        const void* grad_scale_device, void* workspace, rnntOptions options,
                                        int dtype_code, float fastemit_lambda) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, grad_scale_device, workspace, options, dtype_code, 3, -1, fastemit_lambda);
}

rnntStatus_t compute_rnnt_loss_fwd_fastemit(const void* activations, const int* const flat_labels,
                                            const int* const label_lengths, const int* const input_lengths,
                                            int alphabet_size, int minibatch, void* costs_device, void* workspace,
                                            rnntOptions options, int dtype_code, int prepare_backward,
                                            float fastemit_lambda) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, nullptr, workspace, options, dtype_code, 1, prepare_backward != 0 ? 1 : 0,
                     fastemit_lambda);
}

rnntStatus_t compute_rnnt_loss_packed(const void* activations, void* gradients, const int* const flat_labels,
                                      const int* const label_lengths, const int* const input_lengths,
                                      const long long* const row_offsets, long long total_rows, int alphabet_size,
                                      int minibatch, void* costs_device, const void* grad_scale_device,
                                      void* workspace, rnntOptions options, int dtype_code, float fastemit_lambda) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || row_offsets == nullptr)
        return RNNT_STATUS_INVALID_VALUE;
    if (loc_of(options) == RNNT_CPU) {
        // the reference's CPU contract on the packed rows: every array on the host (row_offsets and costs too),
        // log-probabilities in, sparse log-prob gradients out (rnnt_cpu.cpp); fp32 / fp64, no scale, no FastEmit
        if (dtype_code > 1 || dtype_code < 0 || grad_scale_device != nullptr || fastemit_lambda != 0.0f ||
            total_rows <= 0)
            return RNNT_STATUS_INVALID_VALUE;
        return cpu_rnnt_packed(activations, gradients, flat_labels, label_lengths, input_lengths, row_offsets,
                               alphabet_size, minibatch, costs_device, workspace, options, dtype_code == 1);
    }
    if (loc_of(options) != RNNT_GPU) return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, grad_scale_device, workspace, options, dtype_code, 3, -1, fastemit_lambda,
                     row_offsets, total_rows);
}

rnntStatus_t compute_rnnt_loss_packed_fwd(const void* activations, const int* const flat_labels,
                                          const int* const label_lengths, const int* const input_lengths,
                                          const long long* const row_offsets, long long total_rows,
                                          int alphabet_size, int minibatch, void* costs_device, void* workspace,
                                          rnntOptions options, int dtype_code, int prepare_backward,
                                          float fastemit_lambda) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || row_offsets == nullptr || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, nullptr, workspace, options, dtype_code, 1, prepare_backward != 0 ? 1 : 0,
                     fastemit_lambda, row_offsets, total_rows);
}

rnntStatus_t compute_rnnt_loss_packed_bwd(const void* activations, void* gradients, const void* grad_scale_device,
                                          const long long* const row_offsets, long long total_rows,
                                          int alphabet_size, int minibatch, void* workspace, rnntOptions options,
                                          int dtype_code) {
    if (activations == nullptr || gradients == nullptr || row_offsets == nullptr || workspace == nullptr ||
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || loc_of(options) != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, nullptr, nullptr, nullptr, alphabet_size, minibatch, nullptr,
                     grad_scale_device, workspace, options, dtype_code, 2, 1, 0.0f, row_offsets, total_rows);
}

void rnnt_set_aux_stream(CUstream stream) {
    t_aux.stream = reinterpret_cast<hipStream_t>(stream);
    aux_drop_events();          // the next call makes them again on ITS device (a stream of another GPU, NULL = release)
}

int rnnt_host_staging(int mode) {
    const int before = stage_enabled() ? 1 : 0;
    if (mode == 0 || mode == 1) g_stage_mode.store(mode, std::memory_order_relaxed);
    return before;
}

long long rnnt_host_staging_bytes(void) { return g_stage_bytes.load(std::memory_order_relaxed); }

long long rnnt_host_staging_release(void) {
    std::lock_guard<std::mutex> g(g_stage_mu);
    const long long before = g_stage_bytes.load(std::memory_order_relaxed);
    for (HostStage* st : g_stage_all)
        if (!st->busy.load(std::memory_order_relaxed)) stage_free(st);
    return before - g_stage_bytes.load(std::memory_order_relaxed);
}

void rnnt_profile_enable(int on) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    g_prof.on = (on & 1) != 0;            // bit 0: stage timers (HIP events)
    g_ranges.mode = (on & 2) ? 1 : 0;     // bit 1: roctx ranges around the stages
}

void rnnt_profile_collect(void) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    if (g_prof.on && g_prof.ready && g_prof.pending) prof_accumulate();
}

void rnnt_profile_reset(void) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    for (double& m : g_prof.ms) m = 0.0;
    g_prof.calls = 0;
}

int rnnt_profile_read(double* ms, int n) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    for (int i = 0; i < n && i < 5; ++i) ms[i] = g_prof.ms[i];
    return g_prof.calls;
}

}  // extern "C"
#pragma GCC visibility pop
