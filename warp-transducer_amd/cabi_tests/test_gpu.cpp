// C++ consumer of the C-ABI (include/rnnt.h), the way the reference's tests/test_gpu.cu uses it:
// zero-initialised by-value rnntOptions, get_workspace_size with its default dtype_size argument,
// device activations / labels / lengths, HOST costs, throw on a non-success status.
// Golden numbers: reference tests/test_gpu.cu:29-32 (small_test), :100-133 (options_test);
// inf_test / grad_check follow :203-262 and :264-441 (inputs from std::mt19937 as tests/random.cpp).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include <rnnt.h>

static void ok(rnntStatus_t s, const char* what) {
    if (s != RNNT_STATUS_SUCCESS) throw std::runtime_error(std::string(what) + ": " + rnntGetStatusString(s));
}
static void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
template <typename T> struct DeviceArray {
    T* p = nullptr;
    explicit DeviceArray(const std::vector<T>& h) {
        hip_ok(hipMalloc(&p, h.size() * sizeof(T) + 16), "hipMalloc");
        hip_ok(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "H2D");
    }
    explicit DeviceArray(size_t n) { hip_ok(hipMalloc(&p, n * sizeof(T) + 16), "hipMalloc"); }
    ~DeviceArray() { (void)hipFree(p); }
};

// One call through the C-ABI: returns per-sample costs, fills grads (if wanted).
static std::vector<float> run(const std::vector<float>& acts, const std::vector<int>& labels,
                              const std::vector<int>& label_lengths, const std::vector<int>& lengths, int T, int U,
                              int A, int blank, std::vector<float>* grads) {
    const int B = static_cast<int>(lengths.size());
    hipStream_t stream;
    hip_ok(hipStreamCreate(&stream), "hipStreamCreate");
    DeviceArray<float> d_acts(acts), d_grads(acts.size());
    DeviceArray<int> d_labels(labels.empty() ? std::vector<int>{0} : labels), d_ll(label_lengths), d_tl(lengths);
    rnntOptions options{};
    options.maxT = T;
    options.maxU = U;
    options.blank_label = blank;
    options.loc = RNNT_GPU;
    options.stream = reinterpret_cast<CUstream>(stream);
    size_t bytes = 0;
    ok(get_workspace_size(T, U, B, true, &bytes), "get_workspace_size");
    void* ws = nullptr;
    hip_ok(hipMalloc(&ws, bytes), "hipMalloc workspace");
    std::vector<float> costs(B);
    ok(compute_rnnt_loss(d_acts.p, grads ? d_grads.p : nullptr, d_labels.p, d_ll.p, d_tl.p, A, B, costs.data(), ws,
                         options),
       "compute_rnnt_loss");
    if (grads) {
        grads->resize(acts.size());
        hip_ok(hipMemcpy(grads->data(), d_grads.p, acts.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H");
    }
    (void)hipFree(ws);
    (void)hipStreamDestroy(stream);
    return costs;
}

static bool small_test() {
    const std::vector<float> acts = {0.1f, 0.6f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.6f, 0.1f, 0.1f,
                                     0.1f, 0.1f, 0.2f, 0.8f, 0.1f, 0.1f, 0.6f, 0.1f, 0.1f, 0.1f,
                                     0.1f, 0.1f, 0.2f, 0.1f, 0.1f, 0.7f, 0.1f, 0.2f, 0.1f, 0.1f};
    const auto costs = run(acts, {1, 2}, {2}, {2}, 2, 3, 5, 0, nullptr);
    return std::fabs(costs[0] - 4.495666f) < 1e-4f;
}

static bool options_test() {
    const std::vector<float> acts = {
        0.065357f, 0.787530f, 0.081592f, 0.529716f, 0.750675f, 0.754135f, 0.609764f, 0.868140f, 0.622532f,
        0.668522f, 0.858039f, 0.164539f, 0.989780f, 0.944298f, 0.603168f, 0.946783f, 0.666203f, 0.286882f,
        0.094184f, 0.366674f, 0.736168f, 0.166680f, 0.714154f, 0.399400f, 0.535982f, 0.291821f, 0.612642f,
        0.324241f, 0.800764f, 0.524106f, 0.779195f, 0.183314f, 0.113745f, 0.240222f, 0.339470f, 0.134160f,
        0.505562f, 0.051597f, 0.640290f, 0.430733f, 0.829473f, 0.177467f, 0.320700f, 0.042883f, 0.302803f,
        0.675178f, 0.569537f, 0.558474f, 0.083132f, 0.060165f, 0.107958f, 0.748615f, 0.943918f, 0.486356f,
        0.418199f, 0.652408f, 0.024243f, 0.134582f, 0.366342f, 0.295830f, 0.923670f, 0.689929f, 0.741898f,
        0.250005f, 0.603430f, 0.987289f, 0.592606f, 0.884672f, 0.543450f, 0.660770f, 0.377128f, 0.358021f};
    const std::vector<float> expected = {
        -0.186844f, -0.062555f, 0.249399f, -0.203377f, 0.202399f, 0.000977f, -0.141016f, 0.079123f, 0.061893f,
        -0.011552f, -0.081280f, 0.092832f, -0.154257f, 0.229433f, -0.075176f, -0.246593f, 0.146405f, 0.100188f,
        -0.012918f, -0.061593f, 0.074512f, -0.055986f, 0.219831f, -0.163845f, -0.497627f, 0.209240f, 0.288387f,
        0.013605f, -0.030220f, 0.016615f, 0.113925f, 0.062781f, -0.176706f, -0.667078f, 0.367659f, 0.299419f,
        -0.356344f, -0.055347f, 0.411691f, -0.096922f, 0.029459f, 0.067463f, -0.063518f, 0.027654f, 0.035863f,
        -0.154499f, -0.073942f, 0.228441f, -0.166790f, -0.000088f, 0.166878f, -0.172370f, 0.105565f, 0.066804f,
        0.023875f, -0.118256f, 0.094381f, -0.104707f, -0.108934f, 0.213642f, -0.369844f, 0.180118f, 0.189726f,
        0.025714f, -0.079462f, 0.053748f, 0.122328f, -0.238789f, 0.116460f, -0.598687f, 0.302203f, 0.296484f};
    std::vector<float> grads;
    const auto costs = run(acts, {1, 2, 1, 1}, {2, 2}, {4, 4}, 4, 3, 3, 0, &grads);
    bool good = std::fabs(costs[0] - 4.2806528590890736) < 1e-4 && std::fabs(costs[1] - 3.9384369822503591) < 1e-4;
    for (size_t i = 0; i < grads.size(); ++i) good = good && std::fabs(grads[i] - expected[i]) < 1e-4f;
    return good;
}

static std::vector<float> gen_acts(size_t n) {
    std::vector<float> v(n);
    std::mt19937 engine(0);
    std::uniform_real_distribution<> unit(0, 1);
    for (auto& x : v) x = static_cast<float>(unit(engine));
    return v;
}
static std::vector<int> gen_labels(int A, int L) {
    std::vector<int> v(L);
    std::mt19937 engine(1);
    std::uniform_int_distribution<> pick(1, A - 1);
    for (auto& x : v) x = pick(engine);
    if (L >= 3) { v[L / 2] = v[L / 2 + 1]; v[L / 2 - 1] = v[L / 2]; }
    return v;
}

static bool inf_test() {
    const int A = 15, T = 50, L = 10;
    auto labels = gen_labels(A, L - 1);
    labels[0] = 2;
    std::vector<float> grads;
    const auto costs = run(gen_acts(static_cast<size_t>(A) * T * L), labels, {L - 1}, {T}, T, L, A, 0, &grads);
    bool good = !std::isinf(costs[0]) && !std::isnan(costs[0]);
    for (float g : grads) good = good && !std::isnan(g);
    return good;
}

// central differences against the analytic gradient (reference tolerance on the GPU: 1e-2)
static bool grad_check(int A, int T, int L, int B, float tol) {
    auto acts = gen_acts(static_cast<size_t>(A) * T * L * B);
    std::vector<int> labels, ll(B, L - 1), tl(B, T);
    for (int b = 0; b < B; ++b) { auto l = gen_labels(A, L - 1); labels.insert(labels.end(), l.begin(), l.end()); }
    std::vector<float> grads;
    run(acts, labels, ll, tl, T, L, A, 0, &grads);
    std::mt19937 pick(7);
    double diff = 0, tot = 0;
    const float eps = 1e-2f;
    for (int k = 0; k < 200; ++k) {                     // a random subset (the reference perturbs every element)
        const size_t i = pick() % acts.size();
        const float old = acts[i];
        acts[i] = old + eps;
        double cp = 0; for (float c : run(acts, labels, ll, tl, T, L, A, 0, nullptr)) cp += c;
        acts[i] = old - eps;
        double cm = 0; for (float c : run(acts, labels, ll, tl, T, L, A, 0, nullptr)) cm += c;
        acts[i] = old;
        const double num = (cp - cm) / (2 * eps);
        diff += (grads[i] - num) * (grads[i] - num);
        tot += grads[i] * grads[i];
    }
    return diff / tot < tol;
}

// Extension: the packed layout through the C-ABI (compute_rnnt_loss_packed).  options_test's shape with ragged
// lengths {4,3} x {2,1}: sample b keeps only its T_b x U_b rows; costs and gradients must equal the padded
// call's on those rows.
static bool packed_test() {
    const int B = 2, T = 4, U = 3, A = 3;
    std::vector<float> acts(static_cast<size_t>(B) * T * U * A);
    {
        std::mt19937 engine(5);
        std::uniform_real_distribution<> unit(0, 1);
        for (auto& x : acts) x = static_cast<float>(unit(engine));
    }
    const std::vector<int> labels = {1, 2, 1, 1}, ll = {2, 1}, tl = {4, 3};
    std::vector<float> padded_grads;
    const auto padded_costs = run(acts, labels, ll, tl, T, U, A, 0, &padded_grads);
    std::vector<float> packed;
    std::vector<long long> offsets = {0};
    std::vector<size_t> origin;                            // padded element index of every packed element
    for (int b = 0; b < B; ++b) {
        for (int t = 0; t < tl[b]; ++t)
            for (int u = 0; u <= ll[b]; ++u)
                for (int k = 0; k < A; ++k) {
                    const size_t at = ((static_cast<size_t>(b) * T + t) * U + u) * A + k;
                    packed.push_back(acts[at]);
                    origin.push_back(at);
                }
        offsets.push_back(static_cast<long long>(packed.size() / A));
    }
    const long long rows = offsets.back();
    DeviceArray<float> d_acts(packed), d_grads(packed.size()), d_costs(static_cast<size_t>(B));
    DeviceArray<int> d_labels(labels), d_ll(ll), d_tl(tl);
    DeviceArray<long long> d_off(offsets);
    rnntOptions options{};
    options.maxT = T; options.maxU = U; options.blank_label = 0; options.loc = RNNT_GPU;   // default stream
    size_t bytes = 0;
    ok(get_workspace_size(T, U, B, true, &bytes), "get_workspace_size");
    void* ws = nullptr;
    hip_ok(hipMalloc(&ws, bytes), "hipMalloc workspace");
    ok(compute_rnnt_loss_packed(d_acts.p, d_grads.p, d_labels.p, d_ll.p, d_tl.p, d_off.p, rows, A, B, d_costs.p, nullptr,
                                ws, options, 0, 0.0f),
       "compute_rnnt_loss_packed");
    hip_ok(hipDeviceSynchronize(), "sync");
    std::vector<float> costs(B), grads(packed.size());
    hip_ok(hipMemcpy(costs.data(), d_costs.p, B * sizeof(float), hipMemcpyDeviceToHost), "D2H");
    hip_ok(hipMemcpy(grads.data(), d_grads.p, grads.size() * sizeof(float), hipMemcpyDeviceToHost), "D2H");
    bool good = true;
    for (int b = 0; b < B; ++b) good = good && std::fabs(costs[b] - padded_costs[b]) < 1e-5f * std::fabs(padded_costs[b]);
    for (size_t i = 0; i < grads.size(); ++i) good = good && std::fabs(grads[i] - padded_grads[origin[i]]) < 1e-5f;
    // misuse: more rows than the workspace was sized for, no offsets
    good = good && compute_rnnt_loss_packed(d_acts.p, d_grads.p, d_labels.p, d_ll.p, d_tl.p, d_off.p,
                                            static_cast<long long>(B) * T * U + 1, A, B, d_costs.p, nullptr, ws,
                                            options, 0, 0.0f) == RNNT_STATUS_INVALID_VALUE;
    good = good && compute_rnnt_loss_packed(d_acts.p, d_grads.p, d_labels.p, d_ll.p, d_tl.p, nullptr, rows, A, B,
                                            d_costs.p, nullptr, ws, options, 0, 0.0f) == RNNT_STATUS_INVALID_VALUE;
    (void)hipFree(ws);
    return good;
}

int main() {
    if (get_warprnnt_version() != 1) { std::fprintf(stderr, "Invalid Warp-transducer version.\n"); return 1; }
    std::printf("Running GPU tests through the C-ABI\n");
    bool status = true;
    try {
        status &= small_test();              std::printf("finish small_test %d\n", status);
        status &= options_test();            std::printf("finish options_test %d\n", status);
        status &= inf_test();                std::printf("finish inf_test %d\n", status);
        status &= grad_check(20, 50, 15, 1, 1e-2f);
        status &= grad_check(5, 10, 5, 65, 1e-2f);
        std::printf("finish grad_check %d\n", status);
        status &= packed_test();             std::printf("finish packed_test %d\n", status);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    std::printf(status ? "Tests pass\n" : "Some or all tests fail\n");
    return status ? 0 : 1;
}
