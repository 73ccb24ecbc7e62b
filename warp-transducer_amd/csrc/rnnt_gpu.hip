// rnnt_gpu.hip -- host driver of the gfx950 path and the exported C-ABI.
//
// Replaces, behaviour for behaviour, the reference's
//   src/rnnt_entrypoint.cpp:14-185      (exports, validation, dispatch, workspace size)
//   include/detail/gpu_rnnt.h:82-253    (GpuRNNT::compute_cost_and_score / cost_and_grad /
//                                        score_forward: workspace carve, launches, D2H of costs)
// The library never allocates memory: everything lives in the caller's workspace
// (reference README.md:36-37).  The only host<->device traffic is the N-element copy of the
// costs to the caller's HOST array followed by one stream synchronisation, which the
// reference contract requires (gpu_rnnt.h:208-213).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rnnt.h"
#include "rnnt_cpu.h"
#include "rnnt_joint_kernels.h"
#include "rnnt_kernels.h"

namespace rnnt {

// ----------------------------------------------------------------------------- workspace
constexpr size_t kAlign = 256;
static inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Layout {
    size_t lp2, logz, alpha, rowtab, beta, offa, offb, llf, llb, costs, rowmax, side, side_bytes, wmat, total;
};

// lat = bytes of one lattice value (4: fp32 lattice for 16/32-bit activations, 8: fp64).
// joint: also the additive-joint planes (get_workspace_size_add); they sit BEHIND everything the
// materialised path uses, so a plan carved with joint = true is valid for both.
static Layout make_layout(int maxT, int maxU, int N, size_t lat, bool joint) {
    const size_t D = lat_rows(maxT, maxU);      // diagonals + padding rows
    const size_t Up = (static_cast<size_t>(maxU) + 63) / 64 * 64;   // one 64-lane row per wavefront
    const size_t W = Up / 64;
    const size_t sk = D * Up * N;               // skewed lattice cells
    Layout l{};
    size_t o = 0;
    l.lp2 = o;   o = align_up(o + sk * 2 * lat);
    l.logz = o;  o = align_up(o + sk * lat);
    l.alpha = o; o = align_up(o + sk * lat);
    l.rowtab = o; o = align_up(o + static_cast<size_t>(maxT) * maxU * N * 4 * lat);
    l.beta = o;  o = align_up(o + (sk + Up + 64) * lat);
    l.offa = o;  o = align_up(o + D * W * N * sizeof(double));
    l.offb = o;  o = align_up(o + (D * W * N + D) * sizeof(double));
    l.llf = o;   o = align_up(o + N * sizeof(double));
    l.llb = o;   o = align_up(o + N * sizeof(double));
    l.costs = o; o = align_up(o + N * sizeof(double));
    // additive joint only: row maxima of f and g, dense matrices W, CB, CL (row stride = maxU rounded up to 8)
    l.rowmax = o; l.wmat = o; l.side = o; l.side_bytes = 0;
    if (joint) {
        o = align_up(o + ((static_cast<size_t>(maxT) + maxU) * N + 1) * sizeof(float));   // + the +inf sentinel
        // correction sums for the GEMM epilogues: sfb[N*maxT] | sgb[N*maxU] | sgl[N*maxU] floats | far flags[N] ints
        l.side = o;
        l.side_bytes = ((static_cast<size_t>(maxT) + 2 * static_cast<size_t>(maxU)) * N + N) * sizeof(float);
        o = align_up(o + l.side_bytes);
        l.wmat = o;   o = align_up(o + 3 * static_cast<size_t>(maxT) * joint_upad(maxU) * N * sizeof(float));   // W | CB | CL
    }
    l.total = o + kAlign;                       // slack to align the caller's base pointer
    return l;
}

// ----------------------------------------------------------------------------- profiling
// Stage boundaries as HIP events on the caller's stream: 0 start, 1 after the statistics, 2 after the lattice, 3 after
// the coefficients (end of a forward phase), 4 start and 5 end of the gradient stage.  A one-call entry records all six;
// the two-phase entries record 0-3 (compute_rnnt_loss_fwd) and 4-5 (compute_rnnt_loss_bwd), and whatever the caller
// enqueues between the two calls is in neither stage.  One rnnt_profile_collect() reads what has been recorded since
// the last one as ONE step.
struct Profile {
    bool on = false;
    bool ready = false;
    hipEvent_t ev[6];
    double ms[5] = {0, 0, 0, 0, 0};   // statistics, lattice, coefficients, gradient, first event to last
    int calls = 0;
    bool pending = false;   // events of an asynchronous call recorded, not yet read
    bool has_fwd = false, has_bwd = false;
};
static Profile g_prof;

static bool prof_prepare() {
    if (!g_prof.on) return false;
    if (!g_prof.ready) {
        for (auto& e : g_prof.ev)
            if (hipEventCreate(&e) != hipSuccess) return false;
        g_prof.ready = true;
    }
    return true;
}

// mark(i) of the run_* functions: i = 0..4 are the boundaries of the four stages of one call
static void prof_mark(int i, bool do_fwd, bool do_bwd, hipStream_t stream) {
    if (i < 3) { if (do_fwd) (void)hipEventRecord(g_prof.ev[i], stream); return; }
    if (i == 3) {
        if (do_fwd) { (void)hipEventRecord(g_prof.ev[3], stream); g_prof.has_fwd = true; }
        if (do_bwd) (void)hipEventRecord(g_prof.ev[4], stream);
        return;
    }
    if (do_bwd) { (void)hipEventRecord(g_prof.ev[5], stream); g_prof.has_bwd = true; }
}

static void prof_accumulate() {
    float ms = 0.f;
    if (g_prof.has_fwd)
        for (int i = 0; i < 3; ++i)
            if (hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) == hipSuccess) g_prof.ms[i] += ms;
    if (g_prof.has_bwd && hipEventElapsedTime(&ms, g_prof.ev[4], g_prof.ev[5]) == hipSuccess) g_prof.ms[3] += ms;
    if ((g_prof.has_fwd || g_prof.has_bwd) &&
        hipEventElapsedTime(&ms, g_prof.ev[g_prof.has_fwd ? 0 : 4], g_prof.ev[g_prof.has_bwd ? 5 : 3]) == hipSuccess)
        g_prof.ms[4] += ms;
    g_prof.calls++;
    g_prof.pending = g_prof.has_fwd = g_prof.has_bwd = false;
}

// ----------------------------------------------------------------------------- tuning knobs
// The measured-best launch parameters.  A release build has exactly these constants; a development
// build (make dev: -DRNNT_DEV, lib/dev/libwarprnnt.so) can override them for A/B runs with
// RNNT_TUNE="key=value,key=value".  sw = waves per block of the row-stats kernel (2|4|8),
// nta = non-temporal stats loads, gmax = grid cap of the flat gradient kernel, rows = 1 forces
// the row-form gradient kernel, tile / tilekb = LDS-tile stats kernel on/off and its LDS budget,
// ppt = packets per thread of the flat gradient kernel.  Additive joint: jfnk / jgnk = columns
// per lane of the DF / DG kernels (0: widest the alignment allows), jfpf / jgpf = operand ping-pong;
// blk = block-per-row statistics kernel for rows >= 12 KB on/off, jzs = vocabulary split of the Z kernel (1|4|8),
// xcd = XCD-aware tile order of the short-row statistics kernel on/off, ctile = tiled coefficient kernel on/off,
// pskip = the gradient kernel's skip-padded-rows form for rows >= 8 KB on/off, joh = one-hot df corrections in the
// additive-joint DF kernel (-1: vocabularies <= 256), lat2 = two lattice columns per lane (-1: maxU > 256).
struct Tune { int sw = 4, nta = 1, gmax = 4194304, rows = 0, tile = 1, tilekb = 52, ppt = 2;
              int jfnk = 0, jfpf = 1, jgnk = 0, jgpf = 1, blk = 1, jzs = 0, xcd = 1, ctile = 1, pskip = 1, joh = -1;
              int lat2 = -1, xst = 0; };
#ifdef RNNT_DEV
static Tune read_tune() {
    Tune t;
    const char* e = getenv("RNNT_TUNE");
    if (e == nullptr) return t;
    const struct { const char* key; int* dst; } keys[] = {
        {"sw", &t.sw}, {"nta", &t.nta}, {"gmax", &t.gmax}, {"rows", &t.rows}, {"tile", &t.tile},
        {"tilekb", &t.tilekb}, {"ppt", &t.ppt}, {"jfnk", &t.jfnk}, {"jfpf", &t.jfpf}, {"jgnk", &t.jgnk},
        {"jgpf", &t.jgpf}, {"blk", &t.blk}, {"jzs", &t.jzs}, {"xcd", &t.xcd}, {"ctile", &t.ctile},
        {"pskip", &t.pskip}, {"joh", &t.joh}, {"lat2", &t.lat2}, {"xst", &t.xst}};
    // tokens are separated by ',', a token is key=value with the WHOLE key compared
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? static_cast<size_t>(end - p) : strlen(p);
        const char* eq = static_cast<const char*>(memchr(p, '=', len));
        if (eq != nullptr)
            for (const auto& k : keys)
                if (strlen(k.key) == static_cast<size_t>(eq - p) && strncmp(k.key, p, eq - p) == 0) *k.dst = atoi(eq + 1);
        p += len + (end ? 1 : 0);
    }
    return t;
}
static const Tune& tune() {
    static const Tune t = read_tune();     // function-local static: initialised once, thread-safe
    return t;
}
#else
static const Tune& tune() {
    static const Tune t;
    return t;
}
#endif

// ----------------------------------------------------------------------------- launch
// Everything one call needs: problem dimensions, the carved workspace, the stream.
template <typename C> struct Plan {
    int N, maxT, maxU, Up, A, blank;
    int cells_per_sample;          // maxT * maxU
    hipStream_t stream;
    const int *labels, *input_lengths, *label_lengths;
    LogPair<C>* lp2; C *logz, *alpha, *beta; Cell<C>* rowtab;
    double *offa, *offb, *llf, *llb;
    float *rowmax, *wmat, *side;
    size_t side_bytes = 0;
    int lat_cols = 1;              // lattice columns per lane (1 | 2), its wavefronts per block ...
    int lat_w = 1, lat_sh = 6;     // ... and the column -> wavefront shift (coefficient kernels)
    float fastemit = 0.0f;         // FastEmit lambda (extension entries only)
    const long long* offsets = nullptr;        // packed layout: cumulative row offsets (device, N+1 entries) ...
    unsigned long long packed_rows = 0;        // ... and the total number of rows (host)
    C* costs_dev;
    bool failed = false;
    void check() { if (hipGetLastError() != hipSuccess) failed = true; }
};

template <typename C>
static bool make_plan(Plan<C>& p, int A, int N, const rnntOptions& opt, void* workspace, const int* labels,
                      const int* label_lengths, const int* input_lengths, C* costs_device_out, bool joint = false) {
    (void)hipGetLastError();                       // a stale error of an unrelated earlier HIP call is not ours
    p.N = N; p.maxT = opt.maxT; p.maxU = opt.maxU; p.A = A; p.blank = opt.blank_label;
    if (p.blank < 0 || p.blank >= A) return false;
    if (p.maxU > 1024) return false;               // one lane per label position (as the reference)
    if (static_cast<long long>(p.maxT) * p.maxU > 0x7fffffffLL / 4) return false;
    if (N > 65535) return false;
    p.Up = ((p.maxU + 63) / 64) * 64;
    // one sample's skewed lp2 array is addressed through a buffer descriptor with a 32-bit size
    if (lat_rows(p.maxT, p.maxU) * p.Up * sizeof(LogPair<C>) >= (1ull << 31)) return false;
    // lattice kernel form: one wavefront for maxU <= 64; one column per lane while every wavefront of the block
    // has a SIMD to itself (maxU <= 256), two columns per lane beyond (measured, ns per diagonal at T = 1500,
    // one / two columns: U=128 79 / 97, U=192 93 / 111, U=256 120 / 117, U=301 148 / 139, U=512 205 / 188)
    const int lat2 = tune().lat2 >= 0 ? tune().lat2 : (p.Up > 256 ? 1 : 0);
    p.lat_cols = p.Up == 64 ? 1 : ((p.Up > 512 || lat2) ? 2 : 1);
    p.lat_w = lat_waves(p.Up, p.lat_cols);
    p.lat_sh = lat_col_shift(p.lat_cols);
    p.cells_per_sample = p.maxT * p.maxU;
    p.stream = reinterpret_cast<hipStream_t>(opt.stream);
    p.labels = labels; p.input_lengths = input_lengths; p.label_lengths = label_lengths;
    const Layout lay = make_layout(p.maxT, p.maxU, N, sizeof(C), joint);
    char* ws = reinterpret_cast<char*>(align_up(reinterpret_cast<size_t>(workspace)));
    p.lp2 = reinterpret_cast<LogPair<C>*>(ws + lay.lp2);
    p.logz = reinterpret_cast<C*>(ws + lay.logz);
    p.alpha = reinterpret_cast<C*>(ws + lay.alpha);
    p.rowtab = reinterpret_cast<Cell<C>*>(ws + lay.rowtab);
    p.beta = reinterpret_cast<C*>(ws + lay.beta);
    p.offa = reinterpret_cast<double*>(ws + lay.offa);
    p.offb = reinterpret_cast<double*>(ws + lay.offb);
    p.llf = reinterpret_cast<double*>(ws + lay.llf);
    p.llb = reinterpret_cast<double*>(ws + lay.llb);
    p.rowmax = reinterpret_cast<float*>(ws + lay.rowmax);
    p.wmat = reinterpret_cast<float*>(ws + lay.wmat);
    p.side = reinterpret_cast<float*>(ws + lay.side);
    p.side_bytes = lay.side_bytes;
    p.costs_dev = costs_device_out ? costs_device_out : reinterpret_cast<C*>(ws + lay.costs);
    return true;
}

// Stage 1 (materialised path): log-softmax statistics of every (b,t,u) row.
template <typename Tag>
static void launch_row_stats(Plan<typename Tag::comp>& p, const typename Tag::store* acts, int vec_ok) {
    using S = typename Tag::store;
    const Tune& tn = tune();
    const size_t row_bytes = static_cast<size_t>(p.A) * sizeof(S);
    if (tn.tile && vec_ok && row_bytes <= static_cast<size_t>(kTileMaxRowBytes)) {
        // short rows: LDS-tile kernel; smallest lane group G whose tile of 256/G rows fits the budget
        const size_t budget = static_cast<size_t>(tn.tilekb) * 1024;
        int G = 1;
        while (G < 64 && (256 / G) * row_bytes + 32 > budget) G *= 2;
        const int RT = 256 / G;
        const size_t lds = RT * row_bytes + 32;
        const unsigned long long Rall = p.offsets != nullptr ? p.packed_rows
                                                             : static_cast<unsigned long long>(p.N) * p.cells_per_sample;
        const unsigned tgrid = static_cast<unsigned>((Rall + RT - 1) / RT);
        if (lds <= 64 * 1024) {
            const unsigned xgrid = tn.xcd ? (tgrid + 7u) / 8u * 8u : tgrid;   // XCD remap wants a multiple of 8
#define RNNT_TILE(GG)                                                                                       \
    hipLaunchKernelGGL((row_stats_tile_kernel<Tag, GG>), dim3(xgrid), dim3(256), lds, p.stream, acts, p.labels, \
                       p.input_lengths, p.label_lengths, p.lp2, p.logz, Rall, p.maxT, p.maxU, p.Up,           \
                       p.A, p.blank, tn.xcd | (tn.xst << 4), p.offsets, p.N)
            switch (G) {
                case 1: RNNT_TILE(1); break;
                case 2: RNNT_TILE(2); break;
                case 4: RNNT_TILE(4); break;
                case 8: RNNT_TILE(8); break;
                case 16: RNNT_TILE(16); break;
                case 32: RNNT_TILE(32); break;
                default: RNNT_TILE(64); break;
            }
#undef RNNT_TILE
            p.check();
            return;
        }
    }
    // very long rows (>= 12 KB): one 256-thread block per row -- the rows in flight form one contiguous
    // window of the tensor, which streams like a flat read (measured 6.6-6.9 TB/s vs 6.1-6.4 for the
    // wavefront-per-row form on 20-32 KB rows; no gain at 8 KB, a loss below)
    if (tn.blk && vec_ok && row_bytes >= 12288 && p.cells_per_sample <= 0x7fffffff) {
        const dim3 bgrid(p.cells_per_sample, p.N);
        if (tn.nta)
            hipLaunchKernelGGL((row_stats_block_kernel<Tag, true, 4>), bgrid, dim3(256), 0, p.stream, acts, p.labels,
                               p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, vec_ok,
                               p.offsets, p.packed_rows);
        else
            hipLaunchKernelGGL((row_stats_block_kernel<Tag, false, 4>), bgrid, dim3(256), 0, p.stream, acts, p.labels,
                               p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, vec_ok,
                               p.offsets, p.packed_rows);
        p.check();
        return;
    }
    // long rows: one wavefront per row
#define RNNT_STATS(WV, NT)                                                                                       \
    hipLaunchKernelGGL((row_stats_kernel<Tag, WV, NT>), dim3((p.cells_per_sample + WV - 1) / WV, p.N), dim3(WV * 64), \
                       0, p.stream, acts, p.labels, p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, \
                       p.Up, p.A, p.blank, vec_ok, p.offsets, p.packed_rows)
    if (tn.nta) { if (tn.sw == 8) RNNT_STATS(8, true); else if (tn.sw == 2) RNNT_STATS(2, true); else RNNT_STATS(4, true); }
    else { if (tn.sw == 8) RNNT_STATS(8, false); else if (tn.sw == 2) RNNT_STATS(2, false); else RNNT_STATS(4, false); }
#undef RNNT_STATS
    p.check();
}

// Stage 2: alpha (and, for gradients, beta) recursion; writes the costs.
template <typename C> static void launch_lattice(Plan<C>& p, bool with_beta) {
    const int dirs = with_beta ? 2 : 1;
#define RNNT_LATTICE(MW, CC)                                                                                     \
    hipLaunchKernelGGL((lattice_kernel<C, MW, CC>), dim3(p.N * dirs), dim3(p.lat_w * 64), 0, p.stream, p.lp2,          \
                       p.alpha, p.beta, p.offa, p.offb, p.llf, p.llb, p.costs_dev, p.input_lengths, p.label_lengths,  \
                       p.maxT, p.maxU, p.Up, dirs)
    if (p.Up == 64) RNNT_LATTICE(1, 1);                            // one wavefront, no synchronisation
    else if (p.lat_cols == 1) RNNT_LATTICE(8, 1);                  // maxU <= 512, one column per lane
    else if (p.lat_w <= 4) RNNT_LATTICE(4, 2);                     // two columns per lane, one wavefront per SIMD
    else RNNT_LATTICE(8, 2);                                       // maxU <= 1024 in at most 8 wavefronts
#undef RNNT_LATTICE
    p.check();
}

// Stage 3: gradient coefficients per row into the natural-order row table.
template <typename C> static void launch_coef(Plan<C>& p, bool joint = false, bool onehot = false) {
    float* wmat = joint ? p.wmat : nullptr;
    const int Upad = joint_upad(p.maxU);
    // additive joint: W and CL planes always (the DF kernel takes its label corrections from CL); small
    // vocabularies add CB and replace the records by a plane of c
    const int planes = onehot ? joint_planes_onehot(p.maxU) : (joint ? 2 : 1);
    if (p.maxU <= 48 || !tune().ctile) {
        // small lattices: one thread per skewed cell, scattered record store
        const long long skew_cells = static_cast<long long>(p.maxT + p.maxU - 1) * p.Up;
        const dim3 cgrid(static_cast<unsigned>(((skew_cells + 255) / 256 + 7) / 8 * 8), p.N);   // multiple of 8: XCD-aware remap
        hipLaunchKernelGGL((coef_cell_kernel<C>), cgrid, dim3(256), 0, p.stream, p.lp2, p.logz, p.alpha, p.beta, p.offa,
                           p.offb, p.llf, p.labels, p.input_lengths, p.label_lengths, p.rowtab, p.maxT, p.maxU, p.Up,
                           wmat, Upad, p.fastemit, planes, p.offsets, p.lat_w, p.lat_sh);
    } else {
        const int DN = sizeof(C) == 4 ? 32 : 16;           // diagonals per tile (coef_kernel)
        const int tilesU = (p.maxU + 63) / 64, tilesN = (p.maxT + p.maxU - 1 + DN - 1) / DN;
        const dim3 cgrid(static_cast<unsigned>(tilesU * tilesN), p.N);
        hipLaunchKernelGGL((coef_kernel<C>), cgrid, dim3(256), 0, p.stream, p.lp2, p.logz, p.alpha, p.beta, p.offa,
                           p.offb, p.llf, p.labels, p.input_lengths, p.label_lengths, p.rowtab, p.maxT, p.maxU, p.Up,
                           wmat, Upad, tilesU, p.fastemit, planes, p.offsets, p.lat_w, p.lat_sh);
    }
    p.check();
}

// Stage 4 (materialised path): dense gradient write-back.
template <typename Tag>
static void launch_grad(Plan<typename Tag::comp>& p, const typename Tag::store* acts, typename Tag::store* grads,
                        const typename Tag::comp* grad_scale, int vec_ok) {
    using S = typename Tag::store;
    constexpr int V = Vec<Tag>::N;
    const Tune& tn = tune();
    const uintptr_t pa = reinterpret_cast<uintptr_t>(acts), pg = reinterpret_cast<uintptr_t>(grads);
    const size_t row_bytes = static_cast<size_t>(p.A) * sizeof(S);
    const bool packed = p.offsets != nullptr;
    const unsigned long long R = packed ? p.packed_rows : static_cast<unsigned long long>(p.N) * p.cells_per_sample;
    const unsigned long long E = R * p.A;
    const bool flat_ok = vec_ok && (pa & 15u) == 0 && (pg & 15u) == 0 && p.A <= (1 << 23) && (!tn.rows || packed);
    if (packed && !flat_ok) { p.failed = true; return; }   // (run_gpu has validated the alignment: not reached)
    if (flat_ok) {
        const unsigned long long npk = E / V;
        const int ppt = (tn.ppt == 1 || tn.ppt == 4) ? tn.ppt : 2;
        const unsigned long long cpk = static_cast<unsigned long long>(ppt) * 256;
        const unsigned long long nchunks = (npk + cpk - 1) / cpk;
        const unsigned grid = static_cast<unsigned>(nchunks < static_cast<unsigned long long>(tn.gmax)
                                                        ? (nchunks ? nchunks : 1) : tn.gmax);
        const unsigned long long stride = static_cast<unsigned long long>(grid) * cpk * V;
        const unsigned long long dq = stride / p.A;
        const int drem = static_cast<int>(stride % p.A);
        const float invA = 1.0f / static_cast<float>(p.A);
        // packed + per-sample scale: one scale per packed row, in the (by now dead) alpha array of the workspace
        using CC = typename Tag::comp;
        CC* rowscale = nullptr;
        if (packed && grad_scale) {
            rowscale = reinterpret_cast<CC*>(p.alpha);
            hipLaunchKernelGGL((fill_row_scale_kernel<CC>), dim3(p.N, 8), dim3(256), 0, p.stream, p.offsets, grad_scale,
                               rowscale, static_cast<long long>(p.packed_rows));
        }
#define RNNT_FLAT(SC, PP, PS)                                                                                       \
    hipLaunchKernelGGL((grad_flat_kernel<Tag, SC, PP, PS>), dim3(grid), dim3(256), 0, p.stream, acts, grads,        \
                       p.rowtab, grad_scale, E, R, p.A, p.blank, p.cells_per_sample, invA, dq, drem, rowscale)
        const bool padskip = tn.pskip && row_bytes >= 8192;     // skip reading padded rows only where rows are long
        if (grad_scale) { if (padskip) RNNT_FLAT(true, 2, true); else RNNT_FLAT(true, 2, false); }
        else if (ppt == 1) RNNT_FLAT(false, 1, false);
        else if (ppt == 4) RNNT_FLAT(false, 4, false);
        else if (padskip) RNNT_FLAT(false, 2, true);
        else RNNT_FLAT(false, 2, false);
#undef RNNT_FLAT
    } else {
        const dim3 rg((p.cells_per_sample + 3) / 4, p.N);
        if (grad_scale)
            hipLaunchKernelGGL((grad_rows_kernel<Tag, 4, true>), rg, dim3(256), 0, p.stream, acts, grads, p.rowtab,
                               grad_scale, p.maxT, p.maxU, p.A, p.blank, vec_ok);
        else
            hipLaunchKernelGGL((grad_rows_kernel<Tag, 4, false>), rg, dim3(256), 0, p.stream, acts, grads, p.rowtab,
                               grad_scale, p.maxT, p.maxU, p.A, p.blank, vec_ok);
    }
    p.check();
}

// The materialised path.  phases: bit 0 = forward part (row statistics, lattice and -- when gradients
// are wanted -- the coefficient table), bit 1 = gradient kernel; the two-call form
// Pageable host costs (what the reference's callers pass): the lattice kernel writes them into a small PINNED staging
// buffer of the calling thread -- N values, allocated on the thread's first such call, grown when a larger batch comes
// along, host memory only -- and the call copies them out after its stream synchronisation.  The alternative, a
// hipMemcpyAsync into pageable memory behind the last kernel, stages through the runtime's own pinned buffers and
// costs ~10 us of a 50 us call.  (Never freed: a thread_local destructor would call into the HIP runtime while the
// process tears it down.)
struct HostStage { void* host = nullptr; void* dev = nullptr; size_t cap = 0; };
static void* host_stage(size_t bytes, void** host_out) {
    static thread_local HostStage st;
    if (st.cap < bytes) {
        if (st.host != nullptr) (void)hipHostFree(st.host);
        st = HostStage{};
        size_t cap = 4096;
        while (cap < bytes) cap <<= 1;
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            if (h != nullptr) (void)hipHostFree(h);
            (void)hipGetLastError();
            return nullptr;                            // the caller falls back to the asynchronous copy
        }
        st.host = h; st.dev = d; st.cap = cap;
    }
    *host_out = st.host;
    return st.dev;
}

// (compute_rnnt_loss_fwd / _bwd) keeps only the workspace alive in between.  want_grad < 0: decided by
// `grads != nullptr` (the reference's "gradients == NULL means score only").
template <typename Tag>
static rnntStatus_t run_gpu(const typename Tag::store* acts, typename Tag::store* grads,
                            const int* labels, const int* label_lengths, const int* input_lengths,
                            int A, int N, typename Tag::comp* costs_host,
                            typename Tag::comp* costs_device_out, const typename Tag::comp* grad_scale,
                            void* workspace, const rnntOptions& opt, int phases = 3, int want_grad = -1,
                            float fastemit = 0.0f, const long long* offsets = nullptr, long long packed_rows = 0) {
    using S = typename Tag::store;
    using C = typename Tag::comp;
    Plan<C> p;
    // Host costs in PINNED memory (hipHostMalloc / hipHostRegister, e.g. a torch tensor with pin_memory=True) are
    // written by the lattice kernel directly: no copy behind the last kernel, only the stream synchronisation the
    // contract asks for.  Pageable memory, which the reference's callers pass, goes through the thread's pinned staging
    // buffer (host_stage above); the hipMemcpyAsync below is the fallback when that cannot be allocated.
    C* costs_direct = nullptr;
    C* costs_staged = nullptr;                         // host view of the staging buffer, when it is in use
    if (costs_host != nullptr && costs_device_out == nullptr) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, costs_host) == hipSuccess && attr.type == hipMemoryTypeHost &&
            attr.devicePointer != nullptr)
            costs_direct = static_cast<C*>(attr.devicePointer);
        (void)hipGetLastError();                       // (the query of a pageable pointer reports an error: not ours)
        if (costs_direct == nullptr && N > 0) {
            void* h = nullptr;
            costs_direct = static_cast<C*>(host_stage(sizeof(C) * static_cast<size_t>(N), &h));
            if (costs_direct != nullptr) costs_staged = static_cast<C*>(h);
        }
    }
    if (!make_plan(p, A, N, opt, workspace, labels, label_lengths, input_lengths,
                   costs_direct != nullptr ? costs_direct : costs_device_out))
        return RNNT_STATUS_INVALID_VALUE;
    if (!(fastemit >= 0.0f)) return RNNT_STATUS_INVALID_VALUE;
    p.fastemit = fastemit;
    if (offsets != nullptr) {
        // packed layout: the record table is sized for N*maxT*maxU rows, the packed tensor cannot have more
        if (packed_rows <= 0 || static_cast<unsigned long long>(packed_rows) >
                                    static_cast<unsigned long long>(N) * p.cells_per_sample)
            return RNNT_STATUS_INVALID_VALUE;
        p.offsets = offsets;
        p.packed_rows = static_cast<unsigned long long>(packed_rows);
    }
    const bool training = want_grad < 0 ? grads != nullptr : want_grad != 0;
    const bool do_fwd = (phases & 1) != 0, do_bwd = (phases & 2) != 0 && training;
    if (do_bwd && grads == nullptr) return RNNT_STATUS_INVALID_VALUE;

    // 16-byte packets need acts and grads rows to share their alignment phase.
    const uintptr_t pa = reinterpret_cast<uintptr_t>(acts), pg = reinterpret_cast<uintptr_t>(grads);
    int vec_ok = (pa % sizeof(S) == 0) ? 1 : 0;
    if (grads != nullptr && ((pa ^ pg) & 15u)) vec_ok = 0;
    // the packed layout has only the flat gradient kernel: both tensors 16-byte aligned
    if (p.offsets != nullptr && do_bwd && (!vec_ok || (pa & 15u) || (pg & 15u) || A > (1 << 23)))
        return RNNT_STATUS_INVALID_VALUE;

    const bool prof = prof_prepare();
    auto mark = [&](int i) { if (prof) prof_mark(i, do_fwd, do_bwd, p.stream); };

    mark(0);
    if (do_fwd) launch_row_stats<Tag>(p, acts, vec_ok);
    mark(1);
    if (do_fwd) launch_lattice(p, training);
    mark(2);
    if (do_fwd && training) launch_coef(p);
    mark(3);
    if (do_bwd) launch_grad<Tag>(p, acts, grads, grad_scale, vec_ok);
    mark(4);
    if (p.failed) return RNNT_STATUS_EXECUTION_FAILED;

    if (costs_host) {
        // the reference contract: costs in HOST memory, call returns after a stream sync (gpu_rnnt.h:208-213)
        if (costs_direct == nullptr &&
            hipMemcpyAsync(costs_host, p.costs_dev, sizeof(C) * N, hipMemcpyDeviceToHost, p.stream) != hipSuccess)
            return RNNT_STATUS_MEMOPS_FAILED;
        if (hipStreamSynchronize(p.stream) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
        if (costs_staged != nullptr) std::memcpy(costs_host, costs_staged, sizeof(C) * static_cast<size_t>(N));
        if (prof) prof_accumulate();
        // device-side lengths that do not fit the tensor (lattice_kernel marks the sample's cost): the same
        // status the CPU location returns for them (rnnt_cpu.cpp)
        for (int b = 0; b < N; ++b)
            if (is_cost_invalid<C>(costs_host[b])) return RNNT_STATUS_INVALID_VALUE;
    } else if (prof) {
        g_prof.pending = true;     // the caller synchronises, then calls rnnt_profile_collect()
    }
    return RNNT_STATUS_SUCCESS;
}

// ----------------------------------------------------------------------------- additive joint
// f (N,maxT,A) + g (N,maxU,A) -> costs, df, dg without the (N,T,U,A) tensor (rnnt_joint_kernels.h):
// the two streaming stages are replaced, lattice and coefficients are the same launches as above.
// Enqueue only: device costs, no host copy, no synchronisation.  Storage of f, g, df, dg by tag (fp32 / bf16 / fp16),
// arithmetic fp32.
// phases: 1 = forward (row maxima, Z, lattice, and with want_grad the coefficient table + W),
// 2 = backward (DF, DG, corrections from the workspace a forward call left), 3 = both.
template <typename Tag>
static rnntStatus_t run_gpu_joint(const typename Tag::store* f, const typename Tag::store* g, typename Tag::store* df,
                                  typename Tag::store* dg, const int* labels,
                                  const int* label_lengths, const int* input_lengths, int A, int N,
                                  float* costs_device, const float* grad_scale, void* workspace,
                                  const rnntOptions& opt, int phases, bool want_grad, float fastemit = 0.0f) {
    using S = typename Tag::store;
    Plan<float> p;
    if (!make_plan(p, A, N, opt, workspace, labels, label_lengths, input_lengths, costs_device, /*joint=*/true))
        return RNNT_STATUS_INVALID_VALUE;
    if (!(fastemit >= 0.0f)) return RNNT_STATUS_INVALID_VALUE;
    p.fastemit = fastemit;
    const bool do_fwd = (phases & 1) != 0, do_bwd = (phases & 2) != 0;
    if (do_bwd && (df == nullptr || dg == nullptr)) return RNNT_STATUS_INVALID_VALUE;
    // the gradient GEMMs address one sample's rows with 32-bit element offsets
    if (static_cast<long long>(p.maxT > p.maxU ? p.maxT : p.maxU) * A >= (1LL << 31) ||
        static_cast<long long>(N) * (p.maxT + p.maxU) >= (1LL << 31))
        return RNNT_STATUS_INVALID_VALUE;
    const bool training = want_grad;
    const bool prof = prof_prepare();
    auto mark = [&](int i) { if (prof) prof_mark(i, do_fwd, do_bwd, p.stream); };

    const int maxT = p.maxT, maxU = p.maxU;
    // rows made of whole 16-byte packets (row-maximum kernel) and 4-element loads aligned (Z kernel)
    const bool vec = (A % static_cast<int>(16 / sizeof(S)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0;
    const int tilesT = (maxT + 31) / 32, tilesU = (maxU + 31) / 32, tiles = tilesT * tilesU;
    mark(0);
    if (do_fwd) {   // row maxima, then the partition-function GEMM with the log-prob epilogue
        const long long rows = static_cast<long long>(N) * (maxT + maxU);
        const bool per_block = static_cast<size_t>(A) * sizeof(S) >= 12288;       // long rows: a block per row
        const dim3 rgrid(static_cast<unsigned>(per_block ? rows : (rows + 3) / 4));
#define RNNT_JMAX(VV, WW)                                                                                      \
    hipLaunchKernelGGL((joint_rowmax_kernel<Tag, VV, WW>), rgrid, dim3(256), 0, p.stream, f, g, input_lengths,       \
                       label_lengths, p.rowmax, maxT, maxU, A, N, training ? p.side : nullptr,                  \
                       static_cast<unsigned>(p.side_bytes / sizeof(float)))
        if (per_block) { if (vec) RNNT_JMAX(true, 4); else RNNT_JMAX(false, 4); }
        else { if (vec) RNNT_JMAX(true, 1); else RNNT_JMAX(false, 1); }
#undef RNNT_JMAX
        p.check();
        // vocabulary slices per tile: few tiles and a long contraction -> split it over 4 or 8 wavefronts
        const long long all_tiles = static_cast<long long>(N) * tiles;
        const int nchunk = (A + 31) / 32;
        int S = (all_tiles >= 4096 || nchunk < 16) ? 1 : ((all_tiles < 1024 && nchunk >= 32) ? 8 : 4);
        if (tune().jzs == 1 || tune().jzs == 4 || tune().jzs == 8) S = tune().jzs;
#define RNNT_JZ(SS, VV)                                                                                          \
    hipLaunchKernelGGL((joint_z_kernel<Tag, SS, VV>), dim3(SS == 1 ? ((tiles + 3) / 4 + 7) / 8 * 8 : tiles, N),                      \
                       dim3(SS == 1 ? 256 : SS * 64), 0, p.stream, f, g, p.rowmax, labels, input_lengths,        \
                       label_lengths, p.lp2, p.logz, maxT, maxU, p.Up, A, p.blank, tilesU, tiles, N)
        if (S == 1 && A <= kJointZSmallA && tune().jzs != 1)
            hipLaunchKernelGGL((joint_z_small_kernel<Tag>), dim3(((tiles + 3) / 4 + 7) / 8 * 8, N), dim3(256),
                               4 * kJointZSmallSlice * sizeof(float), p.stream, f, g, p.rowmax, labels,
                               input_lengths, label_lengths, p.lp2, p.logz, maxT, maxU, p.Up, A, p.blank, tilesU,
                               tiles, N);
        else if (S == 8) { if (vec) RNNT_JZ(8, true); else RNNT_JZ(8, false); }
        else if (S == 4) { if (vec) RNNT_JZ(4, true); else RNNT_JZ(4, false); }
        else { if (vec) RNNT_JZ(1, true); else RNNT_JZ(1, false); }
#undef RNNT_JZ
        p.check();
    }
    mark(1);
    if (do_fwd) launch_lattice(p, training);
    mark(2);
    // small vocabularies: the df corrections ride along in the DF GEMM as one-hot operands (3x its
    // contraction) instead of one global atomic per lattice cell in the fix-up kernel
    // (fp32 storage only: with 16-bit storage the one-hot DF kernel needs 228 + 128 registers and runs at a third of
    // the speed -- c4 shape 0.81 vs 0.51 ms for the backward phase -- so 16-bit keeps the epilogue corrections)
    const bool onehot = tune().joh >= 0 ? tune().joh != 0 : (A <= 256 && sizeof(S) == 4);
    // correction sums (fp32 side vectors in the workspace) for the epilogues of the gradient GEMMs
    float* sfb = p.side;
    float* sgb = sfb + static_cast<size_t>(N) * maxT;
    float* sgl = sgb + static_cast<size_t>(N) * maxU;
    int* farflag = reinterpret_cast<int*>(sgl + static_cast<size_t>(N) * maxU);
    const float* cplanes = onehot && joint_planes_onehot(maxU) == 4 ? p.wmat : nullptr;   // c / cb / cl as dense planes
    const dim3 fixgrid((maxU + 63) / 64, (maxT + kJointFixT - 1) / kJointFixT, N);
    if (do_fwd && training) {
        launch_coef(p, /*joint=*/true, onehot);
        hipLaunchKernelGGL(joint_sums_kernel, fixgrid, dim3(256), 0, p.stream, p.rowtab, input_lengths, label_lengths, sfb,
                           sgb, sgl, farflag, maxT, maxU, N, cplanes, joint_upad(maxU));
        p.check();
    }
    mark(3);
    if (do_bwd) {
        // gradient GEMMs with the corrections in their epilogues (plain stores of every element, padding
        // included), then the far cells (rare).
        // NK adjacent columns per lane = the widest vector the vocabulary size and alignment allow.
        const int Upad = joint_upad(maxU);
        const uintptr_t all4 = reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(g) |
                               reinterpret_cast<uintptr_t>(df) | reinterpret_cast<uintptr_t>(dg);
        const int NKmax = (A % 4 == 0 && (all4 & (4 * sizeof(S) - 1)) == 0 && A >= 96) ? 4
                        : (A % 2 == 0 && (all4 & (2 * sizeof(S) - 1)) == 0 && A >= 48) ? 2 : 1;
        const Tune& tn = tune();
        auto pick = [&](int want) { int nk = want > 0 ? want : NKmax; while (nk > NKmax) nk >>= 1; return nk; };
        const int NKf = pick(tn.jfnk), NKg = pick(tn.jgnk);
#define RNNT_JDF(NN, PP, OO)                                                                                     \
    hipLaunchKernelGGL((joint_df_kernel<Tag, NN, PP, OO>), dim3((A + 128 * NN - 1) / (128 * NN), tilesT, N), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, labels, input_lengths, label_lengths, df, maxT,  \
                       maxU, Upad, A, N, p.blank, sfb)
#define RNNT_JDG(NN, PP)                                                                                         \
    hipLaunchKernelGGL((joint_dg_kernel<Tag, NN, PP>), dim3((A + 128 * NN - 1) / (128 * NN), tilesU, N), dim3(256), 0, \
                       p.stream, f, g, p.rowmax, p.wmat, grad_scale, input_lengths, label_lengths, dg, maxT,     \
                       maxU, Upad, A, N, labels, p.blank, sgb, sgl)
        // 16-bit storage: the operand ping-pong doubles the AGPR count of these kernels (172 + 128 registers: one wavefront
        // per SIMD at four columns per lane); without it they keep two (DF) / three (DG) wavefronts per SIMD
        const bool pf_f = tn.jfpf != 0 && (sizeof(S) == 4 || NKf < 4), pf_g = tn.jgpf != 0 && (sizeof(S) == 4 || NKg < 4);
        if (onehot)       { if (NKf == 4) RNNT_JDF(4, true, true); else if (NKf == 2) RNNT_JDF(2, true, true); else RNNT_JDF(1, true, true); }
        else if (pf_f) { if (NKf == 4) RNNT_JDF(4, true, false); else if (NKf == 2) RNNT_JDF(2, true, false); else RNNT_JDF(1, true, false); }
        else              { if (NKf == 4) RNNT_JDF(4, false, false); else if (NKf == 2) RNNT_JDF(2, false, false); else RNNT_JDF(1, false, false); }
        if (pf_g) { if (NKg == 4) RNNT_JDG(4, true); else if (NKg == 2) RNNT_JDG(2, true); else RNNT_JDG(1, true); }
        else         { if (NKg == 4) RNNT_JDG(4, false); else if (NKg == 2) RNNT_JDG(2, false); else RNNT_JDG(1, false); }
#undef RNNT_JDF
#undef RNNT_JDG
        p.check();
        hipLaunchKernelGGL((joint_far_kernel<Tag>), fixgrid, dim3(256), 0, p.stream, f, g, p.rowmax, p.rowtab, grad_scale,
                           input_lengths, label_lengths, farflag, df, dg, maxT, maxU, A, N, cplanes, Upad);
        p.check();
    }
    mark(4);
    if (prof) g_prof.pending = true;
    return p.failed ? RNNT_STATUS_EXECUTION_FAILED : RNNT_STATUS_SUCCESS;
}

static bool bad_args(const void* acts, const int* labels, const int* label_lengths,
                     const int* input_lengths, const void* costs, const void* workspace, int A, int N,
                     const rnntOptions& o) {
    // reference src/rnnt_entrypoint.cpp:49-59
    return acts == nullptr || labels == nullptr || label_lengths == nullptr || input_lengths == nullptr ||
           costs == nullptr || workspace == nullptr || A <= 0 || N <= 0 || o.maxT <= 0 || o.maxU <= 0;
}

}  // namespace rnnt

using namespace rnnt;

// The library is compiled with -fvisibility=hidden: only the C entry points of include/rnnt.h are exported
// (the reference exports exactly its five: nm -D of its libwarprnnt.so).
#pragma GCC visibility push(default)
extern "C" {

int get_warprnnt_version() { return 1; }

const char* rnntGetStatusString(rnntStatus_t status) {
    // Same strings as the reference (src/rnnt_entrypoint.cpp:18-35) so log scrapers keep working.
    switch (status) {
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
    if (options.loc == RNNT_CPU)
        return cpu_rnnt_f32(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, workspace, options);
    if (options.loc == RNNT_GPU)
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
    if (options.loc == RNNT_CPU)
        return cpu_rnnt_f64(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, workspace, options);
    if (options.loc == RNNT_GPU)
        return run_gpu<F64>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                            minibatch, costs, nullptr, nullptr, workspace, options);
    return RNNT_STATUS_INVALID_VALUE;
}

rnntStatus_t compute_rnnt_loss_bf16(const uint16_t* const activations, uint16_t* gradients,
                                    const int* const flat_labels, const int* const label_lengths,
                                    const int* const input_lengths, int alphabet_size, int minibatch,
                                    float* costs, void* workspace, rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options) || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu<BF16>(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                         minibatch, costs, nullptr, nullptr, workspace, options);
}

rnntStatus_t compute_rnnt_loss_fp16(const uint16_t* const activations, uint16_t* gradients,
                                    const int* const flat_labels, const int* const label_lengths,
                                    const int* const input_lengths, int alphabet_size, int minibatch,
                                    float* costs, void* workspace, rnntOptions options) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs, workspace, alphabet_size,
                 minibatch, options) || options.loc != RNNT_GPU)
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
                 alphabet_size, minibatch, options) || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, grad_scale_device, workspace, options, dtype_code, 3, -1);
}

rnntStatus_t compute_rnnt_loss_fwd(const void* activations, const int* const flat_labels,
                                   const int* const label_lengths, const int* const input_lengths,
                                   int alphabet_size, int minibatch, void* costs_device, void* workspace,
                                   rnntOptions options, int dtype_code, int prepare_backward) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                     costs_device, nullptr, workspace, options, dtype_code, 1, prepare_backward ? 1 : 0);
}

rnntStatus_t compute_rnnt_loss_bwd(const void* activations, void* gradients, const void* grad_scale_device,
                                   int alphabet_size, int minibatch, void* workspace, rnntOptions options,
                                   int dtype_code) {
    if (activations == nullptr || gradients == nullptr || workspace == nullptr || alphabet_size <= 0 ||
        minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, nullptr, nullptr, nullptr, alphabet_size, minibatch, nullptr,
                     grad_scale_device, workspace, options, dtype_code, 2, 1);
}

rnntStatus_t compute_rnnt_loss_fastemit(const void* activations, void* gradients, const int* const flat_labels,
                                        const int* const label_lengths, const int* const input_lengths,
                                        int alphabet_size, int minibatch, void* costs_device,
                                        const void* grad_scale_device, void* workspace, rnntOptions options,
                                        int dtype_code, float fastemit_lambda) {
    if (bad_args(activations, flat_labels, label_lengths, input_lengths, costs_device, workspace,
                 alphabet_size, minibatch, options) || options.loc != RNNT_GPU)
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
                 alphabet_size, minibatch, options) || options.loc != RNNT_GPU)
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
    if (options.loc == RNNT_CPU) {
        // the reference's CPU contract on the packed rows: every array on the host (row_offsets and costs too),
        // log-probabilities in, sparse log-prob gradients out (rnnt_cpu.cpp); fp32 / fp64, no scale, no FastEmit
        if (dtype_code > 1 || dtype_code < 0 || grad_scale_device != nullptr || fastemit_lambda != 0.0f ||
            total_rows <= 0)
            return RNNT_STATUS_INVALID_VALUE;
        return cpu_rnnt_packed(activations, gradients, flat_labels, label_lengths, input_lengths, row_offsets,
                               alphabet_size, minibatch, costs_device, workspace, options, dtype_code == 1);
    }
    if (options.loc != RNNT_GPU) return RNNT_STATUS_INVALID_VALUE;
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
                 alphabet_size, minibatch, options) || row_offsets == nullptr || options.loc != RNNT_GPU)
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
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_async(activations, gradients, nullptr, nullptr, nullptr, alphabet_size, minibatch, nullptr,
                     grad_scale_device, workspace, options, dtype_code, 2, 1, 0.0f, row_offsets, total_rows);
}

rnntStatus_t compute_rnnt_loss_add(const float* const trans_acts, const float* const pred_acts,
                                   float* trans_grads, float* pred_grads, const int* const flat_labels,
                                   const int* const label_lengths, const int* const input_lengths,
                                   int alphabet_size, int minibatch, float* costs_device, void* workspace,
                                   rnntOptions options) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    if ((trans_grads == nullptr) != (pred_grads == nullptr)) return RNNT_STATUS_INVALID_VALUE;
    const bool training = trans_grads != nullptr;
    return run_gpu_joint<F32>(trans_acts, pred_acts, trans_grads, pred_grads, flat_labels, label_lengths,
                         input_lengths, alphabet_size, minibatch, costs_device, nullptr, workspace, options,
                         training ? 3 : 1, training);
}

rnntStatus_t compute_rnnt_loss_add_fwd(const float* const trans_acts, const float* const pred_acts,
                                       const int* const flat_labels, const int* const label_lengths,
                                       const int* const input_lengths, int alphabet_size, int minibatch,
                                       float* costs_device, void* workspace, rnntOptions options,
                                       int prepare_backward) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, nullptr, nullptr, flat_labels, label_lengths, input_lengths,
                         alphabet_size, minibatch, costs_device, nullptr, workspace, options, 1,
                         prepare_backward != 0);
}

rnntStatus_t compute_rnnt_loss_add_fwd_fastemit(const float* const trans_acts, const float* const pred_acts,
                                                const int* const flat_labels, const int* const label_lengths,
                                                const int* const input_lengths, int alphabet_size, int minibatch,
                                                float* costs_device, void* workspace, rnntOptions options,
                                                int prepare_backward, float fastemit_lambda) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, nullptr, nullptr, flat_labels, label_lengths, input_lengths,
                         alphabet_size, minibatch, costs_device, nullptr, workspace, options, 1,
                         prepare_backward != 0, fastemit_lambda);
}

rnntStatus_t compute_rnnt_loss_add_bwd(const float* const trans_acts, const float* const pred_acts,
                                       float* trans_grads, float* pred_grads, const float* grad_scale_device,
                                       const int* const flat_labels, const int* const label_lengths,
                                       const int* const input_lengths, int alphabet_size, int minibatch,
                                       void* workspace, rnntOptions options) {
    if (trans_acts == nullptr || pred_acts == nullptr || trans_grads == nullptr || pred_grads == nullptr ||
        flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr || workspace == nullptr ||
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    return run_gpu_joint<F32>(trans_acts, pred_acts, trans_grads, pred_grads, flat_labels, label_lengths,
                         input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device, workspace, options, 2,
                         true);
}

// Additive joint with the activations' storage type as an argument (0 fp32, 2 bf16, 3 fp16; fp64 is not offered).
rnntStatus_t compute_rnnt_loss_add_fwd_dt(const void* trans_acts, const void* pred_acts, const int* const flat_labels,
                                          const int* const label_lengths, const int* const input_lengths,
                                          int alphabet_size, int minibatch, float* costs_device, void* workspace,
                                          rnntOptions options, int dtype_code, int prepare_backward,
                                          float fastemit_lambda) {
    if (bad_args(trans_acts, flat_labels, label_lengths, input_lengths, costs_device, workspace, alphabet_size,
                 minibatch, options) || pred_acts == nullptr || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    const bool pb = prepare_backward != 0;
    switch (dtype_code) {
        case 0: return run_gpu_joint<F32>(static_cast<const float*>(trans_acts), static_cast<const float*>(pred_acts), nullptr,
                                          nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                          costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        case 2: return run_gpu_joint<BF16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                           nullptr, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                           costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        case 3: return run_gpu_joint<F16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                          nullptr, nullptr, flat_labels, label_lengths, input_lengths, alphabet_size, minibatch,
                                          costs_device, nullptr, workspace, options, 1, pb, fastemit_lambda);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

rnntStatus_t compute_rnnt_loss_add_bwd_dt(const void* trans_acts, const void* pred_acts, void* trans_grads,
                                          void* pred_grads, const float* grad_scale_device,
                                          const int* const flat_labels, const int* const label_lengths,
                                          const int* const input_lengths, int alphabet_size, int minibatch,
                                          void* workspace, rnntOptions options, int dtype_code) {
    if (trans_acts == nullptr || pred_acts == nullptr || trans_grads == nullptr || pred_grads == nullptr ||
        flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr || workspace == nullptr ||
        alphabet_size <= 0 || minibatch <= 0 || options.maxT <= 0 || options.maxU <= 0 || options.loc != RNNT_GPU)
        return RNNT_STATUS_INVALID_VALUE;
    switch (dtype_code) {
        case 0: return run_gpu_joint<F32>(static_cast<const float*>(trans_acts), static_cast<const float*>(pred_acts),
                                          static_cast<float*>(trans_grads), static_cast<float*>(pred_grads), flat_labels,
                                          label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                          workspace, options, 2, true);
        case 2: return run_gpu_joint<BF16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                           static_cast<uint16_t*>(trans_grads), static_cast<uint16_t*>(pred_grads), flat_labels,
                                           label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                           workspace, options, 2, true);
        case 3: return run_gpu_joint<F16>(static_cast<const uint16_t*>(trans_acts), static_cast<const uint16_t*>(pred_acts),
                                          static_cast<uint16_t*>(trans_grads), static_cast<uint16_t*>(pred_grads), flat_labels,
                                          label_lengths, input_lengths, alphabet_size, minibatch, nullptr, grad_scale_device,
                                          workspace, options, 2, true);
        default: return RNNT_STATUS_INVALID_VALUE;
    }
}

void rnnt_profile_enable(int on) { g_prof.on = on != 0; }

void rnnt_profile_collect(void) {
    if (g_prof.on && g_prof.ready && g_prof.pending) prof_accumulate();
}

void rnnt_profile_reset(void) {
    for (double& m : g_prof.ms) m = 0.0;
    g_prof.calls = 0;
}

int rnnt_profile_read(double* ms, int n) {
    for (int i = 0; i < n && i < 5; ++i) ms[i] = g_prof.ms[i];
    return g_prof.calls;
}

}  // extern "C"
#pragma GCC visibility pop
