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
#include "rnnt_cpu.h"
#include "rnnt_host.h"

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

// Short rows under a wide lattice take the 2-D cell-tile statistics kernel -- when the tensor allows its covering packets
// (first and last byte on 16-byte boundaries).  The one rule of the dispatch that looks at the batch size: a half of the
// two-half schedule is told the whole batch's answer (Plan::stats_tile2d) instead of asking for itself.
template <typename Tag>
static bool stats_is_tile2d(const Plan<typename Tag::comp>& p, const typename Tag::store* acts) {
    const size_t row_bytes = static_cast<size_t>(p.A) * sizeof(typename Tag::store);
    if (!tune().tile2d || p.offsets != nullptr || row_bytes % 8 != 0 || row_bytes > 208 || p.maxU < 64) return false;
    if (p.stats_tile2d >= 0) return p.stats_tile2d == 1;
    return (reinterpret_cast<uintptr_t>(acts) & 15u) == 0 &&
           (static_cast<unsigned long long>(p.N) * p.cells_per_sample * row_bytes) % 16 == 0;
}

// Stage 1 (materialised path): log-softmax statistics of every (b,t,u) row.
template <typename Tag>
static void launch_row_stats(Plan<typename Tag::comp>& p, const typename Tag::store* acts, int vec_ok) {
    using S = typename Tag::store;
    const Tune& tn = tune();
    const size_t row_bytes = static_cast<size_t>(p.A) * sizeof(S);
    // short rows under a wide lattice (c4): 2-D cell tiles, results stored along the anti-diagonals
    // (the kernel loads the aligned 16-byte packets that COVER a piece of rows: with the tensor's first and last byte on
    // 16-byte boundaries no packet reaches outside it, whatever phase the pieces inside have)
    if (stats_is_tile2d<Tag>(p, acts)) {
        // tile shape: 16 x 16 (128-byte runs along the anti-diagonals, pieces of 16 rows; the default: 1.053 against 1.089 ms on c4,
        // alternating inside one process, tools/c4_align_probe.py) | 8 x 32 (64-byte runs, pieces of 32 rows)
        const bool sq = tn.tile2d == 2;
        const int TT = sq ? 16 : 8, TU = sq ? 16 : 32;
        const int tilesT = (p.maxT + TT - 1) / TT, tilesU = (p.maxU + TU - 1) / TU;
        const unsigned long long ntile = static_cast<unsigned long long>(p.N) * tilesT * tilesU;
        const int piece = (static_cast<int>(TU * row_bytes) + 15 + 15) / 16 * 16;      // covering packets of a piece at any phase
        if (ntile < (1ull << 30)) {
            const int order = tn.t2ord;
            const unsigned long long pts = static_cast<unsigned long long>(tilesT) * tilesU;
            const unsigned xgrid = (order & 3) == 2 ? static_cast<unsigned>(static_cast<unsigned long long>(p.N) * 8 * ((pts + 7) / 8))
                                              : static_cast<unsigned>((ntile + 7) / 8 * 8);
            // (the 256 results overlay the tile: TT rows of TU + 1 {pair, log Z} records of the lattice type)
            const size_t lds2 = static_cast<size_t>(TT) * piece > 8192 ? static_cast<size_t>(TT) * piece : 8192;
#ifdef RNNT_DEV
#define RNNT_TILE2D_POISON (tn.xst == 2 ? static_cast<int*>(nullptr) : p.poison)      /* xst=2: natural-order result stores, timing only */
#else
#define RNNT_TILE2D_POISON p.poison
#endif
#define RNNT_TILE2D(T1, U1)                                                                                       \
    hipLaunchKernelGGL((row_stats_tile2d_kernel<Tag, T1, U1>), dim3(xgrid), dim3(256), lds2, p.stream, acts, p.labels, \
                       p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, p.N, tilesT, \
                       tilesU, piece, RNNT_TILE2D_POISON, order)
            if (sq) RNNT_TILE2D(16, 16); else RNNT_TILE2D(8, 32);
#undef RNNT_TILE2D
#undef RNNT_TILE2D_POISON
            p.check();
            return;
        }
    }
    if (tn.tile && vec_ok && row_bytes <= static_cast<size_t>(tn.tilemax)) {
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
                       p.A, p.blank, tn.xcd | (tn.xst << 4), p.offsets, p.N, p.poison)
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
        for (int b0 = 0; b0 < p.N; b0 += kGridSamples) {      // (samples on gridDim.y: slices of the batch)
            const dim3 bgrid(p.cells_per_sample, p.N - b0 < kGridSamples ? p.N - b0 : kGridSamples);
#ifdef RNNT_DEV
            if (!tn.nta)
                hipLaunchKernelGGL((row_stats_block_kernel<Tag, false, 4>), bgrid, dim3(256), 0, p.stream, acts, p.labels,
                                   p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, vec_ok,
                                   p.offsets, p.packed_rows, b0, p.poison);
            else
#endif
                hipLaunchKernelGGL((row_stats_block_kernel<Tag, true, 4>), bgrid, dim3(256), 0, p.stream, acts, p.labels,
                                   p.input_lengths, p.label_lengths, p.lp2, p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, vec_ok,
                                   p.offsets, p.packed_rows, b0, p.poison);
        }
        p.check();
        return;
    }
    // long rows: one wavefront per row
#define RNNT_STATS(WV, NT)                                                                                       \
    for (int b0 = 0; b0 < p.N; b0 += kGridSamples)                                                               \
        hipLaunchKernelGGL((row_stats_kernel<Tag, WV, NT>),                                                      \
                           dim3((p.cells_per_sample + WV - 1) / WV, p.N - b0 < kGridSamples ? p.N - b0 : kGridSamples), \
                           dim3(WV * 64), 0, p.stream, acts, p.labels, p.input_lengths, p.label_lengths, p.lp2,  \
                           p.logz, p.maxT, p.maxU, p.Up, p.A, p.blank, vec_ok, p.offsets, p.packed_rows, b0, p.poison)
    // (the forms a release build never selects exist in the development build only: less device code to load)
#ifdef RNNT_DEV
    if (tn.nta) { if (tn.sw == 8) RNNT_STATS(8, true); else if (tn.sw == 2) RNNT_STATS(2, true); else RNNT_STATS(4, true); }
    else { if (tn.sw == 8) RNNT_STATS(8, false); else if (tn.sw == 2) RNNT_STATS(2, false); else RNNT_STATS(4, false); }
#else
    RNNT_STATS(4, true);
#endif
#undef RNNT_STATS
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
#ifdef RNNT_DEV
        const int ppt = (tn.ppt == 1 || tn.ppt == 4) ? tn.ppt : 2;
#else
        const int ppt = 2;
#endif
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
                       p.rowtab, grad_scale, E, R, p.A, p.blank, p.cells_per_sample, invA, dq, drem, rowscale, p.padflag)
        // padded rows are not read: always for long rows (the record is asked for first), for shorter ones when the
        // coefficient kernel has seen padding in this batch (packed layout: there are no padded rows)
        const int padskip = (!tn.pskip || packed) ? 0 : row_bytes >= static_cast<size_t>(tn.pskipb) ? 1
                                                       : row_bytes >= static_cast<size_t>(tn.pskipmin) ? 2 : 0;
        if (grad_scale && rowscale) RNNT_FLAT(2, 2, 0);
        else if (grad_scale) { if (padskip == 1) RNNT_FLAT(1, 2, 1); else if (padskip == 2) RNNT_FLAT(1, 2, 2); else RNNT_FLAT(1, 2, 0); }
#ifdef RNNT_DEV
        else if (ppt == 1) RNNT_FLAT(0, 1, 0);
        else if (ppt == 4) RNNT_FLAT(0, 4, 0);
#endif
        else if (padskip == 1) RNNT_FLAT(0, 2, 1);
        else if (padskip == 2) RNNT_FLAT(0, 2, 2);
        else RNNT_FLAT(0, 2, 0);
#undef RNNT_FLAT
    } else {
        for (int b0 = 0; b0 < p.N; b0 += kGridSamples) {
            const dim3 rg((p.cells_per_sample + 3) / 4, p.N - b0 < kGridSamples ? p.N - b0 : kGridSamples);
            if (grad_scale)
                hipLaunchKernelGGL((grad_rows_kernel<Tag, 4, true>), rg, dim3(256), 0, p.stream, acts, grads, p.rowtab,
                                   grad_scale, p.maxT, p.maxU, p.A, p.blank, vec_ok, b0);
            else
                hipLaunchKernelGGL((grad_rows_kernel<Tag, 4, false>), rg, dim3(256), 0, p.stream, acts, grads, p.rowtab,
                                   grad_scale, p.maxT, p.maxU, p.A, p.blank, vec_ok, b0);
        }
    }
    p.check();
}

// Host costs.  The contract of the reference (and the default here): costs is a HOST array, the library allocates
// nothing, the N values are copied behind the last kernel (hipMemcpyAsync, then the stream synchronisation:
// gpu_rnnt.h:208-213).  Two faster routes exist, neither of which allocates by default:
//   * costs in PINNED memory (hipHostMalloc / hipHostRegister, a torch tensor with pin_memory=True): the lattice
//     kernel writes them directly, no copy at all;
//   * OPT-IN staging (rnnt_host_staging(1) or WARPRNNT_HOST_STAGING=1): pageable costs go through a small pinned
//     buffer of the calling thread (the copy into pageable memory stages through the runtime's own pinned buffers and
//     costs ~10 us of a 50 us call).  This is the ONLY memory the library can ever allocate, host memory only,
//     at most kStageCap bytes per calling thread, counted (rnnt_host_staging_bytes) and releasable
//     (rnnt_host_staging_release); larger batches fall back to the copy.
struct HostStage {
    void* host = nullptr; void* dev = nullptr; size_t cap = 0; int device = -1;
    std::atomic<bool> busy{false};
};
constexpr size_t kStageCap = 1u << 20;                 // bytes per calling thread
static std::mutex g_stage_mu;
static std::vector<HostStage*> g_stage_all;            // every thread's buffer (for the release call); entries are never removed
static std::atomic<int> g_stage_mode{-1};              // -1: not decided yet (environment), 0 off, 1 on
static std::atomic<long long> g_stage_bytes{0};

static bool stage_enabled() {
    int m = g_stage_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("WARPRNNT_HOST_STAGING");
        m = (e != nullptr && atoi(e) > 0) ? 1 : 0;
        g_stage_mode.store(m, std::memory_order_relaxed);
    }
    return m == 1;
}

static void stage_free(HostStage* st) {                // (g_stage_mu held, or the owner thread with busy set)
    if (st->host != nullptr) {
        (void)hipHostFree(st->host);
        g_stage_bytes.fetch_sub(static_cast<long long>(st->cap), std::memory_order_relaxed);
    }
    st->host = st->dev = nullptr; st->cap = 0; st->device = -1;
}

// Returns the thread's staging record with `busy` set (the caller clears it), or nullptr: staging off, batch too
// large, or the allocation failed -- the caller then uses the asynchronous copy.
static HostStage* stage_acquire(size_t bytes) {
    if (!stage_enabled() || bytes > kStageCap) return nullptr;
    static thread_local HostStage* st = nullptr;
    if (st == nullptr) {
        st = new HostStage;                             // lives as long as the process: a thread_local destructor would
        std::lock_guard<std::mutex> g(g_stage_mu);      // call into the HIP runtime while the process tears it down
        g_stage_all.push_back(st);
    }
    std::lock_guard<std::mutex> g(g_stage_mu);          // (uncontended: taken per call only while staging is on)
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (st->cap < bytes) {
        stage_free(st);
        size_t cap = 4096;
        while (cap < bytes) cap <<= 1;
        void* h = nullptr;
        if (hipHostMalloc(&h, cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        st->host = h; st->cap = cap; st->device = -1;
        g_stage_bytes.fetch_add(static_cast<long long>(cap), std::memory_order_relaxed);
    }
    if (st->device != device) {                         // the device alias belongs to the CURRENT device
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, st->host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        st->dev = d; st->device = device;
    }
    st->busy.store(true, std::memory_order_relaxed);
    return st;
}

// TWO-HALF SCHEDULE (long lattices).  The lattice kernel is a dependent chain -- 1800 anti-diagonals x 139 ns on
// N=64,T=1500,U=301 = 0.27 ms during which 128 small blocks hold the device and HBM idles -- and it sits between the two
// streaming stages.  When the caller has handed the library a second stream (rnnt_set_aux_stream: the library creates none),
// the batch is split into two halves of samples and the lattice of one half runs on that stream WHILE the caller's stream
// streams the other half:
//     caller's stream:  stats(h0) | stats(h1)          | coef(h0) grad(h0)       | coef(h1) grad(h1)
//     auxiliary stream:           | lattice(h0)        | lattice(h1)             |
// (fork / join through four events; capturable: the auxiliary stream joins a capture through its first wait).  Samples are
// independent and every per-sample array of the workspace is indexed by the sample, so a half is the same Plan with its
// pointers advanced (sub_plan).  Used for lattices of kOverlapMinDiagonals diagonals and more: below, the lattice is a few
// microseconds and the four extra launches cost more than it.
constexpr int kOverlapMinDiagonals = 768;
struct AuxStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev[4];
    int made = 0;                               // events of ev[] that exist ...
    int device = -1;                            // ... and the device they belong to (HIP events are bound to their device)
};
static thread_local AuxStream t_aux;            // per calling thread, like options.stream is per call

static void aux_drop_events() {
    for (int i = 0; i < t_aux.made; ++i) (void)hipEventDestroy(t_aux.ev[i]);
    (void)hipGetLastError();
    t_aux.made = 0; t_aux.device = -1;
}

// The fork / join events of the calling thread, on the CURRENT device: a thread that moves to another GPU (the stream it
// hands over "must belong to the device of the call") gets events of that GPU -- the old ones are destroyed, not leaked.
// false: no auxiliary stream, or the events cannot be made -> the caller runs the one-stream schedule (nothing has
// been launched yet, so a failure here can never leave half a fork behind).
static bool aux_prepare() {
    if (t_aux.stream == nullptr) return false;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (t_aux.made == 4 && t_aux.device == dev) return true;
    aux_drop_events();
    for (; t_aux.made < 4; ++t_aux.made)
        if (hipEventCreateWithFlags(&t_aux.ev[t_aux.made], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            aux_drop_events();
            return false;
        }
    t_aux.device = dev;
    return true;
}

// samples [b0, b0 + n) of plan p as a plan of their own
template <typename C> static Plan<C> sub_plan(const Plan<C>& p, int b0, int n) {
    Plan<C> q = p;
    const size_t Dp = lat_rows(p.maxT, p.maxU), sk = static_cast<size_t>(b0) * Dp * p.Up;
    q.N = n;
    q.labels = p.labels + static_cast<size_t>(b0) * (p.maxU - 1);
    q.input_lengths = p.input_lengths + b0;
    q.label_lengths = p.label_lengths + b0;
    q.lp2 = p.lp2 + sk; q.logz = p.logz + sk; q.alpha = p.alpha + sk; q.beta = p.beta + sk;
    q.rowtab = p.rowtab + static_cast<size_t>(b0) * p.cells_per_sample;
    q.offa = p.offa + static_cast<size_t>(b0) * p.lat_w * Dp;
    q.offb = p.offb + static_cast<size_t>(b0) * p.lat_w * Dp;
    q.llf = p.llf + b0; q.llb = p.llb + b0; q.poison = p.poison + b0;
    q.costs_dev = p.costs_dev + b0;
    return q;
}

// The materialised path.  phases: bit 0 = forward part (row statistics, lattice and -- when gradients
// are wanted -- the coefficient table), bit 1 = gradient kernel; the two-call form
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
    // Host costs in PINNED memory are written by the lattice kernel directly: no copy behind the last kernel, only the
    // stream synchronisation the contract asks for.  Pageable memory, which the reference's callers pass, is copied
    // behind the last kernel as the reference does -- or, when the caller has opted in, goes through the thread's
    // pinned staging buffer (stage_acquire above).
    C* costs_direct = nullptr;
    HostStage* stage = nullptr;                        // the staging record, when it is in use
    struct StageGuard { HostStage*& s; ~StageGuard() { if (s != nullptr) s->busy.store(false, std::memory_order_relaxed); } } stage_guard{stage};
    if (costs_host != nullptr && costs_device_out == nullptr) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, costs_host) == hipSuccess && attr.type == hipMemoryTypeHost &&
            attr.devicePointer != nullptr)
            costs_direct = static_cast<C*>(attr.devicePointer);
        (void)hipGetLastError();                       // (the query of a pageable pointer reports an error: not ours)
        if (costs_direct == nullptr && N > 0) {
            stage = stage_acquire(sizeof(C) * static_cast<size_t>(N));
            if (stage != nullptr) costs_direct = static_cast<C*>(stage->dev);
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

    std::unique_lock<std::mutex> prof_lock;
    if (g_prof.on.load(std::memory_order_relaxed)) prof_lock = std::unique_lock<std::mutex>(g_prof_mu);
    const bool prof = prof_prepare(prof_lock.owns_lock());
    const bool ranges = ranges_prepare();
    static const char* const kStages[4] = {"warprnnt:row_stats", "warprnnt:lattice", "warprnnt:coefficients", "warprnnt:gradient"};
    auto mark = [&](int i) {
        if (prof) prof_mark(i, do_fwd, do_bwd, p.stream);
        if (ranges) ranges_mark(i, do_fwd, do_bwd, kStages);
    };

    // (one-call training entries only: measured slower for the forward half of a two-phase pair -- its second lattice has only
    //  the first half's coefficient kernel to hide behind -- and not measured for score-only calls)
    bool overlap = do_fwd && do_bwd && N >= 2 && p.offsets == nullptr && p.maxT + p.maxU - 1 >= kOverlapMinDiagonals &&
                   t_aux.stream != p.stream;
    // Where to cut: the second half's tensors must start on a 16-byte boundary like the whole batch's do, or its kernels would
    // take other forms (row-form gradient kernel, no 2-D statistics tiles) and the "same bits" promise would not hold -- the
    // sample count nearest N/2 whose slab is a whole number of 16-byte packets (none within 8 of N/2: no split)
    int n0 = N / 2;
    if (overlap) {
        const unsigned long long per_sample = static_cast<unsigned long long>(p.cells_per_sample) * A * sizeof(S);
        n0 = 0;
        for (int d = 0; d <= 8 && n0 == 0; ++d)
            for (int c : {N / 2 - d, N / 2 + d})
                if (c >= 1 && c < N && (per_sample * static_cast<unsigned long long>(c)) % 16 == 0) { n0 = c; break; }
        overlap = n0 != 0;
    }
    overlap = overlap && aux_prepare();          // (last: it may create events)
    if (!overlap) {
        if (prof) g_prof.split = false;           // (a two-half call whose events were never collected must not label this one)
        mark(0);
        if (do_fwd) launch_row_stats<Tag>(p, acts, vec_ok);
        mark(1);
        if (do_fwd) launch_lattice(p, training);
        mark(2);
        if (do_fwd && training) launch_coef(p);
        mark(3);
        if (do_bwd) launch_grad<Tag>(p, acts, grads, grad_scale, vec_ok);
        mark(4);
    } else {
        // the two-half schedule (see AuxStream above)
        Plan<C> half[2] = {sub_plan(p, 0, n0), sub_plan(p, n0, N - n0)};
        // kernel forms that depend on the batch size are chosen ONCE, for the whole batch: the halves run the
        // kernels the one-stream schedule would have run
        half[0].lat_form = half[1].lat_form = lattice_is_linear(p, training) ? 1 : 0;
        half[0].stats_tile2d = half[1].stats_tile2d = stats_is_tile2d<Tag>(p, acts) ? 1 : 0;   // (the cut keeps both halves on 16-byte boundaries)
        const size_t slab = static_cast<size_t>(n0) * p.cells_per_sample * A;          // elements of acts / grads in front of the second half
        const S* acts_h[2] = {acts, acts + slab};
        S* grads_h[2] = {grads, grads != nullptr ? grads + slab : nullptr};
        const C* scale_h[2] = {grad_scale, grad_scale != nullptr ? grad_scale + n0 : nullptr};
        hipStream_t aux = t_aux.stream;
        bool forked[2] = {false, false};
        auto pev = [&](hipEvent_t e, hipStream_t st) { if (prof) (void)hipEventRecord(e, st); };
        if (ranges) (void)g_ranges.push("warprnnt:two_half_schedule");
        for (int h = 0; h < 2; ++h) {
            pev(g_prof.hev[h][0], p.stream);
            launch_row_stats<Tag>(half[h], acts_h[h], vec_ok);
            pev(g_prof.hev[h][1], p.stream);
            // the fork.  If it cannot be made (an event of another device, a stream that is gone), this half's lattice simply
            // stays on the caller's stream -- the one-stream order, nothing left dangling on the auxiliary stream
            forked[h] = hipEventRecord(t_aux.ev[2 * h], p.stream) == hipSuccess && hipStreamWaitEvent(aux, t_aux.ev[2 * h], 0) == hipSuccess;
            if (!forked[h]) (void)hipGetLastError();
            // (the lattice kernel zeroes the batch's "has padding" word when it starts: only the first half's may -- the
            //  second runs beside the first half's coefficient kernel, which sets it -- so it gets a word of its own to clear)
            Plan<C> lat = half[h];
            lat.stream = forked[h] ? aux : p.stream;
            if (h == 1) lat.padflag = p.padflag + 1;
            pev(g_prof.lev[h][0], lat.stream);
            launch_lattice(lat, training);
            pev(g_prof.lev[h][1], lat.stream);
            if (lat.failed || (forked[h] && hipEventRecord(t_aux.ev[2 * h + 1], aux) != hipSuccess)) half[h].failed = true;
        }
        for (int h = 0; h < 2; ++h) {
            if (forked[h] && hipStreamWaitEvent(p.stream, t_aux.ev[2 * h + 1], 0) != hipSuccess) half[h].failed = true;   // the join
            pev(g_prof.hev[h][2], p.stream);
            if (training) launch_coef(half[h]);
            pev(g_prof.hev[h][3], p.stream);
            if (do_bwd) launch_grad<Tag>(half[h], acts_h[h], grads_h[h], scale_h[h], vec_ok);
            pev(g_prof.hev[h][4], p.stream);
            p.failed = p.failed || half[h].failed;
        }
        if (ranges) (void)g_ranges.pop();
        if (prof) { g_prof.split = true; g_prof.has_fwd = true; g_prof.has_bwd = do_bwd; }
    }
    if (p.failed) return RNNT_STATUS_EXECUTION_FAILED;

    if (costs_host) {
        // the reference contract: costs in HOST memory, call returns after a stream sync (gpu_rnnt.h:208-213)
        if (costs_direct == nullptr &&
            hipMemcpyAsync(costs_host, p.costs_dev, sizeof(C) * N, hipMemcpyDeviceToHost, p.stream) != hipSuccess)
            return RNNT_STATUS_MEMOPS_FAILED;
        if (hipStreamSynchronize(p.stream) != hipSuccess) return RNNT_STATUS_EXECUTION_FAILED;
        if (stage != nullptr) std::memcpy(costs_host, stage->host, sizeof(C) * static_cast<size_t>(N));
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

}  // namespace rnnt

using namespace rnnt;

// The library is compiled with -fvisibility=hidden: only the C entry points of include/rnnt.h are exported
// (the reference exports exactly its five: nm -D of its libwarprnnt.so).
#pragma GCC visibility push(default)
extern "C" {

int get_warprnnt_version() { return 1; }
int get_warprnnt_extension_version(void) { return 3; }

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
    Rccl::AllReduce all_reduce = nullptr;
    if (rccl_comm != nullptr && (all_reduce = rccl_all_reduce()) == nullptr) return RNNT_STATUS_EXECUTION_FAILED;   // no (unambiguous) RCCL in this process
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
                           label_lengths, sample, p.maxT, p.maxU, p.Up, p.lat_w, p.lat_sh, alpha_device, beta_device);
    } else {
        Plan<float> p;
        if (!make_plan(p, A, minibatch, options, const_cast<void*>(workspace), nullptr, label_lengths, input_lengths, static_cast<float*>(nullptr)))
            return RNNT_STATUS_INVALID_VALUE;
        hipLaunchKernelGGL((lattice_dump_kernel<float>), grid, dim3(256), 0, p.stream, p.alpha, p.beta, p.offa, p.offb, input_lengths,
                           label_lengths, sample, p.maxT, p.maxU, p.Up, p.lat_w, p.lat_sh, alpha_device, beta_device);
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
