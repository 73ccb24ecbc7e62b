// rnnt_gpu_impl.h -- the materialised path's host driver (launchers of the statistics and gradient stages, host-costs staging,
// the two-half schedule, run_gpu), as a header: run_gpu<Tag> is instantiated ONCE per storage type, each in a translation unit of
// its own --
//     rnnt_gpu.hip      F32  (+ every C entry point)        rnnt_gpu_f64.hip   F64        rnnt_gpu_h16.hip   BF16, F16
// -- because a translation unit is a code object and HIP loads a code object on the first launch of one of its kernels: with all
// four storage types in one object the first call of a process cost 3.9 ms, most of it loading kernels of types the caller never
// uses (tools/first_call.py; EXPERIMENTS.md 12).  The kernels that depend on the LATTICE type only (lattice_kernel, lattice_lin_kernel,
// coef_kernel, coef_cell_kernel, ... and joint_prep / joint_sums) are instantiated by several of these units with the same template
// arguments: they are `static` (internal linkage) so that every unit launches -- and every code object holds -- its OWN copy; as
// ordinary templates their host-side handles merged at link time, one module's copy served all of them, and a bf16 call loaded
// the fp32 unit's code object as well (`nm build/*.o` shows no weak kernel handle now).  State shared by the instantiations (staging buffers, the auxiliary stream, the
// profile timers) is defined once, in rnnt_gpu.hip; the small non-template helpers are `inline` so that their function-local
// statics are one object for the whole library.
#pragma once
#include "rnnt_cpu.h"
#include "rnnt_host.h"

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace rnnt {


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
            rowscale = p.rowscale;                 // (lattice blocks that are dead by now: make_layout)
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
extern std::mutex g_stage_mu;
extern std::vector<HostStage*> g_stage_all;            // every thread's buffer (for the release call); entries are never removed
extern std::atomic<int> g_stage_mode;                  // -1: not decided yet (environment), 0 off, 1 on
extern std::atomic<long long> g_stage_bytes;

inline bool stage_enabled() {
    int m = g_stage_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("WARPRNNT_HOST_STAGING");
        m = (e != nullptr && atoi(e) > 0) ? 1 : 0;
        g_stage_mode.store(m, std::memory_order_relaxed);
    }
    return m == 1;
}

inline void stage_free(HostStage* st) {                // (g_stage_mu held, or the owner thread with busy set)
    if (st->host != nullptr) {
        (void)hipHostFree(st->host);
        g_stage_bytes.fetch_sub(static_cast<long long>(st->cap), std::memory_order_relaxed);
    }
    st->host = st->dev = nullptr; st->cap = 0; st->device = -1;
}

// Returns the thread's staging record with `busy` set (the caller clears it), or nullptr: staging off, batch too
// large, or the allocation failed -- the caller then uses the asynchronous copy.
inline HostStage* stage_acquire(size_t bytes) {
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
extern thread_local AuxStream t_aux;            // per calling thread, like options.stream is per call

inline void aux_drop_events() {
    for (int i = 0; i < t_aux.made; ++i) (void)hipEventDestroy(t_aux.ev[i]);
    (void)hipGetLastError();
    t_aux.made = 0; t_aux.device = -1;
}

// The fork / join events of the calling thread, on the CURRENT device: a thread that moves to another GPU (the stream it
// hands over "must belong to the device of the call") gets events of that GPU -- the old ones are destroyed, not leaked.
// false: no auxiliary stream, or the events cannot be made -> the caller runs the one-stream schedule (nothing has
// been launched yet, so a failure here can never leave half a fork behind).
inline bool aux_prepare() {
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
    const size_t Dp = lat_rows(p.maxT, p.maxU), sk = lat_sample(b0, p.maxT, p.maxU, p.Up), skp = lat_sample_pair(b0, p.maxT, p.maxU, p.Up);
    q.N = n;
    q.first_sample = p.first_sample + b0;
    q.labels = p.labels + static_cast<size_t>(b0) * (p.maxU - 1);
    q.input_lengths = p.input_lengths + b0;
    q.label_lengths = p.label_lengths + b0;
    q.lp2 = p.lp2 + skp; q.logz = p.logz + sk; q.alpha = p.alpha + sk; q.beta = p.beta + sk;   // (per-sample blocks: rnnt_kernels.h, lat_block)
    q.rowtab = p.rowtab + static_cast<size_t>(b0) * p.cells_per_sample;
    q.offa = p.offa + static_cast<size_t>(b0) * p.lat_w * Dp;
    q.offb = p.offb + static_cast<size_t>(b0) * p.lat_w * Dp;
    q.llf = p.llf + b0; q.llb = p.llb + b0; q.poison = p.poison + b0; q.coef_done = p.coef_done + kCoefDoneStride * static_cast<size_t>(b0);
    q.costs_dev = p.costs_dev + b0;
    return q;
}

// The materialised path.  phases: bit 0 = forward part (row statistics, lattice and -- when gradients
// are wanted -- the coefficient table), bit 1 = gradient kernel; the two-call form
// (compute_rnnt_loss_fwd / _bwd) keeps only the workspace alive in between.  want_grad < 0: decided by
// `grads != nullptr` (the reference's "gradients == NULL means score only").
template <typename Tag>
rnntStatus_t run_gpu(const typename Tag::store* acts, typename Tag::store* grads,
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
    // gradients == activations is supported (every gradient kernel reads an element and writes the same element from the
    // same thread: rnnt.h, "IN PLACE"); tensors that overlap in any other way are not
    if (do_bwd && pg != pa) {
        const unsigned long long rows = p.offsets != nullptr ? p.packed_rows : static_cast<unsigned long long>(N) * p.cells_per_sample;
        const unsigned long long bytes = rows * static_cast<unsigned long long>(A) * sizeof(S);
        if ((pg > pa ? pg - pa : pa - pg) < bytes) return RNNT_STATUS_INVALID_VALUE;
    }
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

namespace rnnt {
// one definition per storage type (see the top of this file); every other translation unit only declares them
#ifndef RNNT_GPU_INSTANTIATE_F32
extern template rnntStatus_t run_gpu<F32>(const float*, float*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                          const rnntOptions&, int, int, float, const long long*, long long);
#endif
#ifndef RNNT_GPU_INSTANTIATE_F64
extern template rnntStatus_t run_gpu<F64>(const double*, double*, const int*, const int*, const int*, int, int, double*, double*, const double*, void*,
                                          const rnntOptions&, int, int, float, const long long*, long long);
#endif
#ifndef RNNT_GPU_INSTANTIATE_H16
extern template rnntStatus_t run_gpu<BF16>(const uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                           const rnntOptions&, int, int, float, const long long*, long long);
extern template rnntStatus_t run_gpu<F16>(const uint16_t*, uint16_t*, const int*, const int*, const int*, int, int, float*, float*, const float*, void*,
                                          const rnntOptions&, int, int, float, const long long*, long long);
#endif
}  // namespace rnnt
