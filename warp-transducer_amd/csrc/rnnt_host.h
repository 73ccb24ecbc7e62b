// rnnt_host.h -- what the two host translation units of the library share: workspace layout, stage timers, tuning
// constants, the launch plan, and the launchers of the two stages both paths run (lattice, coefficients).
//   rnnt_gpu.hip    the materialised path (row statistics ... gradient stream) and the C entry points of rnnt.h
//   rnnt_joint.hip  the additive-joint path and its entry points (compute_rnnt_loss_add*)
// Two translation units = two code objects: HIP loads a code object on the first launch of one of its kernels, and the
// additive-joint kernels are 45 % of the library's device code -- a caller of compute_rnnt_loss does not pay for loading
// them (first call of a process: tools/first_call.py).  Everything here is static / template code, compiled into both.
#pragma once

#include <atomic>
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/rnnt.h"
#include "rnnt_kernels.h"

namespace rnnt {

// ----------------------------------------------------------------------------- workspace
constexpr size_t kAlign = 256;
static inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

struct Layout {
    size_t lp2, logz, alpha, rowtab, beta, offa, offb, llf, llb, costs, padflag, poison, rowmax, side, side_bytes, wmat, total;
    size_t rowscale;         // packed layout with per-sample scales: one scale per packed row, for the gradient kernel (dead lattice blocks by then)
    size_t coef_done;        // per sample: tiles of the coefficient kernel that have finished reading its block (the overlay guard of coef_kernel)
    int group;               // samples whose records fit the head (N: the record table overlays nothing)
};

// The record table (16 B per row of the tensor for an fp32 lattice, written by the coefficient kernel, read by the gradient
// kernel) is ONE contiguous array at the front of the workspace -- and it OVERLAYS the lattice side data: the per-sample
// blocks [lp2 | logz | alpha | beta] (lat_block) start `head` bytes in.  The records of sample s end at rec * (s + 1) <=
// head + s * block -- the start of block s -- because head >= rec and a sample's records are smaller than its block
// (rec = 4 lat T U < 5 lat Dp Up): they can only fall on blocks of EARLIER samples, about 0.65 (s - head / rec) of them on
// long utterances.  Those must be dead -- read by every coefficient tile that needs them -- before the records are stored:
//   * the tiled coefficient kernel (maxU > 48) stays ONE launch, sample-major, and guards the overlay itself: a tile
//     announces the end of its reads in a per-sample counter and waits, before storing, for the tiles of the samples its
//     records can touch (coef_kernel; they were dispatched thousands of workgroups earlier: the wait is a formality);
//   * the cell-per-thread form (small lattices, many samples) runs in groups of `group` samples, launch after launch: when
//     group [s, s + m) runs every block below s is dead and its records end below head + s * block (head >= rec * m).
// Padded and packed layout alike (packed rows only come earlier); one-call, two-phase and two-half forms (each half's
// coefficient kernel runs in stream order behind the other half's coefficient and gradient kernels).
// The workspace is head + N blocks instead of records + N blocks: c4 (N=64, T=1500, U=301) 1.18 -> 0.77 GB against the
// reference's 0.35 (src/rnnt_entrypoint.cpp:96-128: three words per cell).  Small problems keep ONE group (one launch): the
// head is the whole table below kOneGroupBytes, never less than that above it (get_workspace_size stays monotone).
constexpr size_t kOneGroupBytes = 32u << 20;
constexpr int kCoefGroups = 8;

// lat = bytes of one lattice value (4: fp32 lattice for 16/32-bit activations, 8: fp64).
// joint: also the additive-joint planes (get_workspace_size_add); they sit BEHIND everything the
// materialised path uses, so a plan carved with joint = true is valid for both.
static Layout make_layout(int maxT, int maxU, int N, size_t lat, bool joint) {
    const size_t D = lat_rows(maxT, maxU);      // diagonals + padding rows
    const size_t Up = lat_stride(maxU);         // row stride of the skewed arrays
    const size_t W = (Up + 63) / 64;            // wavefronts of a lattice block at one column per lane
    const size_t block = lat_block(maxT, maxU, static_cast<int>(Up)) * lat;      // one sample's [lp2 | logz | alpha | beta], bytes
    const size_t rec1 = static_cast<size_t>(maxT) * maxU * 4 * lat;              // one sample's records, bytes
    const size_t recs = rec1 * N;
    Layout l{};
    // head of the workspace = the part of the record table no block lies under
    size_t head = recs;
    l.group = N;
#ifdef RNNT_DEV
    // development build only (RNNT_OVHEAD=1): the smallest legal head -- one sample's records -- whatever the table's size, so
    // that the overlay guard of coef_kernel really has to WAIT (tools/overlay_fuzz.py under the dev library: a stress test)
    static const bool tight = getenv("RNNT_OVHEAD") != nullptr;
    if (tight && N > 1) { head = rec1; l.group = 1; } else
#endif
    if (recs > kOneGroupBytes) {
        head = (recs + kCoefGroups - 1) / kCoefGroups;
        if (head < kOneGroupBytes) head = kOneGroupBytes;
        if (head < rec1) head = rec1;
        l.group = static_cast<int>(head / rec1);        // >= 1
        if (l.group > N) l.group = N;
    }
    head = align_up(head);
    l.rowtab = 0;
    l.lp2 = head;
    l.logz = head + lat_block_logz(maxT, maxU, static_cast<int>(Up)) * lat;
    l.alpha = head + lat_block_alpha(maxT, maxU, static_cast<int>(Up)) * lat;
    l.beta = head + lat_block_beta(maxT, maxU, static_cast<int>(Up)) * lat;
    size_t o = align_up(head + block * N);
    // one scale per packed row (packed layout with grad_scale): behind the record table's last possible row, inside blocks
    // that are dead when the gradient stage starts (5 lat T U N <= head + N block)
    l.rowscale = align_up(recs);
    if (l.rowscale + static_cast<size_t>(maxT) * maxU * N * lat > o) o = align_up(l.rowscale + static_cast<size_t>(maxT) * maxU * N * lat);   // (cannot happen: kept as a guard)
    l.offa = o;  o = align_up(o + D * W * N * sizeof(double));
    l.offb = o;  o = align_up(o + (D * W * N + D) * sizeof(double));
    l.llf = o;   o = align_up(o + N * sizeof(double));
    l.llb = o;   o = align_up(o + N * sizeof(double));
    l.costs = o; o = align_up(o + N * sizeof(double));
    l.coef_done = o; o = align_up(o + kCoefDoneStride * static_cast<size_t>(N) * sizeof(int));
    l.padflag = o; o = align_up(o + 4 * sizeof(int));   // [0] "some record of this batch is padding": zeroed by the lattice kernel, set by the coefficient kernel, read by the gradient kernel; [1] the same for the second half of a two-half call; [2] samples whose lattice block the record table has overlaid (lattice dump)
    l.poison = o; o = align_up(o + N * sizeof(int));   // per sample: the statistics kernels' hint of a row with a non-finite log Z (rnnt_kernels.h: note_non_finite)
    // additive joint only: row maxima of f and g, dense matrices W, CB, CL (row stride = maxU rounded up to 8)
    l.rowmax = o; l.wmat = o; l.side = o; l.side_bytes = 0;
    if (joint) {
        o = align_up(o + ((static_cast<size_t>(maxT) + maxU) * N + 2) * sizeof(float));   // + the +inf sentinel + the gate word
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
    std::atomic<bool> on{false};   // read without the mutex to decide whether to take it; everything else below is guarded by g_prof_mu
    bool ready = false;
    hipEvent_t ev[6];
    double ms[5] = {0, 0, 0, 0, 0};   // statistics, lattice, coefficients, gradient, first event to last
    int calls = 0;
    bool pending = false;   // events of an asynchronous call recorded, not yet read
    bool has_fwd = false, has_bwd = false;
    // A call in the two-half schedule (rnnt_set_aux_stream, run_gpu) records these instead of ev[1..4]: on the caller's stream
    // h0 / h1 = {before the half's statistics, after them, before its coefficients, after them, after its gradient kernel};
    // on the auxiliary stream the start and end of each half's lattice kernel.
    bool split = false;
    hipEvent_t hev[2][5], lev[2][2];
};
extern Profile g_prof;     // one instance for the library (defined in rnnt_gpu.hip)
extern std::mutex g_prof_mu;   // held by a profiled call from its first event record to its last, and by the
                               // rnnt_profile_* entries: concurrent callers cannot tear the shared event set (their
                               // calls are serialised while the timers are on; off -- the default -- nobody takes it)

// `locked`: the caller holds g_prof_mu (it took it because it saw `on`); without the lock nothing of the shared event set is touched
static bool prof_prepare(bool locked) {
    if (!locked || !g_prof.on.load(std::memory_order_relaxed)) return false;   // (switched off between the caller's test and its lock: a plain, unprofiled call)
    if (!g_prof.ready) {
        for (auto& e : g_prof.ev)
            if (hipEventCreate(&e) != hipSuccess) return false;
        for (auto& h : g_prof.hev) for (auto& e : h) if (hipEventCreate(&e) != hipSuccess) return false;
        for (auto& h : g_prof.lev) for (auto& e : h) if (hipEventCreate(&e) != hipSuccess) return false;
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

static inline void prof_accumulate() {
    float ms = 0.f;
    if (g_prof.split) {
        // two-half schedule: statistics, coefficients and gradient = the sums over the halves (they run back to back on the
        // caller's stream); lattice = what the auxiliary stream spent on it, CONCURRENTLY with the other half's streaming
        // kernels -- it is not part of the critical path, ms[4] (first event to last) is
        auto add = [&](double& acc, hipEvent_t a, hipEvent_t b) { if (hipEventElapsedTime(&ms, a, b) == hipSuccess) acc += ms; };
        for (int h = 0; h < 2; ++h) {
            add(g_prof.ms[0], g_prof.hev[h][0], g_prof.hev[h][1]);
            add(g_prof.ms[1], g_prof.lev[h][0], g_prof.lev[h][1]);
            if (g_prof.has_fwd && g_prof.has_bwd) {
                add(g_prof.ms[2], g_prof.hev[h][2], g_prof.hev[h][3]);
                add(g_prof.ms[3], g_prof.hev[h][3], g_prof.hev[h][4]);
            } else if (g_prof.has_fwd) {
                add(g_prof.ms[2], g_prof.hev[h][2], g_prof.hev[h][3]);
            }
        }
        add(g_prof.ms[4], g_prof.hev[0][0], g_prof.hev[1][g_prof.has_bwd ? 4 : 3]);
        g_prof.calls++;
        g_prof.pending = g_prof.has_fwd = g_prof.has_bwd = g_prof.split = false;
        return;
    }
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

// ----------------------------------------------------------------------------- stage ranges for external profilers
// rocprofv3 --kernel-trace shows kernel names only; with ranges on (rnnt_profile_enable bit 1, or WARPRNNT_ROCTX=1 in the
// environment) every call brackets the ENQUEUE of its four stages with roctx ranges -- the counterpart of the reference's
// DEBUG_TIME stage timers (include/detail/gpu_rnnt.h:112-122) -- which `rocprofv3 --marker-trace` puts on the same
// timeline as the kernels.  The marker library is looked up at run time (librocprofiler-sdk-roctx, else libroctx64):
// the library does not link against a profiler, and without one the switch does nothing.
struct Ranges {
    std::atomic<int> mode{-1};     // -1: not decided (environment), 0 off, 1 on
    std::once_flag resolved;       // the two entry points below are written once, inside call_once, and only read afterwards
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
};
extern Ranges g_ranges;            // one instance for the library (defined in rnnt_gpu.hip)

static bool ranges_prepare() {
    int mode = g_ranges.mode.load(std::memory_order_relaxed);
    if (mode < 0) {
        const char* e = getenv("WARPRNNT_ROCTX");
        int expected = -1;
        mode = (e != nullptr && atoi(e) > 0) ? 1 : 0;
        if (!g_ranges.mode.compare_exchange_strong(expected, mode, std::memory_order_relaxed)) mode = expected;   // (rnnt_profile_enable got there first)
    }
    if (mode != 1) return false;
    std::call_once(g_ranges.resolved, [] {          // concurrent first calls: one resolves, the others wait; both pointers or neither
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h == nullptr) continue;
            auto push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push != nullptr && pop != nullptr) { g_ranges.push = push; g_ranges.pop = pop; break; }
        }
    });
    return g_ranges.push != nullptr;
}

// boundary i of a call (the same five as prof_mark): closes stage i-1, opens stage i
static void ranges_mark(int i, bool do_fwd, bool do_bwd, const char* const names[4]) {
    auto active = [&](int stage) { return stage >= 0 && stage < 4 && (stage < 3 ? do_fwd : do_bwd); };
    if (active(i - 1)) (void)g_ranges.pop();
    if (active(i)) (void)g_ranges.push(names[i]);
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
// pskip = the gradient kernel's skip-padded-rows forms on/off (pskipb: rows from this size on always test the record first,
//         pskipmin .. pskipb: only when the batch has padding -- the flag of the coefficient kernel), joh = one-hot df corrections in the
// additive-joint DF kernel (-1: vocabularies <= 256), lat2 = two lattice columns per lane (-1: maxU > 256),
// jsamp = additive joint: sampled row references + guard instead of the row-maximum pass (rnnt_joint_kernels.h) on/off,
// j16 = bf16 storage on the bf16 matrix cores (rnnt_joint16_kernels.h), bit 0 DG, bit 1 DF, bit 2 Z; j16pf = operand ping-pong there, j16nt = columns per lane of its DF / DG (8 | 4),
// latlin = linear-domain lattice kernel (chain + helper wavefronts) for one-wavefront fp32 lattices: 0 off, 1 up to one block per CU,
//          2 at any size and every block takes the log-domain fallback (tests), 3 at any size,
// tile2d = 2-D cell-tile statistics kernel for short rows under wide lattices: 0 off, 1 = 8 x 32 cells per block, 2 = 16 x 16 (default),
// jfsum = additive joint: the correction sums of the gradient GEMMs formed inside the tiled coefficient kernel on/off,
// jsplit = additive joint, small vocabularies: the wavefronts of a DF / DG block split the contraction instead of the columns on/off,
// jnocb = additive joint, one-hot DF behind the tiled coefficient kernel: blank corrections from the row sums, no CB plane on/off,
// t2ord = tile order of the 2-D statistics kernel: 0 an eighth of the batch per XCD, 1 plain, 2 an eighth of each sample per XCD.
struct Tune { int sw = 4, nta = 1, gmax = 4194304, rows = 0, tile = 1, tilekb = 52, ppt = 2;
              int jfnk = 0, jfpf = 1, jgnk = 0, jgpf = 1, blk = 1, jzs = 0, xcd = 1, ctile = 1, pskip = 1, joh = -1;
              int lat2 = -1, xst = 0, jsamp = 1, tilemax = kTileMaxRowBytes, j16 = 7, j16pf = 1, j16nt = 4, tile2d = 2, latlin = 1, pskipb = 8192, pskipmin = 128, jfsum = 1, jsplit = 1, jnocb = 1, t2ord = 2;
              int ovg = 2; };   // ovg (dev): overlay guard of the tiled coefficient kernel: 0 none (UNSAFE: timing only), 1 announce only (UNSAFE), 2 announce + wait
#ifdef RNNT_DEV
static Tune read_tune() {
    Tune t;
    const char* e = getenv("RNNT_TUNE");
    if (e == nullptr) return t;
    const struct { const char* key; int* dst; } keys[] = {
        {"sw", &t.sw}, {"nta", &t.nta}, {"gmax", &t.gmax}, {"rows", &t.rows}, {"tile", &t.tile},
        {"tilekb", &t.tilekb}, {"ppt", &t.ppt}, {"jfnk", &t.jfnk}, {"jfpf", &t.jfpf}, {"jgnk", &t.jgnk},
        {"jgpf", &t.jgpf}, {"blk", &t.blk}, {"jzs", &t.jzs}, {"xcd", &t.xcd}, {"ctile", &t.ctile},
        {"pskip", &t.pskip}, {"joh", &t.joh}, {"lat2", &t.lat2}, {"xst", &t.xst}, {"jsamp", &t.jsamp}, {"tilemax", &t.tilemax},
        {"j16", &t.j16}, {"j16pf", &t.j16pf}, {"j16nt", &t.j16nt}, {"tile2d", &t.tile2d}, {"latlin", &t.latlin}, {"pskipb", &t.pskipb}, {"pskipmin", &t.pskipmin}, {"jfsum", &t.jfsum}, {"jsplit", &t.jsplit}, {"jnocb", &t.jnocb}, {"t2ord", &t.t2ord}, {"ovg", &t.ovg}};
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
    // RNNT_TUNE_LIVE set: RNNT_TUNE is read again on EVERY call, so one process can alternate variants on the same buffers
    // (some kernels' times depend on where the process's memory landed: tools/c4_align_probe.py).  Dev build, one thread.
    static const bool live = getenv("RNNT_TUNE_LIVE") != nullptr;
    static Tune t = read_tune();           // function-local static: initialised once, thread-safe
    if (live) t = read_tune();
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
    LogPair<C>* lp2; C *logz, *alpha, *beta; Cell<C>* rowtab;     // lp2 .. beta: SAMPLE 0's arrays; sample b's are lat_sample(b) values on (lp2: lat_sample_pair)
    C* rowscale = nullptr;         // packed layout with per-sample scales (launch_grad)
    int coef_group = 0;            // samples per coefficient launch of the cell-per-thread form (make_layout)
    bool overlay = false;          // the record table reaches into the lattice blocks (whole batch's layout)
    int* coef_done = nullptr;      // the tiled coefficient kernel's per-sample counters (overlay guard)
    int first_sample = 0;          // a half of the two-half schedule: its first sample in the whole batch (the overlay bookkeeping is global)
    size_t head_bytes = 0, block_bytes = 0;
    double *offa, *offb, *llf, *llb;
    int* padflag;
    int* poison;
    float *rowmax, *wmat, *side;
    size_t side_bytes = 0;
    int lat_cols = 1;              // lattice columns per lane (1 | 2), its wavefronts per block ...
    int lat_w = 1, lat_sh = 6;     // ... and the column -> wavefront shift (coefficient kernels)
    int lat_form = -1;             // lattice kernel form: -1 by the rule of launch_lattice, 0 log-domain, 1 linear-domain chain (a half of the
                                   // two-half schedule takes the WHOLE batch's choice: same kernels, same bits)
    int stats_tile2d = -1;         // 2-D tile statistics kernel: -1 by the rule of stats_is_tile2d, 0 no, 1 yes (halves again)
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
    if (joint && N > kGridSamples) return false;   // (the additive-joint kernels keep the samples on ONE grid dimension)
    if (static_cast<long long>(N) * 2 > 0x7fffffffLL) return false;   // lattice kernel: 2 N blocks on gridDim.x
    p.Up = lat_stride(p.maxU);
    // one sample's skewed lp2 array is addressed through a buffer descriptor with a 32-bit size
    if (lat_rows(p.maxT, p.maxU) * p.Up * sizeof(LogPair<C>) >= (1ull << 31)) return false;
    // lattice kernel form: one wavefront for maxU <= 64; one column per lane while every wavefront of the block
    // has a SIMD to itself (maxU <= 256), two columns per lane beyond (measured, ns per diagonal at T = 1500,
    // one / two columns: U=128 79 / 97, U=192 93 / 111, U=256 120 / 117, U=301 148 / 139, U=512 205 / 188)
    const int lat2 = tune().lat2 >= 0 ? tune().lat2 : (p.Up > 256 ? 1 : 0);
    p.lat_cols = p.Up <= 64 ? 1 : ((p.Up > 512 || lat2) ? 2 : 1);
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
    p.rowscale = reinterpret_cast<C*>(ws + lay.rowscale);
    p.coef_group = lay.group;
    p.overlay = lay.group < N;
    p.coef_done = reinterpret_cast<int*>(ws + lay.coef_done);
    p.head_bytes = lay.lp2;
    p.block_bytes = lat_block(p.maxT, p.maxU, p.Up) * sizeof(C);
    p.offa = reinterpret_cast<double*>(ws + lay.offa);
    p.offb = reinterpret_cast<double*>(ws + lay.offb);
    p.llf = reinterpret_cast<double*>(ws + lay.llf);
    p.llb = reinterpret_cast<double*>(ws + lay.llb);
    p.padflag = reinterpret_cast<int*>(ws + lay.padflag);
    p.poison = reinterpret_cast<int*>(ws + lay.poison);
    p.rowmax = reinterpret_cast<float*>(ws + lay.rowmax);
    p.wmat = reinterpret_cast<float*>(ws + lay.wmat);
    p.side = reinterpret_cast<float*>(ws + lay.side);
    p.side_bytes = lay.side_bytes;
    p.costs_dev = costs_device_out ? costs_device_out : reinterpret_cast<C*>(ws + lay.costs);
    return true;
}

// Compute units of the current device (256 on an MI355X in SPX mode, fewer in the partitioned modes), asked once per device.
static inline int device_cus() {
    static std::atomic<int> cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// The linear-domain chain kernel (one-wavefront fp32 lattices, at most one block per compute unit) or the log-domain one?
template <typename C> static bool lattice_is_linear(const Plan<C>& p, bool with_beta) {
    if (p.Up > 64 || sizeof(C) != 4 || !tune().latlin) return false;
    if (p.lat_form >= 0) return p.lat_form == 1;
    return tune().latlin >= 2 || p.N * (with_beta ? 2 : 1) <= device_cus();
}

// Stage 2: alpha (and, for gradients, beta) recursion; writes the costs.
template <typename C> static void launch_lattice(Plan<C>& p, bool with_beta) {
    const int dirs = with_beta ? 2 : 1;
#define RNNT_LATTICE(MW, CC)                                                                                     \
    hipLaunchKernelGGL((lattice_kernel<C, MW, CC>), dim3(p.N * dirs), dim3(p.lat_w * 64), 0, p.stream, p.lp2,          \
                       p.alpha, p.beta, p.offa, p.offb, p.llf, p.llb, p.costs_dev, p.input_lengths, p.label_lengths,  \
                       p.maxT, p.maxU, p.Up, dirs, p.padflag, p.logz, p.poison, p.coef_done)
    // One-wavefront fp32 lattices with at most one block per compute unit: the linear-domain chain with helper wavefronts
    // (range guard + log-domain fallback inside).  Its eight wavefronts per (sample, direction) buy latency with idle
    // SIMDs; past one block per CU there are none and the one-wavefront kernel is the faster again (N=128 T=200 U=41:
    // 13.6 us against 17.2; N=192: 18.9 against 17.4; N=1024: 57 against 37).
    if (lattice_is_linear(p, with_beta)) {
        if constexpr (sizeof(C) == 4)
            hipLaunchKernelGGL((lattice_lin_kernel<0>), dim3(p.N * dirs), dim3(kLinThreads), 0, p.stream, p.lp2, p.alpha, p.beta, p.offa, p.offb,
                               p.llf, p.llb, p.costs_dev, p.input_lengths, p.label_lengths, p.maxT, p.maxU, p.Up, dirs,
                               tune().latlin == 2 ? 1 : 0, p.padflag, p.logz, p.poison, p.coef_done);
    }
    else if (p.Up <= 64) RNNT_LATTICE(1, 1);                       // one wavefront, no synchronisation
    else if (p.lat_cols == 1) RNNT_LATTICE(8, 1);                  // maxU <= 512, one column per lane
    else if (p.lat_w <= 4) RNNT_LATTICE(4, 2);                     // two columns per lane, one wavefront per SIMD
    else RNNT_LATTICE(8, 2);                                       // maxU <= 1024 in at most 8 wavefronts
#undef RNNT_LATTICE
    p.check();
}

// Stage 3: gradient coefficients per row into the natural-order row table.
// joint: the additive-joint path (dense planes for its gradient GEMMs); sums != nullptr: also their correction sums
// {sfb, sgb, sgl, farflag} -- formed inside the tiled kernel; returns false when the caller still has to run
// joint_sums_kernel (the cell-per-thread form of small lattices leaves them to it)
struct JointSums { float *sfb, *sgb, *sgl; int* farflag; };
template <typename C> static bool coef_is_tiled(const Plan<C>& p) { return !(p.maxU <= 48 || !tune().ctile); }
template <typename C> static bool launch_coef(Plan<C>& p, bool joint = false, bool onehot = false, const JointSums* sums = nullptr) {
    float* wmat = joint ? p.wmat : nullptr;
    const int Upad = joint_upad(p.maxU);
    // additive joint: W and CL planes always (the DF kernel takes its label corrections from CL); small
    // vocabularies add CB and drop the records (a far cell's c is kept in their memory)
    int planes = onehot ? joint_planes_onehot(p.maxU) : (joint ? 2 : 1);
    // the tiled kernel forming the sums itself: nothing reads cb per cell (joint_df_kernel<..., BS> and the epilogue form both take
    // the row sums) nor the records -- W and CL only, whatever the vocabulary
    if (coef_is_tiled(p) && sums != nullptr && joint_planes_onehot(p.maxU) == 4 && tune().jnocb) planes = 5;
    // Launches of at most `coef_group` samples, in stream order: the records of a group overlay the lattice blocks of the
    // samples in front of it, which are dead by then (make_layout).  `recycled`: how many blocks, counted from sample 0 of
    // the WHOLE batch, lie under the records written so far -- left in the workspace for compute_rnnt_loss_lattice_dump.
    const int step = p.overlay && p.coef_group > 0 && p.coef_group < kGridSamples ? p.coef_group : kGridSamples;   // (cell-per-thread form only)
    const size_t rec1 = static_cast<size_t>(p.cells_per_sample) * sizeof(Cell<C>);
    auto recycled_after = [&](int b_end) -> int {          // b_end: samples of this plan processed, exclusive
        const size_t end = rec1 * (static_cast<size_t>(p.first_sample) + b_end);
        if (end <= p.head_bytes || p.block_bytes == 0) return 0;
        return static_cast<int>((end - p.head_bytes + p.block_bytes - 1) / p.block_bytes);
    };
    if (!coef_is_tiled(p)) {
        // small lattices: one thread per skewed cell, scattered record store
        const long long skew_cells = static_cast<long long>(p.maxT + p.maxU - 1) * ((p.Up + 63) / 64) * 64;   // whole 64-column segments
        for (int b0 = 0; b0 < p.N; b0 += step) {       // (samples on gridDim.y: slices of the batch)
            const int nb = p.N - b0 < step ? p.N - b0 : step;
            const dim3 cgrid(static_cast<unsigned>(((skew_cells + 255) / 256 + 7) / 8 * 8), nb);    // multiple of 8: XCD-aware remap
            hipLaunchKernelGGL((coef_cell_kernel<C>), cgrid, dim3(256), 0, p.stream, p.lp2, p.logz, p.alpha, p.beta, p.offa,
                               p.offb, p.llf, p.labels, p.input_lengths, p.label_lengths, p.rowtab, p.maxT, p.maxU, p.Up,
                               wmat, Upad, p.fastemit, planes, p.offsets, p.lat_w, p.lat_sh, b0, p.N, p.padflag, recycled_after(b0 + nb));
        }
    } else {
        const int DN = sizeof(C) == 4 ? 32 : 16;           // diagonals per tile (coef_kernel)
        const int tilesU = (p.maxU + 63) / 64, tilesN = (p.maxT + p.maxU - 1 + DN - 1) / DN;
        // ONE launch (slices of 65535 samples), the overlay guarded inside the kernel
        int* const done = (p.overlay && tune().ovg != 0) ? p.coef_done : nullptr;
        const unsigned long long head_arg = tune().ovg == 1 ? (~0ull >> 1) : static_cast<unsigned long long>(p.head_bytes);   // (dev, ovg = 1: nobody waits)
        const int slice = 0x7fffffff / (tilesU * tilesN) < kGridSamples ? 0x7fffffff / (tilesU * tilesN) : kGridSamples;   // samples per launch (grid limit)
        for (int b0 = 0; b0 < p.N; b0 += slice) {
            const int nb = p.N - b0 < slice ? p.N - b0 : slice;
            const int recycled = recycled_after(b0 + nb);
            const dim3 cgrid(static_cast<unsigned>(tilesU * tilesN) * static_cast<unsigned>(nb));   // one-dimensional, sample-major (coef_kernel)
            if (sums != nullptr)
                hipLaunchKernelGGL((coef_kernel<C, true>), cgrid, dim3(256), 0, p.stream, p.lp2, p.logz, p.alpha, p.beta, p.offa,
                                   p.offb, p.llf, p.labels, p.input_lengths, p.label_lengths, p.rowtab, p.maxT, p.maxU, p.Up,
                                   wmat, Upad, tilesU, p.fastemit, planes, p.offsets, p.lat_w, p.lat_sh, b0, p.N, p.padflag,
                                   sums->sfb, sums->sgb, sums->sgl, sums->farflag, recycled, done, static_cast<unsigned long long>(rec1),
                                   head_arg, static_cast<unsigned long long>(p.block_bytes), p.first_sample);
            else
                hipLaunchKernelGGL((coef_kernel<C, false>), cgrid, dim3(256), 0, p.stream, p.lp2, p.logz, p.alpha, p.beta, p.offa,
                                   p.offb, p.llf, p.labels, p.input_lengths, p.label_lengths, p.rowtab, p.maxT, p.maxU, p.Up,
                                   wmat, Upad, tilesU, p.fastemit, planes, p.offsets, p.lat_w, p.lat_sh, b0, p.N, p.padflag,
                                   static_cast<float*>(nullptr), static_cast<float*>(nullptr), static_cast<float*>(nullptr),
                                   static_cast<int*>(nullptr), recycled, done, static_cast<unsigned long long>(rec1),
                                   head_arg, static_cast<unsigned long long>(p.block_bytes), p.first_sample);
        }
        p.check();
        return sums != nullptr;
    }
    p.check();
    return false;
}

// options.loc as the int a caller really passed: a C program or ctypes can put ANY value there, and loading one outside the
// enumerators' range through the enum type is undefined behaviour in C++ (UBSan, `make asan`); every entry point answers an
// unknown location with RNNT_STATUS_INVALID_VALUE, as the reference does (src/rnnt_entrypoint.cpp:90-92).
static inline int loc_of(const rnntOptions& o) {
    int v = 0;
    static_assert(sizeof(v) == sizeof(o.loc), "rnntComputeLocation is int-sized");
    std::memcpy(&v, &o.loc, sizeof(v));
    return v;
}

static inline bool bad_args(const void* acts, const int* labels, const int* label_lengths,
                     const int* input_lengths, const void* costs, const void* workspace, int A, int N,
                     const rnntOptions& o) {
    // reference src/rnnt_entrypoint.cpp:49-59
    return acts == nullptr || labels == nullptr || label_lengths == nullptr || input_lengths == nullptr ||
           costs == nullptr || workspace == nullptr || A <= 0 || N <= 0 || o.maxT <= 0 || o.maxU <= 0;
}

}  // namespace rnnt
